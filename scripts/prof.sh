#!/bin/bash
# rocprofv3 evidence for one workload (run on the GPU box from the repo root):
#   1. kernel trace + stats (per-kernel durations)
#   2. PMC pass FETCH_SIZE, 3. PMC pass WRITE_SIZE (separate passes, as MI355X_MICROARCH.md prescribes)
#   4. PMC pass of SQ counters (VALU / LDS / scalar activity, wait states, instruction counts)
# usage: scripts/prof.sh <tag> <script.py> [args...]      e.g.  scripts/prof.sh r02_bench bench.py --steps 5 --warmup 1
# The profiled program is `python3 <script> <args>` directly behind `--` (no env/bash hop), everything is built
# BEFORE the first rocprofv3 line in a plain process, and BEACON_NO_BUILD=1 makes a stale library an error instead
# of a compiler spawned under the profiler's preload.
set -e
TAG=$1; shift
SCRIPT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
python3 -c "import sys; sys.path.insert(0, '$ROOT'); import __graft_entry__ as g; g.build()" > /dev/null
export BEACON_NO_BUILD=1
EXTRA=""
case "$SCRIPT" in *bench.py) EXTRA="--no-cpu --no-secondary";; *bench_envs.py) EXTRA="--no-cpu";; esac
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o run -- python3 $ROOT/$SCRIPT "$@" $EXTRA > $OUT/trace_stdout.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o run -- python3 $ROOT/$SCRIPT "$@" $EXTRA > $OUT/pmc_fetch_stdout.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o run -- python3 $ROOT/$SCRIPT "$@" $EXTRA > $OUT/pmc_write_stdout.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU --output-format csv -d $OUT/pmc_sq -o run -- python3 $ROOT/$SCRIPT "$@" $EXTRA > $OUT/pmc_sq_stdout.log 2>&1
python3 $ROOT/scripts/summarize_prof.py $OUT "rocprofv3 ... -- python3 $SCRIPT $* $EXTRA" > $OUT/summary.json
cp $OUT/summary.json $ROOT/gpurun_out/${TAG}_summary.json
find $OUT/trace -name '*kernel_stats.csv' -exec cp {} $ROOT/gpurun_out/${TAG}_kernel_stats.csv \;
cat $OUT/summary.json
