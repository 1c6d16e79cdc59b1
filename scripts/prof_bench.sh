#!/bin/bash
# rocprofv3 evidence for the default bench workload (run on the GPU box from the repo root):
#   1. kernel trace + stats (per-kernel durations)
#   2. PMC pass FETCH_SIZE, 3. PMC pass WRITE_SIZE (separate passes: TCC has 4 slots, FETCH 3 + WRITE 2)
# usage: scripts/prof_bench.sh <tag> [bench args]
set -e
TAG=${1:-r01}; shift || true
ARGS=${@:---steps 5 --warmup 1 --no-cpu}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 $ROOT/bench.py $ARGS > $OUT/trace_stdout.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_fetch_stdout.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_write_stdout.log 2>&1
python3 $ROOT/scripts/summarize_prof.py $OUT > $OUT/summary.json
cat $OUT/summary.json
