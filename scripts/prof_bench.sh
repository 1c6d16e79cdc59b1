#!/bin/bash
# rocprofv3 kernel-trace summary of the default bench run (run on the GPU box from the repo root)
set -e
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu > $OUT/bench_stdout.log 2>&1
find $OUT -name "*stats*.csv" | head
