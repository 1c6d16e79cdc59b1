#!/bin/bash
# kernel time (rocprofv3 --kernel-trace --stats) of one 1D env's step kernel over the launch shapes the dispatcher can
# pick: cells per thread K x one-wave-per-replica on/off.  usage (GPU box, repo root): scripts/sweep_1d.sh burgers
ENVN=${1:-burgers}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
python3 -c "import sys; sys.path.insert(0, '$ROOT'); import __graft_entry__ as g; g.build()" > /dev/null
export BEACON_NO_BUILD=1
cd /tmp && export TMPDIR=/tmp
for OW in 1 0; do for K in 1 2 4 8; do
  OUT=$ROOT/gpurun_out/sweep1d_${ENVN}_k${K}_ow${OW}
  rm -rf $OUT && mkdir -p $OUT
  export BCN_1D_K=$K BCN_1D_ONEWAVE=$OW
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python3 $ROOT/scripts/bench_envs.py --only $ENVN --steps 20 --no-cpu > $OUT/stdout.log 2>&1
  echo "K=$K onewave=$OW: $(find $OUT -name '*kernel_stats.csv' -exec grep -h "${ENVN}_step_k" {} \; | cut -d, -f1-4 | head -2 | tr '\n' ' ')"
done; done
