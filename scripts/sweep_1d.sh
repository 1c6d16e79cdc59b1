#!/bin/bash
# kernel time (rocprofv3 --kernel-trace --stats; the HIP-event time of a 40 us kernel is host-bound) of the 1D envs' step kernels.
#   scripts/sweep_1d.sh burgers          the launch shapes the dispatcher can pick: cells per thread K x one-wave-per-replica
#   scripts/sweep_1d.sh all              burgers, shkadov, sloshing as dispatched by default
# (GPU box, repo root)
ENVN=${1:-all}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
python3 -c "import sys; sys.path.insert(0, '$ROOT'); import __graft_entry__ as g; g.build()" > /dev/null
export BEACON_NO_BUILD=1
cd /tmp && export TMPDIR=/tmp
run() {   # $1 env, $2 label
  OUT=$ROOT/gpurun_out/sweep1d_tmp
  rm -rf $OUT && mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python3 $ROOT/scripts/bench_envs.py --only $1 --steps 20 --no-cpu $3 > $OUT/stdout.log 2>&1
  echo "$2: $(find $OUT -name '*kernel_stats.csv' -exec grep -h "${1}_step_k" {} \; | sed 's/.*)",//' | cut -d, -f1-3 | tr '\n' ' ') (calls,total_ns,avg_ns)"
  rm -rf $OUT
}
if [ "$ENVN" = all ]; then
  for e in burgers shkadov sloshing; do run $e $e; done
else
  for OW in 1 0; do for K in 1 2 4 8; do
    run $ENVN "$ENVN K=$K onewave=$OW" "--opt cells_per_thread=$K --opt one_wave=$OW"      # bcn_set_option
  done; done
fi
