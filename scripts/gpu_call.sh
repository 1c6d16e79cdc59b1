#!/bin/bash
# One gpurun call of the build -> measure loop (run on the GPU box from the repo root): the -m gpu tests with the error
# log behind the tolerances, the bench line, and a kernel trace of the N > 1 path with one rank.  A step that was killed
# (exit code >= 124) ends the call: nothing else touches the GPU behind it.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
step() { # name, timeout, command...
  local name=$1 tmo=$2; shift 2
  echo "== $name" | tee -a $OUT/steps.log
  timeout -k 10 $tmo "$@" > $OUT/$name.log 2> $OUT/$name.err
  local rc=$?
  echo "== $name rc=$rc" | tee -a $OUT/steps.log
  if [ $rc -ge 124 ]; then echo "killed: stop" | tee -a $OUT/steps.log; exit $rc; fi
  if [ $rc -ne 0 ] && [ $rc -gt $worst ]; then worst=$rc; fi      # the call ends with the worst code of its steps
  if [ $rc -ne 0 ] && [ "$name" = tests ]; then echo "tests failed: the later steps would measure a broken build: stop" | tee -a $OUT/steps.log; exit $rc; fi
  return $rc
}
worst=0
export BEACON_NO_BUILD=1
for what in "$@"; do
  case $what in
    tests)   BEACON_ERRLOG=$OUT/errlog.jsonl step tests 900 python3 -m pytest tests -m gpu -q -x ;;
    tests_all) BEACON_ERRLOG=$OUT/errlog.jsonl step tests 900 python3 -m pytest tests -m gpu -q ;;
    newtests) BEACON_ERRLOG=$OUT/errlog_new.jsonl step newtests 900 python3 -m pytest tests -m gpu -q -k "episode_drift or developed_film or episode_statistics or shkadov_vs_golden" ;;
    drift) BEACON_ERRLOG=$OUT/errlog_drift.jsonl step drift 900 python3 -m pytest tests -m gpu -q -k "episode_drift" ;;
    mix)     step mix 600 python3 scripts/mix_counters.py f32 ;;
    mixtests) step mixtests 900 python3 -m pytest tests -m gpu -q -k "mixing or fast2 or jit_grid or jit_grids or stop_rule or conv_plan or 100x100 or speculative" ;;
    smoke)   step smoke 300 python3 -c "import __graft_entry__ as g; g.smoke()" ;;
    f64t)    step f64tests 900 python3 -m pytest tests -m gpu -q -k "f64 or float64 or init_generation or identical"
             step kstat64 600 python3 scripts/kstat.py f64 6 ;;
    k:*)     BEACON_ERRLOG=$OUT/errlog_k.jsonl step tests_k 900 python3 -m pytest tests -m gpu -q -k "${what#k:}" ;;
    kstat)   step kstat 600 python3 scripts/kstat.py f32 8 ;;
    kstat64) step kstat64 600 python3 scripts/kstat.py f64 6 ;;
    mixstat) step mixstat 600 python3 scripts/mixstat.py f32 6 ;;
    # diagnostic builds (they REPLACE the box's library: keep them behind everything that measures the product)
    stamp)   BEACON_NO_BUILD=0 step stamp 1100 python3 scripts/stamp2.py ;;
    stamp64) BEACON_NO_BUILD=0 BCN_STAMP_DTYPE=f64 step stamp64 1100 python3 scripts/stamp2.py ;;
    stampmix) BEACON_NO_BUILD=0 step stampmix 1100 python3 scripts/stamp_mixing.py ;;
    dist)    step tests_dist 600 python3 -m pytest tests -m gpu -q -k "rccl or gloo or masked" ;;
    bench)   step bench 600 python3 bench.py --steps 20 --warmup 5 ;;
    benchq)  step benchq 600 python3 bench.py --steps 20 --warmup 5 --no-cpu --no-secondary ;;
    fdist)   step fdist 600 python3 bench.py --gpus 1 --force-dist --steps 20 --warmup 5 --no-cpu --no-secondary
             step fdist_noov 600 python3 bench.py --gpus 1 --force-dist --no-overlap --steps 20 --warmup 5 --no-cpu --no-secondary ;;
    trace_fdist)
      cd /tmp && export TMPDIR=/tmp
      step trace_fdist 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_fdist -o run -- python3 $ROOT/bench.py --gpus 1 --force-dist --steps 10 --warmup 2 --no-cpu --no-secondary
      find $OUT/trace_fdist -name '*kernel_stats.csv' -exec cp {} $OUT/trace_fdist_kernel_stats.csv \;
      find $OUT/trace_fdist -name '*kernel_trace.csv' -exec cp {} $OUT/trace_fdist_kernel_trace.csv \;
      rm -rf $OUT/trace_fdist
      cd $ROOT ;;
    *) echo "unknown step $what" ;;
  esac
done
exit $worst
