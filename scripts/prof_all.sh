set -e
# usage: scripts/prof_all.sh [tag]   (default r03) -- rocprofv3 evidence of every kernel at the BASELINE batches
T=${1:-r04}
scripts/prof.sh ${T}_bench bench.py --steps 5 --warmup 1 > gpurun_out/prof_${T}_bench.log 2>&1
scripts/prof.sh ${T}_f64 bench.py --dtype f64 --steps 3 --warmup 1 > gpurun_out/prof_${T}_f64.log 2>&1
scripts/prof.sh ${T}_mixing scripts/bench_envs.py --only mixing --steps 4 > gpurun_out/prof_${T}_mixing.log 2>&1
scripts/prof.sh ${T}_tall scripts/bench_envs.py --only tall --steps 3 > gpurun_out/prof_${T}_tall.log 2>&1
scripts/prof.sh ${T}_burgers scripts/bench_envs.py --only burgers --steps 20 > gpurun_out/prof_${T}_burgers.log 2>&1
scripts/prof.sh ${T}_shkadov scripts/bench_envs.py --only shkadov --steps 20 > gpurun_out/prof_${T}_shkadov.log 2>&1
scripts/prof.sh ${T}_sloshing scripts/bench_envs.py --only sloshing --steps 20 > gpurun_out/prof_${T}_sloshing.log 2>&1
ls gpurun_out/*_summary.json
