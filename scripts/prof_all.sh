set -e
scripts/prof.sh r02f_bench bench.py --steps 5 --warmup 1 > gpurun_out/prof_r02f_bench.log 2>&1
scripts/prof.sh r02f_mixing scripts/bench_envs.py --only mixing --steps 4 > gpurun_out/prof_r02f_mixing.log 2>&1
scripts/prof.sh r02f_burgers scripts/bench_envs.py --only burgers --steps 20 > gpurun_out/prof_r02f_burgers.log 2>&1
scripts/prof.sh r02f_shkadov scripts/bench_envs.py --only shkadov --steps 20 > gpurun_out/prof_r02f_shkadov.log 2>&1
scripts/prof.sh r02f_sloshing scripts/bench_envs.py --only sloshing --steps 20 > gpurun_out/prof_r02f_sloshing.log 2>&1
ls gpurun_out/*_summary.json
