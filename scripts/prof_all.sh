set -e
scripts/prof.sh r02e_bench bench.py --steps 5 --warmup 1 > gpurun_out/prof_r02e_bench.log 2>&1
scripts/prof.sh r02e_mixing scripts/bench_envs.py --only mixing --steps 4 > gpurun_out/prof_r02e_mixing.log 2>&1
scripts/prof.sh r02e_burgers scripts/bench_envs.py --only burgers --steps 20 > gpurun_out/prof_r02e_burgers.log 2>&1
scripts/prof.sh r02e_shkadov scripts/bench_envs.py --only shkadov --steps 20 > gpurun_out/prof_r02e_shkadov.log 2>&1
scripts/prof.sh r02e_sloshing scripts/bench_envs.py --only sloshing --steps 20 > gpurun_out/prof_r02e_sloshing.log 2>&1
ls gpurun_out/*_summary.json
