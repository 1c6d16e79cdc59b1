#!/bin/bash
# 1D kernels: cells-per-thread sweep on the bench configurations
for k in 8 4 2 1 0; do
  echo "BCN_1D_K=$k"
  BCN_1D_K=$k timeout -k 10 200 python scripts/bench_envs.py --no-cpu --only burgers,shkadov,sloshing 2>/dev/null | cut -c1-200 || exit 1
done
