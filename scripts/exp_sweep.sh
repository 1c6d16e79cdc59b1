#!/bin/bash
# timing experiments on the Jacobi sweep of ns2d_fast (diagnostic; wrong results by construction)
export BCN_SCHED=0
for e in 0 1 3 5 7 9 15; do
  timeout -k 10 200 python scripts/stamp2.py -DBCN_EXP=$e 2>&1 | tail -1 || exit 1
done
