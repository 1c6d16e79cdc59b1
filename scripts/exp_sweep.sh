#!/bin/bash
# timing experiments on the Jacobi sweep of ns2d_fast (diagnostic; EXP != 0 gives wrong results by construction)
export BCN_SCHED=0
for f in "$@"; do
  timeout -k 10 200 python scripts/stamp2.py $f 2>&1 | tail -1 || exit 1
done
