#!/bin/bash
# A/B: -ffp-contract=on vs the hipcc default (fast) on the headline bench (builds on the GPU box)
for rep in 1 2; do
for c in on fast; do
  python - <<PY || exit 1
from beacon_amd import build
fl = ["-fno-slp-vectorize", "-ffp-contract=$c"]
build.FILE_FLAGS = {"ns2d_fast.hip": fl, "ns2d_fast2.hip": fl}
build.build_lib(force=True, verbose=False)
PY
  echo "contract=$c: $(timeout -k 10 200 python bench.py --no-cpu 2>/dev/null | tail -1 | cut -c1-220)"
done
done
