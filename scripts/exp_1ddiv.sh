#!/bin/bash
# 1D kernels: IEEE float32 division (hipcc default) vs -fno-hip-fp32-correctly-rounded-divide-sqrt
for fl in "" "-fno-hip-fp32-correctly-rounded-divide-sqrt"; do
  python - <<PY || exit 1
from beacon_amd import build
build.FILE_FLAGS["env1d_f32.hip"] = "$fl".split()
build.build_lib(force=True, verbose=False)
PY
  echo "env1d flags: [$fl]"
  timeout -k 10 200 python scripts/bench_envs.py --no-cpu --only burgers,shkadov,sloshing 2>/dev/null | cut -c1-200 || exit 1
  timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -k "burgers or shkadov or sloshing" 2>&1 | tail -2
done
