#!/bin/bash
python - <<PY || exit 1
from beacon_amd import build
build.FILE_FLAGS["env1d.hip"] = ["-fno-hip-fp32-correctly-rounded-divide-sqrt"]
build.build_lib(force=True, verbose=False)
PY
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -k "burgers or shkadov or sloshing" 2>&1 | grep -E "^E  |FAILED|passed|failed" | head -40
