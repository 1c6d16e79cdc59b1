import os, sys, subprocess
sys.path.insert(0, os.getcwd())
from beacon_amd import build
flag = sys.argv[1]
if flag != "none":
    build.FLAGS.append(flag)
build.build_lib(force=True)
os.environ["BEACON_NO_BUILD"] = "0"
r = subprocess.run([sys.executable, "scripts/bench_envs.py", "--only", "shkadov", "--steps", "30", "--no-cpu"], capture_output=True, text=True)
print(flag, r.stdout.strip().splitlines()[-1][:260] if r.stdout.strip() else r.stderr[-500:])
