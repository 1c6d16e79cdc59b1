"""Register / spill / scratch figures of the kernels of one csrc/*.hip (hipcc -Rpass-analysis=kernel-resource-usage):
    python scripts/resusage.py ns2d_fast.hip [name-substring] [extra hipcc flags...]
Compiles to /tmp (device code only), prints one line per kernel."""
import os, re, subprocess, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from beacon_amd import build as B
src = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else ""
extra = [a for a in sys.argv[2:] if a.startswith("-")]
path = src if os.path.exists(src) else os.path.join(B.CSRC, src)
cmd = [B.hipcc()] + B.FLAGS + B.FILE_FLAGS.get(os.path.basename(path), []) + extra + [
    "-I", B.INC, "-I", B.CSRC, "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-c", path, "-o", "/tmp/resusage.o"]
r = subprocess.run(cmd, capture_output=True, text=True)
if r.returncode:
    print(r.stderr[-4000:]); sys.exit(1)
cur = None
rows = {}
for ln in r.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", ln)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z][\w ]*?)(?: \[[^\]]*\])?: (\S+) \[-Rpass", ln)
    if m and cur:
        rows[cur][m.group(1)] = m.group(2)
for k, v in rows.items():
    if pat in k:
        print(k[:110])
        print("   ", {a: v.get(a) for a in ("TotalSGPRs", "VGPRs", "SGPRs Spill", "VGPRs Spill", "ScratchSize", "Occupancy") if a in v})
