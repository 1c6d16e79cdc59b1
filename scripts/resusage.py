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
    "-I", B.INC, "-I", B.CSRC, "--cuda-device-only", "--no-gpu-bundle-output", "-Rpass-analysis=kernel-resource-usage", "-c", path, "-o", "/tmp/resusage.o"]
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
# code size of every kernel (bytes of its function symbol in the gfx950 ELF): the instruction cache is a resource too
sizes = {}
readelf = os.path.join(os.path.dirname(os.path.dirname(B.hipcc())), "lib", "llvm", "bin", "llvm-readelf")
if os.path.exists(readelf):
    out = subprocess.run([readelf, "-sW", "/tmp/resusage.o"], capture_output=True, text=True).stdout
    for ln in out.splitlines():
        f = ln.split()
        if len(f) >= 8 and f[3] == "FUNC":
            name = subprocess.run(["c++filt", f[7]], capture_output=True, text=True).stdout.strip()
            sizes[name] = int(f[2])
for k, v in rows.items():
    if pat in k:
        print(k[:110])
        d = {a: v.get(a) for a in ("TotalSGPRs", "VGPRs", "SGPRs Spill", "VGPRs Spill", "ScratchSize", "Occupancy") if a in v}
        d["CodeBytes"] = sizes.get(k)
        print("   ", d)
