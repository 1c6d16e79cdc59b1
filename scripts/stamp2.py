# diagnostic: build variants of the fast kernel (-D flags) and report cycles/sweep + step time
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
from beacon_amd import build
extra = sys.argv[1:]
for x in [x for x in extra if x.startswith("-ffp-contract=")]:      # per-file flags come last on the command line
    build.FILE_FLAGS = {k: [f for f in v if not f.startswith("-ffp-contract=")] + [x] for k, v in build.FILE_FLAGS.items()}
if os.environ.get("BCN_STAMP_NOBUILD") != "1":      # (scripts/variants.py puts a -DBCN_STAMP library in place itself)
    build.FLAGS.extend(["-DBCN_STAMP"] + extra); build.build_lib(force=True)
from beacon_amd import vec as V
z = np.load("tests/golden/rayleigh_128x64_init.npz")
B = 512
env = V.VecRayleigh(B, "cuda:0", os.environ.get("BCN_STAMP_DTYPE", "f32"), z["fields"], L=2.56, H=1.28)
env.set_sched(0)          # plain launch: the stamps are per workgroup = per replica
env.reset()
acts = np.random.default_rng(0).uniform(-1, 1, (3, B, 10))
for k in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    env.step(acts[k]); torch.cuda.synchronize(); t1 = time.perf_counter()
    sw = env.sweeps.cpu().numpy(); itp, cps = sw & 0xffff, sw >> 16
print(extra, "step ms %.2f" % ((t1 - t0) * 1e3), "sweeps/dt %.1f" % itp.mean(), "cycles/sweep %.0f" % ((cps * itp).sum() / itp.sum()),
      "clock MHz %.0f" % env.status.cpu().numpy().mean(),
      "cycles/timestep bcT+buoy/rhs/jacobi/corr/transp-expl+bcuv/(chain||pred)+barrier/chain", env.actions_norm.cpu().numpy()[:, :7].mean(0).round(0))
