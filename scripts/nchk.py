# diagnostic: how many sweeps of a Jacobi solve evaluate the residual (-DBCN_DBG_NCHK build of the fast kernel)
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
from beacon_amd import build
opts = dict(a.split("=") for a in sys.argv[1:] if "=" in a and not a.startswith("-"))       # name=value -> bcn_set_option
build.FLAGS.extend(["-DBCN_DBG_NCHK"] + [a for a in sys.argv[1:] if a.startswith("-")]); build.build_lib(force=True)
from beacon_amd import vec as V
z = np.load("tests/golden/rayleigh_128x64_init.npz")
B = 512
env = V.VecRayleigh(B, "cuda:0", "f32", z["fields"], L=2.56, H=1.28)
for k_, v_ in opts.items():
    env.set_option(k_, int(v_))
env.reset()
acts = np.random.default_rng(1234).uniform(-1, 1, (3, B, 10))
for k in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    env.step(acts[k]); torch.cuda.synchronize(); t1 = time.perf_counter()
    sw = env.sweeps.cpu().numpy(); itp, nchk = sw & 0xffff, sw >> 16
    print("step %d: %.2f ms, sweeps/solve %.1f, evaluations/solve %.1f (%.1f %%), solves with <= 8 sweeps %.1f %%" %
          (k, (t1 - t0) * 1e3, itp.mean(), nchk.mean(), 100.0 * nchk.sum() / itp.sum(), 100.0 * (itp <= 8).mean()))
    for lo, hi in ((1, 8), (9, 32), (33, 64), (65, 128), (129, 1000)):
        m = (itp >= lo) & (itp <= hi)
        if m.any():
            print("   solves with %d..%d sweeps: %.1f %% of solves, %.1f %% of sweeps, evaluations %.1f of %.1f" %
                  (lo, hi, 100.0 * m.mean(), 100.0 * itp[m].sum() / itp.sum(), nchk[m].mean(), itp[m].mean()))
