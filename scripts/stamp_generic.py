# diagnostic: cycles per Jacobi sweep in the generic kernel (compute part / block reduction part)
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from beacon_amd import build
build.FLAGS.append("-DBCN_STAMP"); build.build_lib(force=True)
from beacon_amd import vec as V
for name in ("mixing", "rayleigh"):
    if name == "mixing":
        env = V.VecMixing(512, "cuda:0", "f32"); env.set_ndt_act(20); env.reset(); env.step(np.zeros(512, dtype=np.int64))
    else:
        z = np.load("tests/golden/rayleigh_128x64_init.npz")
        env = V.VecRayleigh(512, "cuda:0", "f32", z["fields"], L=2.56, H=1.28); env.set_variant(0); env.set_ndt_act(20); env.reset()
        env.step(np.random.default_rng(0).uniform(-1, 1, (512, 10)))
    sw = env.sweeps.cpu().numpy()
    itp, c0, c1 = sw & 0xfff, ((sw >> 12) & 0x3ff) * 16, ((sw >> 22) & 0x3ff) * 16
    print(name, "sweeps/dt", itp.mean(), "cycles/sweep compute", c0.mean(), "reduce+barrier", c1.mean())
    env.close()
