# diagnostic: where does the extrapolating residual plan (conv_plan 2) stop later than the reference's rule (conv_plan 0)?
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from beacon_amd import vec as V
from beacon_amd.envs import packaged_init
ra, n_sgts = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0e3, int(sys.argv[2]) if len(sys.argv) > 2 else 2
init = packaged_init("rayleigh")
acts = np.random.default_rng(5).uniform(-1, 1, (8, n_sgts))
res = {}
for plan in (0, 2, 3, 1):
    env = V.VecRayleigh(8, "cuda:0", "f32", init, n_sgts=n_sgts, ra=ra)
    env.set_variant(1)
    env.set_option("conv_plan", plan)
    env.reset(); env.step(acts); env.check_status()
    res[plan] = (env.sweeps.cpu().numpy().copy(), env.get_counters())
    env.close()
lit = res[0][0]
print("sweeps per timestep: mean %.1f min %d max %d" % (lit.mean(), lit.min(), lit.max()))
for plan in (2, 3, 1):
    sw, c = res[plan]
    d = np.argwhere(sw != lit)
    print("plan", plan, "late stops", c[:, 2].tolist(), "repeats", c[:, 3].tolist(), "differing solves", len(d))
    for b, t in d[:12]:
        print("   replica %d timestep %d: literal %d, plan %d: %d; neighbours literal %s" % (b, t, lit[b, t], plan, sw[b, t], lit[b, max(0, t - 2):t + 3].tolist()))
