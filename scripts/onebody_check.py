# diagnostic: float64 one-row-per-lane grids with a narrow last strip (one body, dead columns) against the generic kernel
import os, sys, time
sys.path.insert(0, os.getcwd())
from beacon_amd import jit, vec as V
grids = [(2.2, 1.28), (1.5, 1.0), (1.06, 1.0), (1.92, 1.06), (1.4, 1.06), (1.15, 1.2), (2.5, 1.1), (1.3, 1.0)]
bad = 0
for L, H in grids:
    env = V.VecRayleigh(2, "cuda:0", "f64", None, L=L, H=H)
    m = jit.choose(env.nx, env.ny, True, 0)
    env.close()
    mk = lambda B: V.VecRayleigh(B, "cuda:0", "f64", None, L=L, H=H)
    t = time.time()
    ok, rep = jit.compare_with_generic(mk, 0, True, ndt=12, batch=4)
    bad += not ok
    print("%dx%d %s %s  %s  (%.1fs)" % (env.nx, env.ny, m, "ok " if ok else "BAD", rep, time.time() - t), flush=True)
print("bad:", bad)
sys.exit(1 if bad else 0)
