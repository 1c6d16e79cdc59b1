import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from beacon_amd import vec as V
from beacon_amd.envs import packaged_init
from oracle import oracle as O
rng = np.random.default_rng(1)
# burgers
env = V.VecBurgers(1, "cuda:0", "f64"); env.reset(); o = O.burgers(); o.reset()
d = []
for k in range(20):
    a, n = rng.uniform(-1, 1), rng.uniform(-0.1, 0.1)
    env.step(np.array([a]), np.array([n])); o.step([a], n)
    d.append(np.abs(env.get_state().cpu().numpy()[0, 0] - o.u).max())
print("burgers f64 field maxdiff per step:", ["%.1e" % x for x in d[:3]], "... last", "%.1e" % d[-1])
# sloshing
init = packaged_init("sloshing")
env = V.VecSloshing(1, "cuda:0", "f64", init); env.reset(); o = O.sloshing(init_fields=init); o.reset()
d = []
for k in range(20):
    a = rng.uniform(-1, 1); env.step(np.array([a])); o.step([a])
    st = env.get_state().cpu().numpy()[0]; d.append(max(np.abs(st[0] - o.h).max(), np.abs(st[1] - o.q).max()))
print("sloshing f64 field maxdiff per step:", ["%.1e" % x for x in d[:3]], "... last", "%.1e" % d[-1])
# shkadov
init = packaged_init("shkadov")
env = V.VecShkadov(1, "cuda:0", "f64", init); env.reset(); o = O.shkadov(init_fields=init); o.rand_init = False; o.reset()
d = []
for k in range(5):
    a = rng.uniform(-1, 1, 5); nz = rng.uniform(-5e-4, 5e-4, 50)
    env.step(a[None], nz[None]); o.step(a.tolist(), nz)
    st = env.get_state().cpu().numpy()[0]; d.append(max(np.abs(st[0] - o.h).max(), np.abs(st[1] - o.q).max()))
print("shkadov f64 field maxdiff per step:", ["%.1e" % x for x in d])
