import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from beacon_amd import build
build.FLAGS.append("-DBCN_STAMP"); build.build_lib(force=True)
from beacon_amd import vec as V
env = V.VecMixing(512, "cuda:0", "f32"); env.set_ndt_act(40); env.reset()
for k in range(2):
    env.step(np.full(512, k, dtype=np.int64))
print("sweeps/dt", env.sweeps.float().mean().item(), "cycles/timestep BC/pred+rhs/jacobi/corr/transp-expl/chain", env.obs.cpu().numpy()[:, :6].mean(0).round(0))
