# diagnostic: shkadov timestep phases (cycles per timestep of waves 0, 5, 10, 15: write+barrier / halo reads / compute)
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from beacon_amd import build
build.FLAGS.append("-DBCN_STAMP_1D"); build.build_lib(force=True)
from beacon_amd import vec as V
env = V.VecShkadov(1024, "cuda:0", "f32", None, L0=699.2, n_jets=10); env.reset()
a = torch.zeros((1024, 10), device="cuda:0")
for _ in range(3): env.step(a)
torch.cuda.synchronize()
o = env.obs.cpu().numpy()[:, :12].mean(0).reshape(4, 3)
print("waves 0,5,10,15: cycles per timestep [write+barrier, halo reads, compute]\n", o.round(0), "sum", o.sum(1).round(0))
