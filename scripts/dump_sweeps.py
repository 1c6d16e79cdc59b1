import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from beacon_amd import vec as V
z = np.load("tests/golden/rayleigh_128x64_init.npz")
B = 512
env = V.VecRayleigh(B, "cuda:0", "f32", z["fields"], L=2.56, H=1.28)
env.reset()
acts = np.random.default_rng(1234).uniform(-1, 1, (8, B, 10))
out = []
for k in range(8):
    env.step(acts[k]); out.append(env.sweeps.cpu().numpy().astype(np.int16))
np.save("gpurun_out/sweeps_8steps.npy", np.stack(out))
