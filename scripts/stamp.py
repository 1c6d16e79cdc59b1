# diagnostic: build with -DBCN_STAMP and print cycles per Jacobi sweep (never used for timing claims)
import os, sys, subprocess, numpy as np, torch
sys.path.insert(0, os.getcwd())
from beacon_amd import build
build.FLAGS.append("-DBCN_STAMP"); build.build_lib(force=True)
from beacon_amd import vec as V
z = np.load("tests/golden/rayleigh_128x64_init.npz")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
env = V.VecRayleigh(B, "cuda:0", "f32", z["fields"], L=2.56, H=1.28)
env.reset()
acts = np.random.default_rng(0).uniform(-1, 1, (3, B, 10))
for k in range(3):
    torch.cuda.synchronize(); import time; t0 = time.perf_counter()
    env.step(acts[k]); torch.cuda.synchronize(); t1 = time.perf_counter()
    sw = env.sweeps.cpu().numpy()
    itp, cps = sw & 0xffff, sw >> 16
    print("step", k, "ms", round((t1 - t0) * 1e3, 2), "sweeps/dt", itp.mean(), "cycles/sweep mean", (cps * itp).sum() / itp.sum(),
          "clock MHz", env.status.cpu().numpy().mean(), "replica sweeps/dt min/max", itp.mean(1).min(), itp.mean(1).max(),
          "segments compute/reduce+write/barrier/post", env.actions_norm.cpu().numpy()[:, :4].mean(0).round(1),
          "min", cps.min(), "max", cps.max(), " total jacobi Mcycles/replica", (cps * itp).sum() / B / 1e6)
