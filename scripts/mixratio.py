import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from beacon_amd import vec as V
env = V.VecMixing(512, "cuda:0", "f32"); env.reset()
rng = np.random.default_rng(7)
for k in range(4):
    env.step(rng.integers(0, 4, 512))
sw = env.sweeps.cpu().numpy().astype(np.float64)
r = sw[:, 1:] / sw[:, :-1]
print("sweeps/dt mean %.1f; ratio next/prev: <0.5: %.3f  <0.625: %.3f <0.75: %.3f  <0.875: %.3f  >1: %.3f" % (sw.mean(), (r < 0.5).mean(), (r < 0.625).mean(), (r < 0.75).mean(), (r < 0.875).mean(), (r > 1).mean()))
w = sw[:, :-1]
for f in (0.5, 0.625, 0.75):
    ok = r >= f
    # sweeps whose evaluation a jump to f*prev would avoid vs wasted work on failures (a full wasted partial solve f*prev)
    print("jump %.3f: fails %.3f of solves; wasted sweeps %.1f per solve on average" % (f, (~ok).mean(), ((~ok) * f * w).sum() / r.size))
