"""float32 vs float64 register-resident kernels on the bench workload (same init, same actions):
how far the float32 path drifts from the float64 one over full action steps."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from beacon_amd import vec as V
z = np.load("tests/golden/rayleigh_128x64_init.npz")
B, K = 64, 3
acts = np.random.default_rng(1234).uniform(-1, 1, (K, B, 10))
envs = {dt: V.VecRayleigh(B, "cuda:0", dt, z["fields"], L=2.56, H=1.28) for dt in ("f32", "f64")}
for e in envs.values():
    e.reset()
for k in range(K):
    out = {}
    for dt, e in envs.items():
        obs, rwd, *_ = e.step(acts[k])
        e.check_status()
        out[dt] = (obs.double().cpu(), rwd.double().cpu(), e.get_state().double().cpu(), e.sweeps.cpu().numpy())
    d = lambda i: (out["f32"][i] - out["f64"][i]).abs().max().item()
    sw = np.abs(out["f32"][3] - out["f64"][3])
    print("step %d: max |obs| diff %.2e, reward %.2e, fields u/v/p/T %s, sweep counts differ in %.1f %% of timesteps (max %d)" % (
        k, d(0), d(1), ["%.1e" % (out["f32"][2][:, i] - out["f64"][2][:, i]).abs().max().item() for i in range(4)],
        100.0 * (sw > 0).mean(), sw.max()))
