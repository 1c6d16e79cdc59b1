import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from beacon_amd import vec as V
B, K = 8, 2
acts = np.random.default_rng(5).integers(0, 4, (K, B))
envs = {dt: V.VecMixing(B, "cuda:0", dt) for dt in ("f32", "f64")}
for e in envs.values(): e.reset()
for k in range(K):
    out = {}
    for dt, e in envs.items():
        obs, rwd, *_ = e.step(acts[k]); e.check_status()
        out[dt] = (obs.double().cpu(), rwd.double().cpu(), e.get_state().double().cpu(), e.sweeps.cpu().numpy())
    sw = np.abs(out["f32"][3] - out["f64"][3]); rel = sw / np.maximum(out["f64"][3], 1)
    print(k, envs["f32"].kernel_name, envs["f64"].kernel_name, "obs %.2e rwd %.2e fields %s sweeps: differ %.1f%% max %d maxrel %.3f" % (
        (out["f32"][0]-out["f64"][0]).abs().max().item(), (out["f32"][1]-out["f64"][1]).abs().max().item(),
        ["%.1e" % (out["f32"][2][:, i]-out["f64"][2][:, i]).abs().max().item() for i in range(4)], 100*(sw>0).mean(), sw.max(), rel.max()))
