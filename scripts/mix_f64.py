import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
from beacon_amd import vec as V
for dt in ("f64",):
    env = V.VecMixing(512, "cuda:0", dt)
    env.reset()
    rng = np.random.default_rng(7)
    a = torch.as_tensor(rng.integers(0, 4, (3, 512)), device="cuda:0")
    env.step(a[0]); torch.cuda.synchronize()
    t0 = time.perf_counter(); env.step(a[1]); env.step(a[2]); torch.cuda.synchronize()
    print(dt, env.kernel_name, "ms/step %.1f" % ((time.perf_counter() - t0) * 500), "sweeps/dt %.1f" % env.sweeps.float().mean().item())
