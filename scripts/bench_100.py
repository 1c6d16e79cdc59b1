import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
from beacon_amd import vec as V
B = 512; nx = ny = 100
rng = np.random.default_rng(17)
x, y = (np.arange(nx + 2) - 0.5) / nx, (np.arange(ny + 2) - 0.5) / ny
init = np.zeros((4, nx + 2, ny + 2))
init[3] = (0.5 - y)[None, :] + 0.1 * np.sin(4 * np.pi * x)[:, None] * np.sin(np.pi * y)[None, :]
acts = torch.as_tensor(rng.uniform(-1, 1, (3, B, 10)), dtype=torch.float32, device="cuda:0")
for variant in (1, 0):
    env = V.VecRayleigh(B, "cuda:0", "f32", init, L=2.0, H=2.0)
    env.set_variant(variant); env.reset()
    env.step(acts[0]); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(1, 3): env.step(acts[k])
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 2 * 1e3
    print("100x100 rayleigh %-18s %.1f ms/step  %.0f env steps/s  sweeps/dt %.1f" % (env.kernel_name, ms, B / ms * 1e3, env.sweeps.float().mean().item()))
    env.close()
