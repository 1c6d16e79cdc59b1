import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from beacon_amd import vec as V
z = np.load("tests/golden/rayleigh_128x64_init.npz")
B, steps = 512, 10
acts = torch.as_tensor(np.random.default_rng(1234).uniform(-1, 1, (steps + 2, B, 10)), dtype=torch.float32, device="cuda:0")
for q, tail in ((10, 6), (5, 6), (8, 6), (12, 6), (16, 6), (20, 6), (10, 4), (10, 8), (10, 12), (8, 8), (12, 4), (10, 6)):
    env = V.VecRayleigh(B, "cuda:0", "f32", z["fields"], L=2.56, H=1.28)
    env.set_sched(2, 0, q)
    env.set_option("sched_tail", tail)
    env.reset()
    for k in range(2):
        env.step(acts[k])
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for k in range(steps):
        env.step(acts[2 + k])
    e.record(); torch.cuda.synchronize()
    env.check_status()
    print("q %2d tail %2d: %.3f ms/step" % (q, tail, s.elapsed_time(e) / steps), flush=True)
    env.close()
