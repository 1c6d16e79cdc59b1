"""Host cost of an eager step() call of the 1D envs through both bindings of the C ABI (torch.library ops / ctypes): wall time
per call over a long loop (no events inside) and the time the host needs to ISSUE the calls, with the kernel's own duration
(HIP graph replay) next to it; cProfile of the Python side on request.
  PYTHONPATH=. python scripts/host_cost.py [profile]"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())

import torch

from beacon_amd import vec as V

dev = "cuda:0"
for name, mk, shape in (("burgers", lambda: V.VecBurgers(1024, dev, "f32", nx=512), (1024,)),
                        ("sloshing", lambda: V.VecSloshing(1024, dev, "f32"), (1024,)),
                        ("shkadov", lambda: V.VecShkadov(1024, dev, "f32", None, L0=699.2, n_jets=10), (1024, 10))):
    env = mk()
    env.reset()
    a = torch.zeros(shape, dtype=torch.float32, device=dev).uniform_(-1, 1)
    N = 3000 if name != "shkadov" else 300
    for binding in ("torch ops", "ctypes"):             # the two bindings of the same C ABI (beacon_amd/torch_ext.py, _lib.py)
        if env.use_torch_ops(binding == "torch ops") != (binding == "torch ops"):
            print("%-9s %s: not available" % (name, binding))
            continue
        for _ in range(50):
            env.step(a)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(N):
            env.step(a)
        t_issue = time.perf_counter() - t
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t
        print("%-9s %-9s eager %.2f us per call, host issue %.2f us per call" % (name, binding, 1e6 * t_all / N, 1e6 * t_issue / N), flush=True)
    env.use_torch_ops(True)
    g = env.capture(a.unsqueeze(0).expand(16, *a.shape).contiguous(), None, n_steps=16, keep_steps=False)
    g.replay(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    t_graph = (time.perf_counter() - t) / (20 * 16)
    print("%-9s in a HIP graph %.2f us per step" % (name, 1e6 * t_graph), flush=True)
    if len(sys.argv) > 1 and name == "burgers":
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(2000):
            env.step(a)
        pr.disable()
        torch.cuda.synchronize()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
    env.close()
