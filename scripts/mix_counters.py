# diagnostic: mixing B=512 -- share of the shader cycles inside the Jacobi loop and cycles per sweep (bcn_get_counters)
# usage: python scripts/mix_counters.py [dtype] [opt=value ...]
import os, sys, numpy as np, torch, time
sys.path.insert(0, os.getcwd())
from beacon_amd import vec as V
env = V.VecMixing(512, "cuda:0", sys.argv[1] if len(sys.argv) > 1 else "f32")
for a in sys.argv[2:]:
    k, v = a.split("="); env.set_option(k, int(v))
env.reset()
rng = np.random.default_rng(7)
for k in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    env.step(rng.integers(0, 4, 512)); torch.cuda.synchronize(); t1 = time.perf_counter()
    c = env.get_counters().astype(np.float64); sw = env.sweeps.cpu().numpy()
    if k >= 4:
        print(sys.argv[2:], k, "ms %.2f" % ((t1 - t0) * 1e3), "sweeps/dt %.1f" % sw.mean(), "jacobi share %.3f" % (c[:, 0].sum() / c[:, 1].sum()),
              "cycles/sweep %.0f" % (c[:, 0].sum() / sw.sum()), "cycles/timestep outside the solve %.0f" % ((c[:, 1].sum() - c[:, 0].sum()) / sw.size),
              "late %d repeats %d" % (c[:, 2].sum(), c[:, 3].sum()), env.kernel_name)
