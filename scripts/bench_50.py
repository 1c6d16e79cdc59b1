"""rayleigh 50x50 (the reference's default grid), B = 512: step time of the register-resident kernel for
different strip widths (-DBCN_R50=...), float32 and float64."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
from beacon_amd import build
extra = sys.argv[1:]
build.FLAGS.extend(extra); build.build_lib(force=True)
from beacon_amd import vec as V
from beacon_amd.envs import packaged_init
B = 512
acts = np.random.default_rng(3).uniform(-1, 1, (4, B, 10))
for dt in ("f32", "f64"):
    env = V.VecRayleigh(B, "cuda:0", dt, packaged_init("rayleigh"))
    env.reset(); env.step(acts[0]); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(1, 4): env.step(acts[k])
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 3 * 1e3
    print(extra, dt, env.kernel_name, "%.2f ms/step  %.0f env steps/s  sweeps/dt %.1f" % (ms, B / ms * 1e3, env.sweeps.float().mean().item()))
    env.close()
