"""Tall grids (ny > 128): ns2d_fast4_step against ns2d_generic_step on the same workload -- ms per action step, Jacobi
sweeps per timestep, shader cycles per sweep and outside the Poisson solve (bcn_get_counters).
  PYTHONPATH=. [BEACON_JIT_DEFS=BCN_F4_STAMP] python scripts/tall_base.py [B [name filter]]      (the flag: cycles per phase)"""
import os
import sys
import time

import torch

from beacon_amd.vec import VecMixing, VecRayleigh


def run(env, acts, n=2):
    g = torch.Generator(device="cuda").manual_seed(1)

    def draw():                      # integer actions (mixing): a new random wall motion every step, as bench.py does
        return torch.randint(0, 4, acts.shape, generator=g, device="cuda", dtype=torch.int32) if acts.dtype == torch.int32 else acts
    env.reset()
    for _ in range(3):
        env.step(draw())
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(n):
        env.step(draw())
    torch.cuda.synchronize()
    ms = (time.time() - t) / n * 1e3
    sw = env.sweeps.double().cpu().numpy()
    c = env.get_counters().astype("float64")
    if "BCN_F4_STAMP" in os.environ.get("BEACON_JIT_DEFS", "") and env.kernel_name == "ns2d_fast4_step":
        names = ["bc", "predictor", "plain sweeps", "rhs+load+evaluated sweeps", "phi+corrector", "transport coeff", "chain", "copy out"]
        print("   cycles/dt: " + "  ".join("%s %.0f" % (names[k], c[k::8, 2].mean() / sw.shape[1]) for k in range(8)))
        print("   evaluated sweeps/dt %.1f" % (c[:, 3].mean() / sw.shape[1]))
    return ms, float(sw.mean()), float(c[:, 0].mean() / sw.sum(1).mean()), float((c[:, 1] - c[:, 0]).mean() / sw.shape[1])


B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ONLY = sys.argv[2] if len(sys.argv) > 2 else ""
for name, mk, act in (("mixing 100x200", lambda: VecMixing(B, dtype=torch.float32, L=1.0, H=2.0), lambda e: (torch.arange(B, dtype=torch.int32, device="cuda") % 4)),
                      ("rayleigh 50x150", lambda: VecRayleigh(B, dtype=torch.float32, L=1.0, H=3.0), lambda e: torch.zeros(B, e.n_sgts, dtype=torch.float32, device="cuda")),
                      ("rayleigh 50x150 f64", lambda: VecRayleigh(B, dtype=torch.float64, L=1.0, H=3.0), lambda e: torch.zeros(B, e.n_sgts, dtype=torch.float64, device="cuda")),
                      ("mixing 200x100", lambda: VecMixing(B, dtype=torch.float32, L=2.0, H=1.0), lambda e: (torch.arange(B, dtype=torch.int32, device="cuda") % 4)),
                      ("rayleigh 300x50", lambda: VecRayleigh(B, dtype=torch.float32, L=6.0, H=1.0), lambda e: torch.zeros(B, e.n_sgts, dtype=torch.float32, device="cuda")),
                      ("rayleigh 110x64 f64", lambda: VecRayleigh(B, dtype=torch.float64, L=2.2, H=1.28), lambda e: torch.zeros(B, e.n_sgts, dtype=torch.float64, device="cuda")),
                      ("mixing 100x100", lambda: VecMixing(B, dtype=torch.float32), lambda e: torch.zeros(B, dtype=torch.int32, device="cuda"))):
    if ONLY not in name:
        continue
    for variant in (1, 0) if not ONLY else (1,):
        env = mk()
        env.set_variant(variant)
        ms, sw, cps, outside = run(env, act(env))
        print("%-20s B=%d %-18s %8.2f ms  sweeps/dt %6.1f  cycles/sweep %6.0f  outside the solve %7.0f cycles/dt" % (name, B, env.kernel_name, ms, sw, cps, outside), flush=True)
        env.close()
