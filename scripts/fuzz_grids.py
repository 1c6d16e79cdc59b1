"""Random domain sizes (the reference takes any L, H: rayleigh.py:20-27, mixing.py:20-28): every grid gets its kernel
plugin compiled on the spot (beacon_amd/jit.py) and must agree with the generic kernel on the same inputs after an action
step of 12 timesteps -- fields to the float32 / float64 tolerance of the on-demand grid tests, sweep counts within 3 / 1.
  PYTHONPATH=. python scripts/fuzz_grids.py [n] [seed]"""
import sys

import numpy as np
import torch

from beacon_amd import jit
from beacon_amd import vec as V

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for t in range(n):
    kind = int(rng.integers(0, 2))
    f64 = bool(rng.integers(0, 4) == 0)
    if kind == 0:
        L, H = round(float(rng.uniform(1.0, 4.5)), 2), round(float(rng.uniform(1.0, 3.2)), 2)   # (below 1 the reference divides by zero)
        nx, ny = int(50 * L), int(50 * H)
    else:
        L, H = round(float(rng.uniform(1.0, 1.7)), 2), round(float(rng.uniform(1.0, 2.3)), 2)
        nx, ny = int(100 * L), int(100 * H)
    m = jit.choose(nx, ny, f64, kind)
    if m is None or nx * ny > 30000:
        print("skip %s %dx%d %s (no mapping)" % ("rayleigh" if kind == 0 else "mixing", nx, ny, "f64" if f64 else "f32"))
        continue
    dt = "f64" if f64 else "f32"
    out = {}
    for variant in (1, 0):
        env = V.VecRayleigh(3, "cuda:0", dt, None, L=L, H=H) if kind == 0 else V.VecMixing(3, "cuda:0", dt, L=L, H=H)
        env.set_ndt_act(12)
        got = env.set_variant(variant)
        env.reset()
        if kind == 0:
            x, y = (np.arange(env.nx + 2) - 0.5) / env.nx, (np.arange(env.ny + 2) - 0.5) / env.ny
            st0 = np.zeros((4, env.nx + 2, env.ny + 2))
            st0[3] = (0.5 - y)[None, :] + 0.08 * np.sin(2 * np.pi * x * L)[:, None] * np.sin(np.pi * y)[None, :]
            env.set_state(np.tile(np.ascontiguousarray(st0.transpose(0, 2, 1))[None], (3, 1, 1, 1)))
            a = np.random.default_rng(t).uniform(-1, 1, (3, env.n_sgts))
        else:
            a = np.array([0, 2, 3])
        env.step(a)
        env.check_status()
        out[variant] = (env.get_state().double().cpu().numpy(), env.sweeps.cpu().numpy(), env.kernel_name, got)
        env.close()
    tol = 1e-9 if f64 else 2e-4
    d = np.abs(out[1][0] - out[0][0])
    dp = d[:, 2].max()
    dr = np.delete(d, 2, axis=1).max()
    ds = np.abs(out[1][1] - out[0][1]).max()
    ok = dr <= tol and dp <= 50 * tol and ds <= max(3, 0.02 * out[0][1].max()) and out[1][3] == 1
    bad += not ok
    print("%s %-8s %3dx%-3d %s rows=%d R=%d: %s  |d| fields %.2e p %.2e sweeps %d (max %d)"
          % ("ok " if ok else "BAD", "rayleigh" if kind == 0 else "mixing", nx, ny, dt, m["rows"], m["R"], out[1][2], dr, dp, ds, out[0][1].max()), flush=True)
print("bad:", bad)
sys.exit(1 if bad else 0)
