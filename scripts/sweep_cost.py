"""Cost of a PLAIN Jacobi sweep of ns2d_fast4_impl.h in isolation: tol = 0 and a small itmax, so that every solve runs
itmax sweeps, all but two of them without residual (the replica then stops with BCN_ST_ITMAX: timing only).
  PYTHONPATH=. python scripts/sweep_cost.py [B]
(Round 4 also built it without its barrier / without its edge exchange -- wrong results, timing only: the sweep is
issue-bound, docs/history/DESIGN_rounds_1-4.md 4.2c; those switches were removed in round 5.)"""
import sys

import torch

from beacon_amd.vec import VecMixing, VecRayleigh

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ITMAX = 400


def patched(cls):
    class P(cls):
        def _derive(self, *a, **k):
            super()._derive(*a, **k)
            self.tol, self.itmax = 0.0, ITMAX
            return self
    return P


for name, mk, act in (("mixing 100x200", lambda: patched(VecMixing)(B, dtype=torch.float32, L=1.0, H=2.0), lambda e: torch.zeros(B, dtype=torch.int32, device="cuda")),
                      ("rayleigh 50x150", lambda: patched(VecRayleigh)(B, dtype=torch.float32, L=1.0, H=3.0), lambda e: torch.zeros(B, e.n_sgts, dtype=torch.float32, device="cuda")),
                      ("rayleigh 50x150 f64", lambda: patched(VecRayleigh)(B, dtype=torch.float64, L=1.0, H=3.0), lambda e: torch.zeros(B, e.n_sgts, dtype=torch.float64, device="cuda"))):
    env = mk()
    env.reset()
    env.step(act(env))
    torch.cuda.synchronize()
    sw = env.sweeps.double().cpu().numpy()
    c = env.get_counters().astype("float64")
    print("%-20s %-16s sweeps per solve %.0f   cycles per sweep %.0f" % (name, env.kernel_name, sw[:, 0].mean(), c[:, 0].mean() / sw[:, 0].mean()), flush=True)
    env.close()
