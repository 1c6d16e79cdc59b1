#!/usr/bin/env python3
"""Developed-flow initial state for grids the reference ships no init_field.dat for
(BASELINE's rayleigh 128x64: L=2.56, H=1.28).  Equivalent of beacon/rayleigh/init.py
(n_sgts=1, zero action, n_warmup=100 action steps) run with the float64 CPU oracle,
started from the conduction profile plus a seeded perturbation (from the reference's
all-zero start an exactly x-uniform state can never leave pure conduction).

TEST/BENCH INFRASTRUCTURE: writes tests/golden/rayleigh_128x64_init.npz, which bench.py
and the full-size tests load as synthetic input.   usage: python oracle/make_init.py [steps]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402


def start_state(L=2.56, H=1.28, seed=2024):
    """(oracle env, T[nx+2, ny+2]): conduction profile plus five seeded long-wave perturbations, T ghosts rows zeroed."""
    e = O.rayleigh(init=False, L=L, H=H, n_sgts=1)
    e.reset_fields()
    c = e.cfg
    rng = np.random.default_rng(seed)
    xm = (np.arange(c.nx + 2) - 0.5) * c.dx
    ym = (np.arange(c.ny + 2) - 0.5) * c.dy
    X, Y = np.meshgrid(xm, ym, indexing="ij")
    T = c.Th + (c.Tc - c.Th) * Y / H
    for k in range(1, 6):
        T += 0.02 * rng.uniform(-1, 1) * np.sin(np.pi * Y / H) * np.cos(k * np.pi * X / L + rng.uniform(0, 6.28))
    T[:, 0] = 0.0
    T[:, -1] = 0.0
    return e, T


def main(n_steps=100, L=2.56, H=1.28, seed=2024):
    e, T = start_state(L, H, seed)
    e.S[:] = T
    t0 = time.time()
    nus = []
    for k in range(n_steps):
        o, r, d, t, _ = e.step([0.0])
        nus.append(-r)
        if k % 10 == 0:
            print("step %d  Nu %.5f  sweeps/dt %.1f  (%.0fs)" % (k, -r, e.itp.mean(), time.time() - t0), flush=True)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                       "rayleigh_128x64_init.npz")
    np.savez_compressed(out, fields=e.st[:4].copy(), nu=np.array(nus), L=L, H=H, seed=seed, n_steps=n_steps)
    print("wrote", out)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 100)
