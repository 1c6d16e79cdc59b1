"""Adversarial right-hand side for the stop rule of the register-resident kernels -- TEST INFRASTRUCTURE.

The reference stops its Jacobi solve at the first sweep whose residual norm err = sum over the whole array of (phi - phin)^2
is <= tol (rayleigh.py:448-454).  That norm can GROW from one sweep to a later one (scripts/weighted_norm_bound.py: by up to
1.030), so "the first evaluation behind skipped sweeps fails" does not by itself prove that no skipped sweep passed.  This
script builds a rayleigh state (50x50, Ra = 1e12 so that the predictor's diffusion is negligible, T = p = 0) whose first
Poisson solve does exactly that: err_1 .. err_{J0-1} > tol, err_J0 = 0.992 tol (the reference stops HERE), and the next
sweeps sit at up to 1.016 tol before the decay resumes (sweep 17 passes again): an evaluation plan that skips sweep J0 and lands
in that stretch finds the residual ABOVE tol and, unless it knows the bound, carries on to a stop twelve sweeps late.  Output: tests/golden/rayleigh_adversarial_50x50.npz (state, expected
stop sweep, the residual sequence).  Run from the repo root:  python oracle/make_adversarial.py"""
import ctypes as C
import os
import sys

import numpy as np
from scipy.fft import dctn, idctn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from oracle import oracle as O                      # noqa: E402
from weighted_norm_bound import growth, operators   # noqa: E402

NX = NY = 50
TOL, GUARD = 1.0e-8, 1.035
J0, M = 5, 4            # the packet's first passing sweep, and the lag of its maximal growth


def lam(nx, ny):
    p, q = np.arange(nx)[:, None], np.arange(ny)[None, :]
    return 0.5 * (np.cos(np.pi * p / nx) + np.cos(np.pi * q / ny))


def residuals(d1, n):
    """err_k = d_k' W d_k for k = 1 .. n, d_{k+1} = J d_k (the reference's norm, ghost copies included)."""
    J, sw, _ = operators(NX, NY, 0)
    out, d = [], d1.copy()
    for _ in range(n):
        out.append(float(((sw * d) ** 2).sum()))
        d = J(d)
    return np.array(out)


def plan2_pairs(err, tol, tol_l, overshoot=0):
    """The kernels' extrapolating plan on a residual sequence (ns2d_fast_impl.h, evaluations in pairs; `overshoot`: the
    plan_overshoot test hook, which lengthens every skip): (stop sweep or None, sweeps evaluated, landing flagged unverified)."""
    itp, k_prev, evaluated = 0, -1, []
    while itp + 2 <= len(err):
        e1, e2, itp0 = err[itp], err[itp + 1], itp
        evaluated += [itp + 1, itp + 2]
        itp += 2
        landing = itp0 > 0 and k_prev != itp0
        unv = landing and not (e1 > tol_l)
        if e1 <= tol or e2 <= tol or unv:
            return (itp0 + 1 if e1 <= tol else (itp0 + 2 if e2 <= tol else None)), evaluated, unv
        room, rho = np.log2(e2) - np.log2(tol_l * 1.003), np.log2(e2) - np.log2(e1)
        jw = 0
        if room > 0:
            jw = int(min(room / -rho, 256.0)) if rho < 0 else 256
        j = max(jw - 1 - (jw >> 4) + overshoot, 0) & ~1
        k_prev = itp
        itp += j
    return None, evaluated, False


def main():
    lm = lam(NX, NY)
    c_m, v = growth(NX, NY, 0, M, vectors=True)          # err(v) = 1, err(J^M v) = c_m
    # the packet M sweeps in front of its maximum, planted at sweep J0: d_1 = J^-(J0-1) v on the modes that survive the way
    vh = dctn(v, type=2, norm="ortho")
    keep = np.abs(lm) > 0.2
    d1p = idctn(np.where(keep, vh / np.where(keep, lm, 1.0) ** (J0 - 1), 0.0), type=2, norm="ortho")
    e = residuals(d1p, J0 + 12)
    d1p *= np.sqrt(0.992 * TOL / e[J0 - 1])               # err_J0 = 0.992 tol
    d1 = d1p
    e = residuals(d1, 320)
    first = int(np.argmax(e <= TOL)) + 1
    assert first == J0 and (e[J0:J0 + 6] > TOL).all(), e[:12] / TOL
    print("err / tol for sweeps 1..20: %s" % np.round(e[:20] / TOL, 4).tolist())
    again = J0 + 1 + int(np.argmax(e[J0:] <= TOL))
    print("reference stops at sweep %d; the residual is above tol again for sweeps %d..%d" % (J0, J0 + 1, again - 1))
    table = []
    for ov in range(0, 13):
        s2, ev2, _ = plan2_pairs(e, TOL, TOL, ov)
        s3, ev3, unv3 = plan2_pairs(e, TOL, TOL * GUARD, ov)
        table.append((ov, -1 if s2 is None else s2, int(unv3), -1 if s3 is None else s3))
        print("plan_overshoot %2d: unguarded plan evaluates %s -> stops at %s | guarded plan evaluates %s -> %s"
              % (ov, ev2[:10], s2, ev3[:10], "unverified landing: repeated under the proven plan" if unv3 else "stops at %s" % s3))
    stop2 = max(t[1] for t in table)
    # d_1 = phi_1 = -b dx^2 / 4, b = div(u*) / dt  ->  u* = grad(psi), laplace(psi) = div
    cfg = O.rayleigh(init=False, ra=1.0e12).cfg
    dx, dt = cfg.dx, cfg.dt
    div = -4.0 * d1 / (dx * dx) * dt
    pp, qq = np.arange(NX)[:, None], np.arange(NY)[None, :]
    lap = (2 * np.cos(np.pi * pp / NX) - 2) / dx ** 2 + (2 * np.cos(np.pi * qq / NY) - 2) / dx ** 2
    lap[0, 0] = 1.0
    ph = dctn(div, type=2, norm="ortho") / lap
    ph[0, 0] = 0.0
    psi = idctn(ph, type=2, norm="ortho")
    us_t, vs_t = np.zeros((NX + 2, NY + 2)), np.zeros((NX + 2, NY + 2))
    us_t[2:NX + 1, 1:NY + 1] = (psi[1:, :] - psi[:-1, :]) / dx
    vs_t[1:NX + 1, 2:NY + 1] = (psi[:, 1:] - psi[:, :-1]) / dx
    # u, v with predictor(u, v) = (u*, v*): fixed point through the oracle's own boundary conditions + predictor (p = T = 0)
    env = O.rayleigh(init=False, ra=1.0e12)
    L = O.lib()
    st = env.st
    st[0], st[1] = us_t, vs_t
    a0 = np.zeros(10)
    for it in range(30):
        w = st.copy()
        L.orc_ns2d_bc(C.byref(env.cfg), O.dp(w[0]), O.dp(w[1]), O.dp(w[3]), O.dp(a0), C.c_double(0), C.c_double(0), C.c_double(0), C.c_double(0))
        w[3] = 0.0      # (T stays zero: no buoyancy; the BCs only set its ghosts)
        L.orc_ns2d_predictor(C.byref(env.cfg), O.dp(w[0]), O.dp(w[1]), O.dp(w[4]), O.dp(w[5]), O.dp(w[2]), O.dp(w[3]))
        ru, rv = us_t - w[4], vs_t - w[5]
        ru[:2] = 0; ru[NX + 1:] = 0; ru[:, 0] = 0; ru[:, NY + 1] = 0
        rv[:, :2] = 0; rv[:, NY + 1:] = 0; rv[0] = 0; rv[NX + 1] = 0
        st[0] += ru
        st[1] += rv
        r = max(np.abs(ru).max(), np.abs(rv).max())
        if r < 1e-17:
            break
    print("predictor inverted in %d iterations (residual %.1e); max |u| %.2e" % (it + 1, r, np.abs(st[0]).max()))
    state = st[:4].copy()
    # what the oracle makes of it: ONE timestep
    env2 = O.rayleigh(init=False, ra=1.0e12)
    env2.cfg.ndt_act = 1
    env2.st[:4] = state
    env2.itp = np.zeros(1, dtype=np.int32)
    env2.solve([0.0] * 10)
    print("oracle: first solve stops at sweep", int(env2.itp[0]))
    assert int(env2.itp[0]) == J0
    out = os.path.join(ROOT, "tests", "golden", "rayleigh_adversarial_50x50.npz")
    np.savez_compressed(out, state=state, ra=1.0e12, stop_sweep=J0, second_crossing=again, err_over_tol=e[:24] / TOL,
                        growth=c_m, lag=M, plan_table=np.array(table), final_state=env2.st[:4].copy())
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
