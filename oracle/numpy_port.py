"""Vectorised-NumPy restatement of the rayleigh solver path -- TEST / BENCH INFRASTRUCTURE ONLY.

This is the "NumPy CPU" leg of BASELINE.json's metric: the reference's per-cell
numba loops (rayleigh/rayleigh.py:370-487) written as whole-array NumPy expressions, one env at a time,
float64, same operation order per cell, so fields are bit-identical to the reference:
  * predictor / Poisson / corrector: array slices (every cell of those loops reads only old values);
  * transport (rayleigh.py:468-487) is an in-place sequential sweep (cell (i,j) reads the already updated
    (i-1,j) and (i,j-1)); cells of one anti-diagonal i+j = d are independent, so the sweep runs diagonal by
    diagonal with gathered index vectors -- the same per-cell expression, the same result bit for bit
    (SURVEY.md 7.3-1).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; it is checked
against the reference-captured golden vectors in tests/test_oracle.py."""
import math

import numpy as np


def condition_actions(a, C):
    """rayleigh.py:162-171: zero-mean, then scaled into [-C, C]."""
    a = np.array(a, dtype=np.float64)
    a -= np.mean(a)
    a /= max(1.0, np.max(np.abs(a)) / C)
    return a


class Rayleigh(object):
    """State and one-timestep / one-action-step updates of rayleigh/rayleigh.py:16-366 (solver part)."""

    def __init__(self, init_fields=None, L=1.0, H=1.0, n_sgts=10, ra=1.0e4):
        self.nx, self.ny = int(50 * L), int(50 * H)
        self.pr, self.ra, self.Tc, self.Th, self.C = 0.71, ra, -0.5, 0.5, 0.75
        self.dt, self.dx, self.dy = 0.01, float(L / self.nx), float(H / self.ny)
        self.ndt_act = int(2.0 / self.dt)
        self.n_sgts, self.nx_sgts = n_sgts, self.nx // n_sgts
        nx, ny = self.nx, self.ny
        self.u, self.v, self.p, self.T = (np.zeros((nx + 2, ny + 2)) for _ in range(4))
        if init_fields is not None:
            self.u[:], self.v[:], self.p[:], self.T[:] = np.asarray(init_fields, dtype=np.float64)
        self.us, self.vs, self.phi = (np.zeros((nx + 2, ny + 2)) for _ in range(3))
        # anti-diagonals of the interior: index vectors (i, j) with i + j = d, i ascending
        self.diag = []
        for d in range(2, nx + ny + 1):
            i = np.arange(max(1, d - ny), min(nx, d - 1) + 1)
            self.diag.append((i, d - i))
        self.itp = []

    # -- rayleigh.py:180-202 -----------------------------------------------------------------
    def bcs(self, a):
        u, v, T, nx = self.u, self.v, self.T, self.nx
        u[1, 1:-1] = 0.0
        u[-1, 1:-1] = 0.0
        v[0, 2:-1] = -v[1, 2:-1]
        v[-1, 2:-1] = -v[-2, 2:-1]
        T[0, 1:-1] = T[1, 1:-1]
        T[-1, 1:-1] = T[-2, 1:-1]
        u[1:, -1] = -u[1:, -2]
        v[1:-1, -1] = 0.0
        T[1:-1, -1] = 2.0 * self.Tc - T[1:-1, -2]
        u[1:, 0] = -u[1:, 1]
        v[1:-1, 1] = 0.0
        for j in range(self.n_sgts):
            s = 1 + j * self.nx_sgts
            e = s + self.nx_sgts
            T[s:e, 0] = 2.0 * (self.Th + a[j]) - T[s:e, 1]

    # -- rayleigh.py:370-407 -----------------------------------------------------------------
    def predictor(self):
        u, v, p, T, us, vs = self.u, self.v, self.p, self.T, self.us, self.vs
        dx, dy, dt = self.dx, self.dy, self.dt
        k = math.sqrt(self.pr / self.ra)
        c = u[2:-1, 1:-1]
        uE = 0.5 * (u[3:, 1:-1] + c)
        uW = 0.5 * (c + u[1:-2, 1:-1])
        uN = 0.5 * (u[2:-1, 2:] + c)
        uS = 0.5 * (c + u[2:-1, :-2])
        vN = 0.5 * (v[2:-1, 2:] + v[1:-2, 2:])
        vS = 0.5 * (v[2:-1, 1:-1] + v[1:-2, 1:-1])
        conv = (uE * uE - uW * uW) / dx + (uN * vN - uS * vS) / dy
        diff = ((u[3:, 1:-1] - 2.0 * c + u[1:-2, 1:-1]) / (dx ** 2) +
                (u[2:-1, 2:] - 2.0 * c + u[2:-1, :-2]) / (dy ** 2))
        diff *= k
        pres = (p[2:-1, 1:-1] - p[1:-2, 1:-1]) / dx
        us[2:-1, 1:-1] = c + dt * (diff - conv - pres)
        c = v[1:-1, 2:-1]
        vE = 0.5 * (v[2:, 2:-1] + c)
        vW = 0.5 * (c + v[:-2, 2:-1])
        uE = 0.5 * (u[2:, 2:-1] + u[2:, 1:-2])
        uW = 0.5 * (u[1:-1, 2:-1] + u[1:-1, 1:-2])
        vN = 0.5 * (v[1:-1, 3:] + c)
        vS = 0.5 * (c + v[1:-1, 1:-2])
        conv = (uE * vE - uW * vW) / dx + (vN * vN - vS * vS) / dy
        diff = ((v[2:, 2:-1] - 2.0 * c + v[:-2, 2:-1]) / (dx ** 2) +
                (v[1:-1, 3:] - 2.0 * c + v[1:-1, 1:-2]) / (dy ** 2))
        diff *= k
        pres = (p[1:-1, 2:-1] - p[1:-1, 1:-2]) / dy
        vs[1:-1, 2:-1] = c + dt * (diff - conv - pres + T[1:-1, 2:-1])

    # -- rayleigh.py:411-456 -----------------------------------------------------------------
    def poisson(self, tol=1.0e-8, itmax=300000):
        us, vs, phi = self.us, self.vs, self.phi
        dx, dy, dt = self.dx, self.dy, self.dt
        b = ((us[2:, 1:-1] - us[1:-1, 1:-1]) / dx + (vs[1:-1, 2:] - vs[1:-1, 1:-1]) / dy) / dt
        phi[:, :] = 0.0
        phin = np.zeros_like(phi)
        itp, err = 0, 1.0e10
        while err > tol:
            phin[:, :] = phi
            phi[1:-1, 1:-1] = 0.5 * ((phin[2:, 1:-1] + phin[:-2, 1:-1]) * dy * dy +
                                     (phin[1:-1, 2:] + phin[1:-1, :-2]) * dx * dx -
                                     b * dx * dx * dy * dy) / (dx * dx + dy * dy)
            phi[0, 1:-1] = phi[1, 1:-1]
            phi[-1, 1:-1] = phi[-2, 1:-1]
            phi[1:-1, -1] = phi[1:-1, -2]
            phi[1:-1, 0] = phi[1:-1, 1]
            dphi = np.reshape(phi - phin, (-1))
            err = np.dot(dphi, dphi)
            itp += 1
            if itp > itmax:
                raise RuntimeError("Exceeded max number of iterations in solver")
        return itp

    # -- rayleigh.py:460-464 -----------------------------------------------------------------
    def corrector(self):
        u, v, us, vs, phi, dt = self.u, self.v, self.us, self.vs, self.phi, self.dt
        u[2:-1, 1:-1] = us[2:-1, 1:-1] - dt * (phi[2:-1, 1:-1] - phi[1:-2, 1:-1]) / self.dx
        v[1:-1, 2:-1] = vs[1:-1, 2:-1] - dt * (phi[1:-1, 2:-1] - phi[1:-1, 1:-2]) / self.dy

    # -- rayleigh.py:468-487 -----------------------------------------------------------------
    def transport(self):
        u, v, T = self.u, self.v, self.T
        dx, dy, dt = self.dx, self.dy, self.dt
        k = math.sqrt(self.pr * self.ra)
        for i, j in self.diag:
            Tc = T[i, j]
            TE = 0.5 * (T[i + 1, j] + Tc)
            TW = 0.5 * (T[i - 1, j] + Tc)
            TN = 0.5 * (T[i, j + 1] + Tc)
            TS = 0.5 * (T[i, j - 1] + Tc)
            conv = (u[i + 1, j] * TE - u[i, j] * TW) / dx + (v[i, j + 1] * TN - v[i, j] * TS) / dy
            diff = ((T[i + 1, j] - 2.0 * Tc + T[i - 1, j]) / (dx ** 2) +
                    (T[i, j + 1] - 2.0 * Tc + T[i, j - 1]) / (dy ** 2))
            diff /= k
            T[i, j] = Tc + dt * (diff - conv)

    def timestep(self, a):
        """One solver timestep with the conditioned action vector `a` (rayleigh.py:177-231); returns the
        number of Jacobi sweeps."""
        self.bcs(a)
        self.predictor()
        itp = self.poisson()
        self.p += self.phi
        self.corrector()
        self.transport()
        return itp

    def solve(self, a):
        a = condition_actions(a, self.C)
        self.itp = [self.timestep(a) for _ in range(self.ndt_act)]
        return a
