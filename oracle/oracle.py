"""ctypes front-end of the CPU oracle (oracle/beacon_oracle.c) + the reference's
env-level logic (reset/step/episode bookkeeping) restated around it.

TEST INFRASTRUCTURE ONLY.  May be imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never by the product package beacon_amd/.
Parity status: PINNED by tests/test_oracle.py against tests/golden/*.npz, which
were captured from the unmodified reference (oracle/capture/capture.py).

Each class mirrors one reference env (same ctor kwargs, reset()/step() return
shapes and dtypes); citations are file:line into /root/reference/beacon/.
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_dp = C.POINTER(C.c_double)


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "beacon_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        if os.environ.get("BEACON_NO_BUILD") == "1":   # profiled processes must not spawn compilers
            raise RuntimeError("oracle/liboracle.so is missing or stale and BEACON_NO_BUILD=1")
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_ns2d_rwd.restype = C.c_double
        L.orc_burgers_obs_rwd.restype = C.c_double
        L.orc_shkadov_obs_rwd.restype = C.c_double
        L.orc_sloshing_obs_rwd.restype = C.c_double
        _LIB = L
    return _LIB


def dp(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_dp)


class ns2d_cfg(C.Structure):
    _fields_ = [("kind", C.c_int32), ("nx", C.c_int32), ("ny", C.c_int32), ("ndt_act", C.c_int32),
                ("n_sgts", C.c_int32), ("nx_sgts", C.c_int32),
                ("nx_obs_pts", C.c_int32), ("ny_obs_pts", C.c_int32), ("nx_obs", C.c_int32),
                ("ny_obs", C.c_int32), ("n_obs_steps", C.c_int32), ("itmax", C.c_int32),
                ("dx", C.c_double), ("dy", C.c_double), ("dt", C.c_double),
                ("pr", C.c_double), ("ra", C.c_double), ("Tc", C.c_double), ("Th", C.c_double),
                ("C", C.c_double),
                ("re", C.c_double), ("pe", C.c_double), ("u_max", C.c_double), ("ref_c", C.c_double),
                ("tol", C.c_double)]


class burgers_cfg(C.Structure):
    _fields_ = [("nx", C.c_int32), ("ndt_act", C.c_int32), ("ctrl_pos", C.c_int32), ("n_obs_pts", C.c_int32),
                ("dx", C.c_double), ("dt", C.c_double), ("amp", C.c_double), ("u_target", C.c_double)]


class shkadov_cfg(C.Structure):
    _fields_ = [("nx", C.c_int32), ("ndt_act", C.c_int32), ("n_jets", C.c_int32), ("jet_pos", C.c_int32),
                ("jet_hw", C.c_int32), ("jet_space", C.c_int32), ("l_obs", C.c_int32), ("l_rwd", C.c_int32),
                ("n_obs", C.c_int32), ("n_interp", C.c_int32), ("obs_stride", C.c_int32),
                ("dx", C.c_double), ("dt", C.c_double), ("delta", C.c_double), ("jet_amp", C.c_double),
                ("eps", C.c_double)]


class sloshing_cfg(C.Structure):
    _fields_ = [("nx", C.c_int32), ("ndt_act", C.c_int32), ("n_interp", C.c_int32),
                ("dx", C.c_double), ("dt", C.c_double), ("g", C.c_double), ("amp", C.c_double),
                ("alpha", C.c_double)]


# ---------------------------------------------------------------------------
# 2D envs
# ---------------------------------------------------------------------------
class _ns2d(object):
    """Shared body of rayleigh / mixing.  Fields live in self.st[8, nx+2, ny+2] =
    (u, v, p, S, us, vs, phi, phin) in the reference's [i, j] layout."""

    def _alloc(self):
        c = self.cfg
        self.st = np.zeros((8, c.nx + 2, c.ny + 2))
        self.u, self.v, self.p, self.S = self.st[0], self.st[1], self.st[2], self.st[3]
        self.us, self.vs, self.phi = self.st[4], self.st[5], self.st[6]
        self.obs = np.zeros((c.n_obs_steps, 3, c.nx_obs_pts, c.ny_obs_pts))
        self.n_obs_tot = self.obs.size
        self.itp = np.zeros(c.ndt_act, dtype=np.int32)
        self.stp = 0

    def get_obs(self):
        lib().orc_ns2d_obs(C.byref(self.cfg), dp(self.u), dp(self.v), dp(self.S), dp(self.obs))
        return np.reshape(self.obs, [-1])

    def get_rwd(self):
        return lib().orc_ns2d_rwd(C.byref(self.cfg), dp(self.S))

    def _solve(self, avec):
        self.itp = np.zeros(self.cfg.ndt_act, dtype=np.int32)
        rc = lib().orc_ns2d_solve(C.byref(self.cfg), dp(self.st.reshape(-1)), dp(avec),
                                  self.itp.ctypes.data_as(C.POINTER(C.c_int32)))
        if rc:
            raise RuntimeError("Exceeded max number of iterations in solver")

    def _finish_step(self):
        obs = self.get_obs()
        rwd = self.get_rwd()
        done = trunc = (self.stp == self.n_act - 1)
        self.stp += 1
        return obs, rwd, done, trunc, None


class rayleigh(_ns2d):
    """rayleigh/rayleigh.py:16-366"""

    def __init__(self, cpu=0, init=True, L=1.0, H=1.0, n_sgts=10, ra=1.0e4, init_fields=None):
        self.L, self.H = L, H
        nx, ny = int(50 * L), int(50 * H)
        self.nx, self.ny = nx, ny
        dt, dt_act, t_act = 0.01, 2.0, 200.0
        self.n_act = int(t_act / dt_act)
        self.n_warmup = int(200.0 / dt_act)
        self.n_sgts = n_sgts
        nxo, nyo = 4 * int(L), 4 * int(H)
        self.cfg = ns2d_cfg(kind=0, nx=nx, ny=ny, ndt_act=int(dt_act / dt), n_sgts=n_sgts,
                            nx_sgts=nx // n_sgts, nx_obs_pts=nxo, ny_obs_pts=nyo, nx_obs=nx // nxo,
                            ny_obs=ny // nyo, n_obs_steps=4, itmax=300000,
                            dx=float(L / nx), dy=float(H / ny), dt=dt, pr=0.71, ra=ra, Tc=-0.5, Th=0.5,
                            C=0.75, re=0.0, pe=0.0, u_max=0.0, ref_c=0.0, tol=1.0e-8)
        self._alloc()
        self.init = np.zeros((4, nx + 2, ny + 2))
        if init:
            if init_fields is None:
                raise ValueError("init=True needs init_fields[4, nx+2, ny+2] (u, v, p, T)")
            self.init[:] = init_fields
        self.a = [0.0] * n_sgts

    def reset_fields(self):
        self.st[:] = 0.0
        self.obs[:] = 0.0
        self.a = [0.0] * self.n_sgts
        self.stp = 0

    def reset(self):
        self.reset_fields()
        self.st[:4] = self.init
        return self.get_obs(), None

    def solve(self, a=None):
        if a is None:
            a = list(self.a)
        av = np.array(a, dtype=np.float64)
        lib().orc_rayleigh_condition_action(C.byref(self.cfg), dp(av))
        for i in range(self.n_sgts):
            a[i] = av[i]          # the reference normalises the caller's list in place (:165-168)
        self.a = av.tolist()
        self._solve(av)

    def step(self, a=None):
        self.solve(a)
        return self._finish_step()

    T = property(lambda self: self.S)


class mixing(_ns2d):
    """mixing/mixing.py:16-378"""

    def __init__(self, cpu=0, L=1.0, H=1.0, re=100.0, pe=10000.0, side=0.5, C0=1.0):
        nx, ny = int(100 * L), int(100 * H)
        self.L, self.H, self.nx, self.ny, self.side, self.C0 = L, H, nx, ny, side, C0
        dt, dt_act, t_act = 0.002, 0.5, 50.0
        self.n_act = int(t_act / dt_act)
        nxo, nyo = 4 * int(L), 4 * int(H)
        self.cfg = ns2d_cfg(kind=1, nx=nx, ny=ny, ndt_act=int(dt_act / dt), n_sgts=0, nx_sgts=0,
                            nx_obs_pts=nxo, ny_obs_pts=nyo, nx_obs=nx // nxo, ny_obs=ny // nyo,
                            n_obs_steps=4, itmax=300000, dx=float(L / nx), dy=float(H / ny), dt=dt,
                            pr=0.0, ra=0.0, Tc=0.0, Th=0.0, C=0.0, re=re, pe=pe, u_max=re * 0.01 / L,
                            ref_c=(side * side) / (L * H) * C0, tol=1.0e-4)
        self._alloc()
        self.a = 1

    def reset_fields(self):
        c = self.cfg
        self.st[:] = 0.0
        i_min = math.floor(0.5 * (self.L - self.side) / c.dx)
        i_max = i_min + math.floor(self.side / c.dx)
        j_min = math.floor(0.5 * (self.H - self.side) / c.dy)
        j_max = j_min + math.floor(self.side / c.dy)
        self.S[i_min:i_max, j_min:j_max] = self.C0
        self.obs[:] = 0.0
        self.a = 1
        self.stp = 0

    def reset(self):
        self.reset_fields()
        return self.get_obs(), None

    def step(self, a=None):
        if a is None:
            a = self.a
        self.a = a
        self._solve(np.array([float(a)]))
        return self._finish_step()

    Cf = property(lambda self: self.S)


# ---------------------------------------------------------------------------
# 1D envs
# ---------------------------------------------------------------------------
class burgers(object):
    """burgers/burgers.py:17-227.  `nx` is an extra kwarg (the reference hard-codes 500, :26)."""

    def __init__(self, cpu=0, u_target=0.5, amp=10.0, sigma=0.1, ctrl_pos=1.0, L=2.0, nx=500):
        self.nx, self.sigma, self.u_target = nx, sigma, u_target
        dx = float(L / nx)
        dt = 0.2 * dx
        self.n_act = int(10.0 / 0.05)
        self.cfg = burgers_cfg(nx=nx, ndt_act=int(0.05 / dt), ctrl_pos=int(ctrl_pos / dx), n_obs_pts=5,
                               dx=dx, dt=dt, amp=amp, u_target=u_target)
        self.w = np.zeros((5, nx))
        self.u, self.up, self.upp, self.du, self.rhs = self.w
        self.a = [0.0]
        self.stp = 0

    def reset(self):
        self.w[:3] = self.u_target
        self.w[3:] = 0.0
        self.a = [0.0]
        self.stp = 0
        return self.get_obs(), None

    def get_obs(self):
        obs = np.zeros(self.cfg.n_obs_pts)
        self._rwd = lib().orc_burgers_obs_rwd(C.byref(self.cfg), dp(self.u), dp(obs))
        return obs

    def step(self, a=None, noise=None):
        """noise: the uniform(-sigma, sigma) inlet draw of this step (burgers.py:127); the
        reference takes it from the global numpy stream, here it is explicit."""
        if a is None:
            a = list(self.a)
        self.a = [a[0]]
        if noise is None:
            noise = float(np.random.uniform(-self.sigma, self.sigma, 1)[0])
        lib().orc_burgers_solve(C.byref(self.cfg), dp(self.u), dp(self.up), dp(self.upp), dp(self.du),
                                dp(self.rhs), C.c_double(float(a[0])), C.c_double(float(noise)))
        obs = self.get_obs()
        done = trunc = (self.stp == self.n_act - 1)
        self.stp += 1
        return obs, self._rwd, done, trunc, None


class shkadov(object):
    """shkadov/shkadov.py:16-372"""

    def __init__(self, cpu=0, init=True, L0=150.0, n_jets=5, jet_pos=150.0, jet_space=10.0, delta=0.1,
                 t_act=20.0, init_fields=None):
        L = L0 + jet_space * (n_jets + 2)
        nx = int(5 * L)
        dx = float(L / nx)
        dt, dt_act = 0.001, 0.05
        self.nx, self.n_jets, self.sigma = nx, n_jets, 5.0e-4
        self.n_act = int(t_act / dt_act)
        self.rand_init, self.rand_steps = True, 400
        l_obs = 10.0
        self.cfg = shkadov_cfg(nx=nx, ndt_act=int(dt_act / dt), n_jets=n_jets, jet_pos=int(jet_pos / dx),
                               jet_hw=int(2.0 / dx), jet_space=int(jet_space / dx), l_obs=int(l_obs / dx),
                               l_rwd=int(10.0 / dx), n_obs=int(l_obs), n_interp=int(0.02 / dt),
                               obs_stride=int(1.0 / dx), dx=dx, dt=dt, delta=delta, jet_amp=5.0, eps=1.0e-8)
        self.w = np.zeros((4, nx))
        self.h, self.q, self.rhsh, self.rhsq = self.w
        self.h_init, self.q_init = np.zeros(nx), np.zeros(nx)
        if init:
            if init_fields is None:
                raise ValueError("init=True needs init_fields[2, >=nx] (h, q)")
            self.h_init[:] = init_fields[0][:nx]
            self.q_init[:] = init_fields[1][:nx]
        self.reset_fields()

    def reset_fields(self):
        self.h[:] = 1.0
        self.q[:] = 1.0
        self.rhsh[:] = 0.0
        self.rhsq[:] = 0.0
        self.u = [0.0] * self.n_jets
        self.up = [0.0] * self.n_jets
        self.stp = 0

    def reset(self, n_rand=None, noise=None):
        """n_rand / noise[n_rand, ndt_act]: the reference draws them from python `random`
        and the numpy global stream (shkadov.py:119-123, :204); explicit here."""
        self.reset_fields()
        self.h[:] = self.h_init
        self.q[:] = self.q_init
        if self.rand_init:
            for i in range(n_rand):
                self.step(self.u, noise[i])
            self.stp = 0
        return self.get_obs(), None

    def get_obs(self):
        obs = np.zeros(self.cfg.n_obs * self.n_jets)
        blow = C.c_int(0)
        self._rwd = lib().orc_shkadov_obs_rwd(C.byref(self.cfg), dp(self.h), dp(self.q), dp(obs), C.byref(blow))
        self._blow = bool(blow.value)
        return obs

    def step(self, u=None, noise=None):
        if u is None:
            u = list(self.u)
        self.up = list(self.u)
        self.u = [float(x) for x in u]
        noise = np.ascontiguousarray(noise, dtype=np.float64)
        lib().orc_shkadov_solve(C.byref(self.cfg), dp(self.h), dp(self.q), dp(self.rhsh), dp(self.rhsq),
                                dp(np.array(self.u)), dp(np.array(self.up)), dp(noise))
        obs = self.get_obs()
        rwd = self._rwd
        done = trunc = (self.stp == self.n_act - 1)
        if self._blow:
            done, trunc, rwd = True, False, -1.0
        self.stp += 1
        return obs, rwd, done, trunc, None


class sloshing(object):
    """sloshing/sloshing.py:16-320"""

    def __init__(self, cpu=0, init=True, L=2.5, amp=5.0, alpha=0.0005, g=9.81, init_fields=None):
        nx = int(80 * L)
        dt, dt_act = 0.001, 0.05
        self.nx = nx
        self.n_act = int(10.0 / dt_act)
        self.n_warmup = int(2.0 / dt_act)
        self.dt_act = dt_act
        self.cfg = sloshing_cfg(nx=nx, ndt_act=int(dt_act / dt), n_interp=int(0.01 / dt), dx=float(L / nx),
                                dt=dt, g=g, amp=amp, alpha=alpha)
        self.n_obs = nx // 2 + (1 if nx % 2 else 0)
        self.w = np.zeros((4, nx + 2))
        self.h, self.q, self.rhsh, self.rhsq = self.w
        self.h_init, self.q_init = np.zeros(nx + 2), np.zeros(nx + 2)
        if init:
            if init_fields is None:
                raise ValueError("init=True needs init_fields[2, nx+2] (h, q)")
            self.h_init[:] = init_fields[0]
            self.q_init[:] = init_fields[1]
        self.reset_fields()

    def reset_fields(self):
        self.h[:] = 1.0
        self.q[:] = 0.0
        self.rhsh[:] = 0.0
        self.rhsq[:] = 0.0
        self.u, self.up = [0.0], [0.0]
        self.stp = 0

    def reset(self):
        self.reset_fields()
        self.h[:] = self.h_init
        self.q[:] = self.q_init
        return self.get_obs(), None

    @staticmethod
    def signal(t, dt):
        return 0.5 * (np.cos(np.pi * t) + 3.0 * np.cos(4.0 * np.pi * t))

    def get_obs(self):
        obs = np.zeros(self.n_obs)
        blow = C.c_int(0)
        self._rwd = lib().orc_sloshing_obs_rwd(C.byref(self.cfg), dp(self.h), dp(self.q),
                                               C.c_double(float(self.u[0])), dp(obs), C.byref(blow))
        self._blow = bool(blow.value)
        return obs

    def step(self, u=None):
        if u is None:
            u = list(self.u)
        self.up = list(self.u)
        self.u = [float(u[0])]
        lib().orc_sloshing_solve(C.byref(self.cfg), dp(self.h), dp(self.q), dp(self.rhsh), dp(self.rhsq),
                                 C.c_double(self.u[0]), C.c_double(self.up[0]))
        obs = self.get_obs()
        done = trunc = (self.stp == self.n_act - 1)
        if self._blow:                      # sloshing.py:152-160 (the -10 reward is dead code there)
            done, trunc = True, False
        self.stp += 1
        return obs, self._rwd, done, trunc, None


class lorenz(object):
    """lorenz/lorenz.py:18-260"""

    def __init__(self, cpu=0, sigma=10.0, rho=28.0, beta=8.0 / 3.0):
        self.sigma, self.rho, self.beta = sigma, rho, beta
        self.dt, self.ndt_act = 0.05, 1
        self.n_act = int(25.0 / 0.05)
        self.actions = np.array([-1.0, 0.0, 1.0])
        self.w = np.zeros((3, 3))
        self.x, self.xk, self.fx = self.w
        self.u = 1
        self.stp = 0

    def reset(self):
        self.w[:] = 0.0
        self.x[:] = 10.0
        self.u = 1
        self.stp = 0
        return self.get_obs(), None

    def get_obs(self):
        return np.concatenate([self.x, self.fx])

    def step(self, u=None):
        if u is None:
            u = self.u
        self.u = int(u)
        lib().orc_lorenz_solve(dp(self.x), dp(self.xk), dp(self.fx), C.c_double(self.sigma),
                               C.c_double(self.rho), C.c_double(self.beta), C.c_double(self.dt),
                               C.c_int(self.ndt_act), C.c_double(float(self.actions[self.u])))
        obs = self.get_obs()
        rwd = 1.0 if self.x[0] < 0.0 else 0.0
        done = trunc = (self.stp == self.n_act - 1)
        self.stp += 1
        return obs, rwd, done, trunc, None
