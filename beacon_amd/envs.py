"""Drop-in single-env mirrors of the reference's Gym classes: same class
names, constructor kwargs, reset()/step() return types and quirks -- each is a batch=1 view
of the corresponding batched env in vec.py, so every number comes from the HIP kernels.

    from beacon_amd.envs import rayleigh
    env = rayleigh()                 # loads the packaged developed-flow init state
    obs, _ = env.reset()
    obs, rwd, done, trunc, _ = env.step(act_list)

Differences from the reference, all deliberate (SURVEY.md 8b "Errors"/"Threading"):
  * Poisson non-convergence raises RuntimeError instead of print + exit(1);
  * init fields come from the packaged data file, not from ./init_field.dat in the cwd;
  * inlet noise / random warm-up counts still come from numpy's / python's global streams
    (drawn on the host and passed to the kernel as inputs), so seeding behaves the same.
Citations are file:line into /root/reference/beacon/.
"""
import os
import random

import numpy as np

from . import spaces, vec

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "init_fields.npz")


def packaged_init(name):
    """Parsed content of the reference's <env>/init_field.dat (see oracle/capture: init_data)."""
    with np.load(_DATA) as z:
        return z[name].copy()


class _Single(object):
    metadata = {"render.modes": ["human"]}

    def _np(self, t):
        return t.detach().cpu().numpy().astype(np.float64)

    def _episode(self):
        return bool(self.vec.done[0].item()), bool(self.vec.trunc[0].item())

    stp_plot = 0        # frames written since the last reset (rayleigh.py:128, ...)
    _render = None      # function of beacon_amd.render

    def render(self, mode="human", show=False, dump=True):
        """Host-side frame + text dumps in the reference's render/ layout (beacon_amd/render.py)."""
        from . import render as R
        getattr(R, self._render)(self, show, dump)

    def close(self):
        self.vec.close()

    @property
    def stp(self):
        return int(self.vec.get_stp()[0])

    def _obs_view(self, t):
        """rayleigh.py:243-262 / mixing.py:237-258: get_obs() fills the env's own history array `self.obs`
        [n_obs_steps, 3, nx_obs_pts, ny_obs_pts] in place and returns np.reshape(self.obs, [-1]) -- a VIEW: an observation a
        caller kept from the previous step shows the new values after the next one.  Same here: one array per env, refilled."""
        new = self._np(t)[0]
        if getattr(self, "obs", None) is None:
            self.obs = np.zeros((self.n_obs_steps, 3, self.nx_obs_pts, self.ny_obs_pts))
        self.obs[...] = new.reshape(self.obs.shape)
        return np.reshape(self.obs, [-1])


def _fields_2d(state):
    """device layout [4, ny+2, nx+2] -> the reference's four [nx+2, ny+2] arrays"""
    s = state.detach().cpu().numpy().astype(np.float64)[0]
    return [np.ascontiguousarray(s[k].T) for k in range(4)]


class rayleigh(_Single):
    _render = "rayleigh"
    """rayleigh/rayleigh.py:16-366"""

    def __init__(self, cpu=0, init=True, L=1.0, H=1.0, n_sgts=10, ra=1.0e4, device="cuda:0", dtype="f64"):
        init_fields = None
        if init:
            init_fields = packaged_init("rayleigh")
            if init_fields.shape != (4, int(50 * L) + 2, int(50 * H) + 2):
                raise ValueError("the packaged init field is 50x50 (L=H=1); pass init=False for other grids")
        self.vec = vec.VecRayleigh(1, device, dtype, init_fields, L, H, n_sgts, ra)
        v = self.vec
        for k in ("L", "H", "nx", "ny", "ra", "pr", "Tc", "Th", "C", "dt", "dt_act", "n_sgts", "nx_sgts",
                  "ndt_act", "n_act", "n_warmup", "dx", "dy", "n_obs_tot", "nx_obs", "ny_obs", "nx_obs_pts",
                  "ny_obs_pts", "n_obs_steps", "action_space", "observation_space"):
            setattr(self, k, getattr(v, k))
        self.a = [0.0] * n_sgts
        self.nu = np.empty((0, 2))

    def reset(self):
        obs, _ = self.vec.reset()
        self.a = [0.0] * self.n_sgts
        self.stp_plot = 0
        self.nu = np.empty((0, 2))                # (stp, Nusselt) per step since reset (rayleigh.py:124)
        return self._obs_view(obs), None

    def step(self, a=None):
        if a is None:
            self.vec.step(None)
        else:
            self.vec.step(np.asarray(a, dtype=np.float64).reshape(1, self.n_sgts))
        self.vec.check_status()
        an = self._np(self.vec.actions_norm)[0]
        if a is not None:
            for i in range(self.n_sgts):
                a[i] = an[i]            # the reference normalises the caller's list in place (:165-168)
        self.a = an.tolist()
        done, trunc = self._episode()
        rwd = float(self.vec.rwd[0].item())
        self.nu = np.append(self.nu, np.array([[self.stp - 1, -rwd]]), axis=0)   # get_rwd records (stp, nu) (:273)
        return self._obs_view(self.vec.obs), rwd, done, trunc, None

    def _field(self, k):
        return _fields_2d(self.vec.get_state())[k]

    u = property(lambda self: self._field(0))
    v = property(lambda self: self._field(1))
    p = property(lambda self: self._field(2))
    T = property(lambda self: self._field(3))

    def dump(self, field_name, act_name, nusselt_name=None):
        """Same text formats as rayleigh.py:344-353 (4 stacked blocks u,v,p,T, '%.5e'; the Nusselt history
        is the negated reward of every step since reset)."""
        np.savetxt(field_name, np.vstack(_fields_2d(self.vec.get_state())), fmt="%.5e")
        np.savetxt(act_name, self.a, fmt="%.5e")
        if nusselt_name is not None:
            np.savetxt(nusselt_name, self.nu, fmt="%.5e")

    def load(self, filename):
        """Read an init file in the reference's format (rayleigh.py:356-362) and make it the state
        that reset() restores."""
        f = np.loadtxt(filename)
        n = self.nx + 2
        fields = np.stack([f[k * n:(k + 1) * n, :] for k in range(4)])
        v = self.vec
        v._init_np = fields
        v._init_dev = v._real(np.ascontiguousarray(fields.transpose(0, 2, 1)), (4, self.ny + 2, self.nx + 2))

    def warmup(self):
        """rayleigh.warmup (rayleigh.py:131-135): n_warmup uncontrolled action steps."""
        self.vec.warmup(self.n_warmup)


class mixing(_Single):
    _render = "mixing"
    """mixing/mixing.py:16-378"""

    def __init__(self, cpu=0, L=1.0, H=1.0, re=100.0, pe=10000.0, side=0.5, C0=1.0, device="cuda:0",
                 dtype="f64"):
        self.vec = vec.VecMixing(1, device, dtype, L, H, re, pe, side, C0)
        v = self.vec
        for k in ("L", "H", "nx", "ny", "re", "pe", "C0", "side", "u_max", "dt", "dt_act", "ndt_act", "n_act",
                  "dx", "dy", "n_obs_tot", "nx_obs", "ny_obs", "nx_obs_pts", "ny_obs_pts", "n_obs_steps", "action_space",
                  "observation_space"):
            setattr(self, k, getattr(v, k))
        self.a = 1

    def reset(self):
        obs, _ = self.vec.reset()
        self.a = 1
        self.stp_plot = 0
        return self._obs_view(obs), None

    def step(self, a=None):
        if a is None:
            self.vec.step(None)
        else:
            self.a = a
            self.vec.step(np.asarray([int(a)]))
        self.vec.check_status()
        done, trunc = self._episode()
        return self._obs_view(self.vec.obs), float(self.vec.rwd[0].item()), done, trunc, None

    def _field(self, k):
        return _fields_2d(self.vec.get_state())[k]

    u = property(lambda self: self._field(0))
    v = property(lambda self: self._field(1))
    p = property(lambda self: self._field(2))
    C = property(lambda self: self._field(3))

    def dump(self, field_name, action_name):
        """Same formats as mixing.py:362-373 (4 stacked blocks u,v,p,C '%.5e'; action appended)."""
        np.savetxt(field_name, np.vstack(_fields_2d(self.vec.get_state())), fmt="%.5e")
        with open(action_name, "a") as f:
            f.write(str(self.a) + "\n")


class burgers(_Single):
    _render = "burgers"
    """burgers/burgers.py:17-227"""

    def __init__(self, cpu=0, u_target=0.5, amp=10.0, sigma=0.1, ctrl_pos=1.0, L=2.0, nx=500,
                 device="cuda:0", dtype="f64"):
        self.vec = vec.VecBurgers(1, device, dtype, u_target, amp, sigma, ctrl_pos, L, nx)
        v = self.vec
        for k in ("L", "nx", "amp", "sigma", "u_target", "dx", "dt", "ctrl_pos", "ndt_act", "n_act",
                  "n_obs_pts", "action_space", "observation_space"):
            setattr(self, k, getattr(v, k))
        self.a = [0.0]

    def reset(self):
        obs, _ = self.vec.reset()
        self.a = [0.0]
        self.stp_plot = 0
        return self._np(obs)[0], None

    def step(self, a=None):
        noise = np.random.uniform(-self.sigma, self.sigma, 1)     # burgers.py:127: global legacy stream
        if a is not None:
            self.a = [float(a[0])]
        self.vec.step(None if a is None else np.asarray([float(a[0])]), noise)
        done, trunc = self._episode()
        return self._np(self.vec.obs)[0], float(self.vec.rwd[0].item()), done, trunc, None

    u = property(lambda self: self._np(self.vec.get_state())[0, 0])
    up = property(lambda self: self._np(self.vec.get_state())[0, 1])
    upp = property(lambda self: self._np(self.vec.get_state())[0, 2])

    def dump(self, filename):
        """Same text format as burgers.py:216-222 (columns x, u)."""
        x = np.linspace(0, self.nx, num=self.nx, endpoint=False) * self.dx
        np.savetxt(filename, np.transpose(np.vstack((x, self.u))), fmt="%.5e")


class shkadov(_Single):
    _render = "shkadov"
    """shkadov/shkadov.py:16-372"""

    def __init__(self, cpu=0, init=True, L0=150.0, n_jets=5, jet_pos=150.0, jet_space=10.0, delta=0.1,
                 t_act=20.0, render_style="dynamic", device="cuda:0", dtype="f64"):
        init_fields = packaged_init("shkadov") if init else None
        self.vec = vec.VecShkadov(1, device, dtype, init_fields, L0, n_jets, jet_pos, jet_space, delta, t_act)
        v = self.vec
        if init and init_fields.shape[1] < v.nx:
            raise ValueError("the packaged init field has %d points < nx=%d; pass init=False" %
                             (init_fields.shape[1], v.nx))
        for k in ("L", "nx", "dx", "dt", "sigma", "delta", "n_jets", "jet_amp", "jet_pos", "jet_hw",
                  "jet_space", "l_obs", "l_rwd", "n_obs", "ndt_act", "n_act", "n_interp", "action_space",
                  "observation_space"):
            setattr(self, k, getattr(v, k))
        self.init = init
        self.render_style = render_style
        self.rand_init = True          # :49
        self.rand_steps = 400          # :50
        self.u = [0.0] * n_jets

    def _noise(self):
        # one np.random.uniform(-sigma, sigma, 1) per timestep (:204) == one bulk draw
        return np.random.uniform(-self.sigma, self.sigma, self.ndt_act).reshape(1, -1)

    def reset(self):
        """With init=False the reference's reset() copies an all-zero h_init/q_init over the
        fields (:115-117) and the solver divides by h; here init=False resets to the flat film
        h=q=1, which is what the reference's own init.py starts from (reset_fields, init.py:14)."""
        self.vec.reset()
        self.u = [0.0] * self.n_jets
        self.stp_plot = 0
        if self.rand_init and self.init:
            n = random.randint(0, self.rand_steps)                 # :120
            for i in range(n):
                self.vec.step(None, self._noise())
            self.vec.set_stp(0)                                    # :122
        # obs after the random steps is whatever the last step wrote; a fresh reset wrote it too
        return self._np(self.vec.obs)[0], None

    def step(self, u=None):
        if u is not None:
            self.u = [float(x) for x in u]
        self.vec.step(None if u is None else np.asarray(u, dtype=np.float64).reshape(1, -1), self._noise())
        if int(self.vec.status[0].item()) & 2:
            print("Blowup")                                        # :177
        done, trunc = self._episode()
        return self._np(self.vec.obs)[0], float(self.vec.rwd[0].item()), done, trunc, None

    h = property(lambda self: self._np(self.vec.get_state())[0, 0])
    q = property(lambda self: self._np(self.vec.get_state())[0, 1])

    def dump(self, field_name, jet_name):
        """Same text format as shkadov.py:353-361 (columns x, h, q)."""
        x = np.linspace(0, self.nx, num=self.nx, endpoint=False) * self.dx
        np.savetxt(field_name, np.transpose(np.vstack((x, self.h, self.q))), fmt="%.5e")
        np.savetxt(jet_name, self.u, fmt="%.5e")

    def load(self, filename):
        """shkadov.py:364-368: first nx rows of columns 1 (h) and 2 (q)."""
        f = np.loadtxt(filename)
        v = self.vec
        v._init_np = np.stack([f[:self.nx, 1], f[:self.nx, 2]])
        v._init_dev = v._real(np.ascontiguousarray(v._init_np), (2, self.nx))
        self.init = True


class sloshing(_Single):
    _render = "sloshing"
    """sloshing/sloshing.py:16-320"""

    def __init__(self, cpu=0, init=True, L=2.5, amp=5.0, alpha=0.0005, g=9.81, device="cuda:0", dtype="f64"):
        init_fields = None
        if init:
            init_fields = packaged_init("sloshing")
            if init_fields.shape[1] != int(80 * L) + 2:
                raise ValueError("the packaged init field has nx=200 (L=2.5); pass init=False")
        self.vec = vec.VecSloshing(1, device, dtype, init_fields, L, amp, alpha, g)
        v = self.vec
        for k in ("L", "nx", "dx", "dt", "dt_act", "g", "amp", "alpha", "ndt_act", "n_act", "n_warmup",
                  "n_interp", "n_obs", "action_space", "observation_space"):
            setattr(self, k, getattr(v, k))
        self.u = [0.0]

    signal = staticmethod(vec.VecSloshing.signal)

    def reset(self):
        obs, _ = self.vec.reset()
        self.u = [0.0]
        self.stp_plot = 0
        return self._np(obs)[0], None

    def step(self, u=None):
        if u is not None:
            self.u = [float(u[0])]
        self.vec.step(None if u is None else np.asarray([float(u[0])]))
        if int(self.vec.status[0].item()) & 2:
            print("Blowup")                                        # :157
        done, trunc = self._episode()
        return self._np(self.vec.obs)[0], float(self.vec.rwd[0].item()), done, trunc, None

    h = property(lambda self: self._np(self.vec.get_state())[0, 0])
    q = property(lambda self: self._np(self.vec.get_state())[0, 1])


    def dump(self, field_name, control_name=None):
        """Same text format as sloshing.py:297-307 (columns x, h[1:nx+1], q[1:nx+1])."""
        x = np.linspace(0, self.nx, num=self.nx, endpoint=False) * self.dx
        np.savetxt(field_name, np.transpose(np.vstack((x, self.h[1:self.nx + 1], self.q[1:self.nx + 1]))), fmt="%.5e")
        if control_name is not None:
            np.savetxt(control_name, self.u, fmt="%.5e")

    def load(self, filename):
        """sloshing.py:309-313: interior values of h and q, ghosts left at zero."""
        f = np.loadtxt(filename)
        init = np.zeros((2, self.nx + 2))
        init[0, 1:self.nx + 1], init[1, 1:self.nx + 1] = f[:, 1], f[:, 2]
        v = self.vec
        v._init_np = init
        v._init_dev = v._real(init, (2, self.nx + 2))

    def warmup(self):
        """Excitation warm-up of beacon/sloshing/init.py (signal(), n_warmup action steps)."""
        t = 0.0
        for _ in range(self.n_warmup):
            self.vec.step(np.asarray([self.signal(t, self.dt_act)]))
            t += self.dt_act


class shkadov_separable(shkadov):
    """shkadov/shkadov.py:376-481: round-robin per-jet view of shkadov -- the solver advances only
    when the jet counter is 0; every call returns the 10 observations and the reward term of ONE
    jet.  Host-side re-indexing of the batched kernel's outputs; no new numerics."""

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.observation_space = spaces.sym_box(1.0, self.n_obs)           # shkadov.py:394-398
        self.count = 0
        self.act = np.zeros(self.n_jets)
        self._obs_all = np.zeros(self.n_obs * self.n_jets)
        self._blow = False

    def _jet_rwd(self, i):
        h = self.h
        s = self.jet_pos + i * self.jet_space
        d = h[s:s + self.l_rwd] - 1.0
        return -(np.sum(np.square(d)) * self.dx) / (self.n_jets * self.l_rwd)

    def _advance(self):
        if self.count == self.n_jets - 1:
            self.count = 0
            return True
        self.count += 1
        return False

    def reset(self):
        if self.count == 0:
            self._obs_all, _ = super().reset()
            self._blow = False
            self._stp_shadow = 0
        obs = self._obs_all[self.count * self.n_obs:(self.count + 1) * self.n_obs].copy()
        self._advance()
        return obs, None

    def step(self, u=None):
        if self.count == 0:
            if u is not None:
                self.u = [float(x) for x in u]
            self.vec.step(None if u is None else np.asarray(u, dtype=np.float64).reshape(1, -1), self._noise())
            self.vec.set_stp(self._stp_shadow)            # the batched kernel counts solver steps;
            self._obs_all = self._np(self.vec.obs)[0]      # the separable episode counter moves per round
            self._blow = bool(int(self.vec.status[0].item()) & 2)
        obs = self._obs_all[self.count * self.n_obs:(self.count + 1) * self.n_obs].copy()
        rwd = self._jet_rwd(self.count)
        done = trunc = (self._stp_shadow == self.n_act - 1)
        if self._blow:
            print("Blowup")
            done, trunc, rwd = True, False, -1.0
        if self._advance():
            self._stp_shadow += 1
            self.vec.set_stp(self._stp_shadow)
        return obs, rwd, done, trunc, None

    _stp_shadow = 0

    @property
    def stp(self):
        return self._stp_shadow
