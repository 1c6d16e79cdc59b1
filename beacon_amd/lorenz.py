"""lorenz-v0: the reference's 3-variable ODE env (lorenz/lorenz.py:18-260) -- BASELINE.json's
"plumbing" config: single env, CPU, no GPU kernel (3 unknowns, 5 RK stages per action).
Host-only NumPy; it exists to prove the boundary (ctor / reset / step signatures, Discrete
actions given as NumPy integers, obs = (x, f(x)) of the LAST RK stage)."""
import numpy as np

from . import spaces

# Carpenter-Kennedy 5-stage 4th-order low-storage RK coefficients (lorenz.py:272-280)
_A = (0.000000000000000, -0.417890474499852, -1.192151694642677, -1.697784692471528, -1.514183444257156)
_B = (0.149659021999229, 0.379210312999627, 0.822955029386982, 0.699450455949122, 0.153057247968152)


class lorenz(object):
    metadata = {"render.modes": ["human"]}

    def __init__(self, cpu=0, sigma=10.0, rho=28.0, beta=8.0 / 3.0):
        self.dt, self.dt_act, self.t_max = 0.05, 0.05, 25.0
        self.sigma, self.rho, self.beta = sigma, rho, beta
        self.n_obs = 6
        self.ndt_act = int(self.dt_act / self.dt)
        self.n_act = int(self.t_max / self.dt_act)
        self.x, self.xk, self.fx = np.zeros(3), np.zeros(3), np.zeros(3)
        self.action_space = spaces.discrete(3)
        self.actions = np.array([-1.0, 0.0, 1.0])
        self.observation_space = spaces.sym_box(1.0, self.n_obs)
        self.reset_fields()

    def reset_fields(self):
        self.t = 0.0
        self.x[:] = 10.0
        self.xk[:] = 0.0
        self.fx[:] = 0.0
        self.hx = [self.x.copy()]
        self.ht = [self.t]
        self.u = 1
        self.stp = 0
        self.stp_plot = 0

    def reset(self):
        self.reset_fields()
        return self.get_obs(), None

    def get_obs(self):
        return np.concatenate([self.x, self.fx])

    def get_rwd(self):
        return 1.0 if self.x[0] < 0.0 else 0.0

    def solve(self, u=None):
        if u is None:
            u = self.u
        self.u = int(u)
        force = self.actions[self.u]
        x, xk, fx = self.x, self.xk, self.fx
        for _ in range(self.ndt_act):
            xk[:] = x
            for j in range(5):
                fx[0] = self.sigma * (xk[1] - xk[0])
                fx[1] = xk[0] * (self.rho - xk[2]) - xk[1]
                fx[2] = xk[0] * xk[1] - self.beta * xk[2]
                fx[1] += force
                for i in range(3):                      # lsrk4.update (lorenz.py:293-297)
                    x[i] = _A[j] * x[i] + self.dt * fx[i]
                    xk[i] += _B[j] * x[i]
            x[:] = xk
            self.t += self.dt
            self.hx.append(x.copy())
            self.ht.append(self.t)

    def step(self, u=None):
        self.solve(u)
        obs, rwd = self.get_obs(), self.get_rwd()
        done = trunc = (self.stp == self.n_act - 1)
        self.stp += 1
        return obs, rwd, done, trunc, None

    def render(self, mode="human", show=False, dump=True):
        """Host-side frames in the reference's png/ layout (beacon_amd/render.py; lorenz.py:175-248)."""
        from . import render as R
        R.lorenz(self, show, dump)

    def dump(self, filename):
        """lorenz.py:251-259: columns t, x, y, z of the trajectory since reset, '%.5e'."""
        hx = np.asarray(self.hx)
        np.savetxt(filename, np.column_stack((np.asarray(self.ht), hx[:, 0], hx[:, 1], hx[:, 2])), fmt="%.5e")

    def close(self):
        pass
