#!/usr/bin/env python3
"""Golden-vector capture from the real beacon reference (build container ONLY).

TEST INFRASTRUCTURE.  This script imports the unmodified reference modules
read-only from /root/reference (with the two stand-in packages under
./stubs for `gymnasium` and `numba`, neither of which is installable here),
drives their own reset()/step()/kernel functions on seeded inputs and writes
inputs + expected outputs as small .npz fixtures under tests/golden/.

Nothing from the reference is copied: fixtures are data only.  The script is
never run on the GPU box (the reference does not exist there) and is never
imported by the product package.

usage:  python oracle/capture/capture.py <job> [<job> ...]     (jobs: see JOBS)
        python oracle/capture/capture.py all                   (serial)
"""
import importlib.util
import os
import random
import sys
import time

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = os.environ.get("BEACON_REFERENCE", "/root/reference")
GOLD = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, os.path.join(HERE, "stubs"))

import matplotlib  # noqa: E402

matplotlib.use("Agg")


def load_ref(env):
    """Import /root/reference/beacon/<env>/<env>.py under the name ref_<env>."""
    d = os.path.join(REF, "beacon", env)
    path = os.path.join(d, env + ".py")
    spec = importlib.util.spec_from_file_location("ref_" + env, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod, d


class cwd(object):
    """The reference loads init_field.dat by relative path (rayleigh.py:72)."""

    def __init__(self, d):
        self.d = d

    def __enter__(self):
        self.old = os.getcwd()
        os.chdir(self.d)

    def __exit__(self, *a):
        os.chdir(self.old)


def save(name, **arrs):
    os.makedirs(GOLD, exist_ok=True)
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024.0), flush=True)


# --------------------------------------------------------------------------
# 2D envs: wrappers around the reference's module-level kernels record
# per-stage snapshots (the reference's solve() looks the kernels up in module
# globals at call time, so rebinding the names is enough).
# --------------------------------------------------------------------------
class StageRecorder(object):
    def __init__(self, mod, scalar_name, n_snap):
        self.mod, self.sname, self.n_snap = mod, scalar_name, n_snap
        self.itp = []
        self.snaps = {}
        self.n = {"pred": 0, "pois": 0, "corr": 0, "tran": 0}
        self.orig = (mod.predictor, mod.poisson, mod.corrector, mod.transport)
        mod.predictor, mod.poisson = self.predictor, self.poisson
        mod.corrector, mod.transport = self.corrector, self.transport

    def restore(self):
        (self.mod.predictor, self.mod.poisson,
         self.mod.corrector, self.mod.transport) = self.orig

    def put(self, key, arr):
        self.snaps.setdefault(key, []).append(np.array(arr, copy=True))

    def predictor(self, u, v, us, vs, p, *rest):
        k = self.n["pred"]
        self.n["pred"] += 1
        if k < self.n_snap:
            self.put("bc_u", u), self.put("bc_v", v), self.put("bc_p", p)
            if self.sname == "T":
                self.put("bc_T", rest[0])
        self.orig[0](u, v, us, vs, p, *rest)
        if k < self.n_snap:
            self.put("pred_us", us), self.put("pred_vs", vs)

    def poisson(self, us, vs, u, phi, *rest):
        k = self.n["pois"]
        self.n["pois"] += 1
        itp, ovf = self.orig[1](us, vs, u, phi, *rest)
        self.itp.append(itp)
        if k < self.n_snap:
            self.put("pois_phi", phi)
        return itp, ovf

    def corrector(self, u, v, us, vs, phi, *rest):
        k = self.n["corr"]
        self.n["corr"] += 1
        self.orig[2](u, v, us, vs, phi, *rest)
        if k < self.n_snap:
            self.put("corr_u", u), self.put("corr_v", v)

    def transport(self, u, v, S, *rest):
        k = self.n["tran"]
        self.n["tran"] += 1
        if k < self.n_snap:
            self.put("tran_in", S)
        self.orig[3](u, v, S, *rest)
        if k < self.n_snap:
            self.put("tran_out", S)

    def stacked(self):
        return {k: np.stack(v) for k, v in self.snaps.items()}


def job_rayleigh_default():
    """rayleigh 50x50, shipped init_field.dat: 3 step()s, stage snapshots of
    the first 3 timesteps (pins SURVEY R1-R11)."""
    mod, d = load_ref("rayleigh")
    with cwd(d):
        s = mod.rayleigh()
    out = dict(u_init=s.u_init, v_init=s.v_init, p_init=s.p_init, T_init=s.T_init,
               params=np.array([s.nx, s.ny, s.ndt_act, s.n_act, s.n_sgts, s.nx_sgts,
                                s.nx_obs_pts, s.ny_obs_pts, s.nx_obs, s.ny_obs], dtype=np.int64),
               fparams=np.array([s.L, s.H, s.dx, s.dy, s.dt, s.pr, s.ra, s.Tc, s.Th, s.C]))
    obs0, info = s.reset()
    assert info is None
    out["reset_obs"] = obs0.copy()
    rng = np.random.default_rng(0)
    acts = rng.uniform(-1.0, 1.0, (2, s.n_sgts))
    rec = StageRecorder(mod, "T", 3)
    t0 = time.time()
    for k in range(3):
        a = acts[k].tolist() if k < 2 else None  # 3rd step: a=None repeats (rayleigh.py:162)
        obs, rwd, done, trunc, info = s.step(a)
        out["step%d_obs" % k] = obs.copy()
        out["step%d_rwd" % k] = rwd
        out["step%d_done" % k] = np.array([done, trunc])
        out["step%d_a_norm" % k] = np.array(s.a)
        for f in "uvpT":
            out["step%d_%s" % (k, f)] = getattr(s, f).copy()
        if k < 2:
            out["step%d_a_mutated" % k] = np.array(a)  # caller's list is normalised in place
        print("rayleigh_default step", k, "rwd", rwd, "%.1fs" % (time.time() - t0), flush=True)
    rec.restore()
    out["actions"] = acts
    out["itp"] = np.array(rec.itp, dtype=np.int64).reshape(3, s.ndt_act)
    out["nu_hist"] = s.nu.copy()
    out.update(rec.stacked())
    save("rayleigh_default", **out)


def synth_state_2d(nx, ny, dx, dy, seed, amp_psi, with_T, Th=0.5, Tc=-0.5):
    """Seeded smooth synthetic state: discretely divergence-free (u, v) from a
    corner streamfunction, linear conduction profile + smooth perturbation."""
    rng = np.random.default_rng(seed)
    L, H = nx * dx, ny * dy
    xc = (np.arange(nx + 2) - 1.0) * dx          # corner (i,j) = south-west of cell (i,j)
    yc = (np.arange(ny + 2) - 1.0) * dy
    X, Y = np.meshgrid(xc, yc, indexing="ij")
    psi = np.zeros((nx + 2, ny + 2))
    for k in range(1, 4):
        a = amp_psi * rng.uniform(0.5, 1.0) / k
        psi += a * np.sin(k * np.pi * X / L) ** 2 * np.sin(np.pi * Y / H) ** 2 * np.cos(k * np.pi * X / L + rng.uniform(0, 6.28))
    u = np.zeros((nx + 2, ny + 2))
    v = np.zeros((nx + 2, ny + 2))
    u[1:nx + 2, 1:ny + 1] = (psi[1:nx + 2, 2:ny + 2] - psi[1:nx + 2, 1:ny + 1]) / dy
    v[1:nx + 1, 1:ny + 2] = -(psi[2:nx + 2, 1:ny + 2] - psi[1:nx + 1, 1:ny + 2]) / dx
    S = np.zeros((nx + 2, ny + 2))
    if with_T:
        xm = (np.arange(nx + 2) - 0.5) * dx
        ym = (np.arange(ny + 2) - 0.5) * dy
        XM, YM = np.meshgrid(xm, ym, indexing="ij")
        S[:, :] = Th + (Tc - Th) * YM / H
        S += 0.1 * rng.uniform(0.5, 1.0) * np.sin(np.pi * YM / H) * np.cos(2.0 * np.pi * XM / L + rng.uniform(0, 6.28))
        S[:, 0] = 0.0    # bottom ghosts beyond the last segment are never set by the BC loop
        S[:, -1] = 0.0
        S[0, :] = 0.0
        S[-1, :] = 0.0
    p = 0.01 * rng.standard_normal((nx + 2, ny + 2))
    return u, v, p, S


def job_rayleigh_128x64():
    """rayleigh at the BASELINE grid (L=2.56,H=1.28 -> 128x64), seeded synthetic
    state, one step() of 5 timesteps with a non-trivial action."""
    mod, d = load_ref("rayleigh")
    s = mod.rayleigh(init=False, L=2.56, H=1.28)
    s.reset_fields()
    u, v, p, T = synth_state_2d(s.nx, s.ny, s.dx, s.dy, 7, 0.02, True)
    s.u[:], s.v[:], s.p[:], s.T[:] = u, v, p, T
    s.ndt_act = 5
    out = dict(u0=u, v0=v, p0=p, T0=T,
               params=np.array([s.nx, s.ny, s.ndt_act, s.n_act, s.n_sgts, s.nx_sgts,
                                s.nx_obs_pts, s.ny_obs_pts, s.nx_obs, s.ny_obs], dtype=np.int64),
               fparams=np.array([s.L, s.H, s.dx, s.dy, s.dt, s.pr, s.ra, s.Tc, s.Th, s.C]))
    obs0 = s.get_obs()
    out["obs0"] = obs0.copy()
    acts = np.random.default_rng(11).uniform(-1.0, 1.0, (1, s.n_sgts))
    rec = StageRecorder(mod, "T", 5)
    t0 = time.time()
    obs, rwd, done, trunc, _ = s.step(acts[0].tolist())
    rec.restore()
    print("rayleigh_128x64 itp", rec.itp, "%.1fs" % (time.time() - t0), flush=True)
    out.update(actions=acts, step0_obs=obs.copy(), step0_rwd=rwd, step0_a_norm=np.array(s.a),
               itp=np.array(rec.itp, dtype=np.int64).reshape(1, -1))
    for f in "uvpT":
        out["step0_" + f] = getattr(s, f).copy()
    out.update(rec.stacked())
    save("rayleigh_128x64", **out)


def job_rayleigh_128x64_step(k):
    """One FULL reference step() (200 timesteps, ~93 Jacobi sweeps each) at the BASELINE grid from the bench's
    developed initial state (tests/golden/rayleigh_128x64_init.npz) with the action vector bench.py gives replica k
    at its first step: default_rng(1234).uniform(-1, 1, (steps, B, 10))[0, k].  Pins the float32 tolerance of the
    timed dispatch against the reference itself (pure-Python loops here: ~10 minutes)."""
    mod, d = load_ref("rayleigh")
    s = mod.rayleigh(init=False, L=2.56, H=1.28)
    s.reset_fields()
    init = np.load(os.path.join(GOLD, "rayleigh_128x64_init.npz"))["fields"]
    s.u[:], s.v[:], s.p[:], s.T[:] = init
    act = np.random.default_rng(1234).uniform(-1.0, 1.0, (k + 1, s.n_sgts))[k]
    rec = StageRecorder(mod, "T", 0)
    t0 = time.time()
    obs, rwd, done, trunc, _ = s.step(act.tolist())
    rec.restore()
    print("rayleigh_128x64_step%d mean itp %.1f  %.1fs" % (k, np.mean(rec.itp), time.time() - t0), flush=True)
    out = dict(replica=k, action=act, obs=obs.copy(), rwd=rwd, a_norm=np.array(s.a),
               itp=np.array(rec.itp, dtype=np.int64))
    for f in "uvpT":
        out[f] = getattr(s, f).copy()
    save("rayleigh_128x64_step%d" % k, **out)


def job_mixing(action):
    """mixing 100x100 from reset, one step() of 3 timesteps with `action`
    (first Poisson solve from rest takes ~2.5k sweeps: pins M3's stop logic)."""
    mod, d = load_ref("mixing")
    s = mod.mixing()
    obs0, _ = s.reset()
    out = dict(reset_obs=obs0.copy(), reset_C=s.C.copy(), reset_rwd=s.get_rwd(),
               params=np.array([s.nx, s.ny, 3, s.n_act, s.nx_obs_pts, s.ny_obs_pts,
                                s.nx_obs, s.ny_obs], dtype=np.int64),
               fparams=np.array([s.L, s.H, s.dx, s.dy, s.dt, s.re, s.pe, s.u_max, s.side, s.C0]),
               full_ndt_act=s.ndt_act)
    s.ndt_act = 3
    rec = StageRecorder(mod, "C", 3)
    t0 = time.time()
    obs, rwd, done, trunc, _ = s.step(np.int64(action))
    rec.restore()
    print("mixing a=%d itp" % action, rec.itp, "%.1fs" % (time.time() - t0), flush=True)
    out.update(action=action, step0_obs=obs.copy(), step0_rwd=rwd,
               itp=np.array(rec.itp, dtype=np.int64).reshape(1, -1))
    for f in "uvpC":
        out["step0_" + f] = getattr(s, f).copy()
    out.update(rec.stacked())
    save("mixing_a%d" % action, **out)


def job_mixing_synth():
    """mixing 100x100 from a seeded developed-like state (few sweeps), 4 timesteps
    per action 0..3 and an out-of-range action 4 (all walls at rest)."""
    mod, d = load_ref("mixing")
    out = {}
    for action in range(5):
        s = mod.mixing()
        s.reset()
        u, v, p, _ = synth_state_2d(s.nx, s.ny, s.dx, s.dy, 21, 0.01, False)
        s.u[:], s.v[:], s.p[:] = u, v, p
        s.ndt_act = 4
        rec = StageRecorder(mod, "C", 0)
        t0 = time.time()
        obs, rwd, done, trunc, _ = s.step(np.int64(action))
        rec.restore()
        print("mixing_synth a=%d itp" % action, rec.itp, "%.1fs" % (time.time() - t0), flush=True)
        if action == 0:
            out.update(u0=u, v0=v, p0=p, C0=s_C0(mod))
        out["a%d_obs" % action] = obs.copy()
        out["a%d_rwd" % action] = rwd
        out["a%d_itp" % action] = np.array(rec.itp, dtype=np.int64)
        for f in "uvpC":
            out["a%d_%s" % (action, f)] = getattr(s, f).copy()
    save("mixing_synth", **out)


def _ctor_case(out, tag, mod, s, sname, state, action, ndt, nsnap):
    """`ndt` timesteps of one step() from `state` with stage snapshots of the first `nsnap` timesteps, stored under the prefix `tag`."""
    for f, a in zip("uvp" + sname, state):
        if a is not None:
            getattr(s, f)[:] = a
    s.ndt_act = ndt
    for f in "uvp" + sname:
        out["%s_%s0" % (tag, f)] = getattr(s, f).copy()
    rec = StageRecorder(mod, sname, nsnap)
    t0 = time.time()
    obs, rwd, done, trunc, _ = s.step(action)
    rec.restore()
    print(tag, "itp", rec.itp, "%.1fs" % (time.time() - t0), flush=True)
    out[tag + "_obs"], out[tag + "_rwd"] = obs.copy(), rwd
    out[tag + "_itp"] = np.array(rec.itp, dtype=np.int64)
    for f in "uvp" + sname:
        out["%s_%s" % (tag, f)] = getattr(s, f).copy()
    for k, v in rec.stacked().items():
        out["%s_%s" % (tag, k)] = v


def job_ctor_args():
    """The reference's NON-geometric constructor arguments: mixing(re, pe, side, C0) (mixing.py:21-34; u_max =
    re nu / L scales the lid speed, so re moves the transport's CFL number and pe its diffusion number) and rayleigh(n_sgts, ra)
    (rayleigh.py:20-27), a few timesteps each from a seeded synthetic state with stage snapshots, plus reset() where side / C0 matter."""
    out = {}
    mod, d = load_ref("mixing")
    for tag, kw, amp, action in (("mix_re50_pe1e3_a0", dict(re=50.0, pe=1.0e3), 0.01, 0),
                                 ("mix_re50_pe1e3_a3", dict(re=50.0, pe=1.0e3), 0.01, 3),
                                 ("mix_re200_pe1e5_a1", dict(re=200.0, pe=1.0e5, side=0.3, C0=2.0), 0.02, 1),
                                 ("mix_re200_pe1e5_a2", dict(re=200.0, pe=1.0e5, side=0.3, C0=2.0), 0.02, 2),
                                 ("mix_re400_pe2e3_a0", dict(re=400.0, pe=2.0e3, side=0.62, C0=0.5), 0.04, 0)):
        s = mod.mixing(**kw)
        obs0, _ = s.reset()
        out[tag + "_reset_obs"], out[tag + "_reset_C"], out[tag + "_reset_rwd"] = obs0.copy(), s.C.copy(), s.get_rwd()
        out[tag + "_fparams"] = np.array([s.L, s.H, s.dx, s.dy, s.dt, s.re, s.pe, s.u_max, s.side, s.C0])
        u, v, p, _ = synth_state_2d(s.nx, s.ny, s.dx, s.dy, 21, amp, False)
        # (the second action of a parameter set starts from the same state: only its results are kept, and no snapshots)
        first = tag in ("mix_re50_pe1e3_a0", "mix_re200_pe1e5_a1", "mix_re400_pe2e3_a0")
        _ctor_case(out, tag, mod, s, "C", (u, v, p, None), np.int64(action), 4, 1 if first else 0)
        if not first:
            for suf in ("u0", "v0", "p0", "C0", "reset_obs", "reset_C", "reset_rwd", "fparams"):
                del out["%s_%s" % (tag, suf)]
    mod, d = load_ref("rayleigh")
    for tag, kw, seed in (("ray_sgts5_ra5e4", dict(n_sgts=5, ra=5.0e4), 31), ("ray_sgts12_ra8e3_50x75", dict(n_sgts=12, ra=8.0e3, H=1.5), 32),
                          ("ray_sgts3_ra2e5", dict(n_sgts=3, ra=2.0e5), 33)):
        s = mod.rayleigh(init=False, **kw)
        s.reset_fields()
        out[tag + "_fparams"] = np.array([s.L, s.H, s.dx, s.dy, s.dt, s.pr, s.ra, s.Tc, s.Th, s.C])
        out[tag + "_params"] = np.array([s.nx, s.ny, s.n_sgts, s.nx_sgts], dtype=np.int64)
        u, v, p, T = synth_state_2d(s.nx, s.ny, s.dx, s.dy, seed, 0.02, True)
        act = np.random.default_rng(seed).uniform(-1.0, 1.0, s.n_sgts)
        out[tag + "_action"] = act.copy()
        _ctor_case(out, tag, mod, s, "T", (u, v, p, T), act.tolist(), 5, 2)
        out[tag + "_a_norm"] = np.array(s.a)
    save("ctor_args", **out)


def s_C0(mod):
    s = mod.mixing()
    s.reset()
    return s.C.copy()


def noise_stream(seed, sigma, n):
    """The reference draws np.random.uniform(-s, s, 1) one at a time from the
    global legacy stream; n successive draws equal one bulk draw."""
    np.random.seed(seed)
    return np.random.uniform(-sigma, sigma, n)


def job_burgers():
    mod, d = load_ref("burgers")
    out = {}
    for seed in (0, 1):
        s = mod.burgers()
        obs0, _ = s.reset()
        n = s.n_act if seed == 0 else 25
        acts = np.random.default_rng(100 + seed).uniform(-1.0, 1.0, (n, 1))
        np.random.seed(seed)
        obs, rwd, dn = [], [], []
        for k in range(n):
            o, r, done, trunc, _ = s.step(acts[k].tolist())
            obs.append(o.copy()), rwd.append(r), dn.append([done, trunc])
        pre = "s%d_" % seed
        out.update({pre + "reset_obs": obs0, pre + "actions": acts, pre + "obs": np.array(obs),
                    pre + "rwd": np.array(rwd), pre + "done": np.array(dn),
                    pre + "noise": noise_stream(seed, s.sigma, n),
                    pre + "u": s.u.copy(), pre + "up": s.up.copy(), pre + "upp": s.upp.copy()})
        out["params"] = np.array([s.nx, s.ndt_act, s.n_act, s.ctrl_pos, s.n_obs_pts], dtype=np.int64)
        out["fparams"] = np.array([s.L, s.dx, s.dt, s.amp, s.sigma, s.u_target])
    save("burgers", **out)


def shkadov_params(s):
    return (np.array([s.nx, s.ndt_act, s.n_act, s.n_jets, s.jet_pos, s.jet_hw, s.jet_space,
                      s.l_obs, s.l_rwd, s.n_obs, s.n_interp], dtype=np.int64),
            np.array([s.L, s.dx, s.dt, s.delta, s.sigma, s.jet_amp, s.eps]))


def job_shkadov():
    mod, d = load_ref("shkadov")
    out = {}
    for tag, kw, init, n in (("j5", dict(n_jets=5), True, 30),
                             ("j10", dict(n_jets=10), True, 30),
                             ("n4096", dict(L0=699.2, n_jets=10), False, 6)):
        with cwd(d):
            s = mod.shkadov(init=init, **kw)
        s.rand_init = False
        if init:
            obs0, _ = s.reset()
        else:                     # as beacon/shkadov/init.py:14: flat film h=q=1, no init file
            s.reset_fields()
            obs0 = s.get_obs()
        acts = np.random.default_rng(200).uniform(-1.0, 1.0, (n, s.n_jets))
        seed = 5
        np.random.seed(seed)
        obs, rwd, dn = [], [], []
        for k in range(n):
            o, r, done, trunc, _ = s.step(acts[k].tolist())
            obs.append(o.copy()), rwd.append(r), dn.append([done, trunc])
        ip, fp = shkadov_params(s)
        pre = tag + "_"
        out.update({pre + "params": ip, pre + "fparams": fp, pre + "h_init": s.h_init.copy(),
                    pre + "q_init": s.q_init.copy(), pre + "reset_obs": obs0, pre + "actions": acts,
                    pre + "obs": np.array(obs), pre + "rwd": np.array(rwd), pre + "done": np.array(dn),
                    pre + "noise": noise_stream(seed, s.sigma, n * s.ndt_act).reshape(n, s.ndt_act),
                    pre + "h": s.h.copy(), pre + "q": s.q.copy(),
                    pre + "rhsh": s.rhsh.copy(), pre + "rhsq": s.rhsq.copy()})
        print("shkadov", tag, "done", flush=True)
    # rand_init reset path (shkadov.py:119-123): python `random` picks the count
    with cwd(d):
        s = mod.shkadov(n_jets=5)
    random.seed(3)
    np.random.seed(9)
    obs0, _ = s.reset()
    random.seed(3)
    n_rand = random.randint(0, s.rand_steps)
    out.update(rand_n=n_rand, rand_reset_obs=obs0, rand_h=s.h.copy(), rand_q=s.q.copy(),
               rand_noise=noise_stream(9, s.sigma, n_rand * s.ndt_act).reshape(n_rand, s.ndt_act))
    save("shkadov", **out)


def job_sloshing():
    mod, d = load_ref("sloshing")
    with cwd(d):
        s = mod.sloshing()
    obs0, _ = s.reset()
    n = 40
    acts = np.random.default_rng(300).uniform(-1.0, 1.0, (n, 1))
    obs, rwd, dn = [], [], []
    for k in range(n):
        o, r, done, trunc, _ = s.step(acts[k].tolist())
        obs.append(o.copy()), rwd.append(r), dn.append([done, trunc])
    out = dict(params=np.array([s.nx, s.ndt_act, s.n_act, s.n_interp, s.n_obs], dtype=np.int64),
               fparams=np.array([s.L, s.dx, s.dt, s.g, s.amp, s.alpha]),
               h_init=s.h_init.copy(), q_init=s.q_init.copy(), reset_obs=obs0, actions=acts,
               obs=np.array(obs), rwd=np.array(rwd), done=np.array(dn),
               h=s.h.copy(), q=s.q.copy(), rhsh=s.rhsh.copy(), rhsq=s.rhsq.copy())
    # excitation warm-up from rest as beacon/sloshing/init.py does (signal(), init=False)
    s2 = mod.sloshing(init=False)
    s2.reset_fields()
    t = 0.0
    for it in range(s2.n_warmup):
        s2.step([s2.signal(t, s2.dt_act)])
        t += s2.dt_act
    out.update(warm_h=s2.h.copy(), warm_q=s2.q.copy())
    save("sloshing", **out)


def job_lorenz():
    mod, d = load_ref("lorenz")
    out = {}
    for tag in ("a0", "a1", "a2", "rnd"):
        s = mod.lorenz()
        obs0, _ = s.reset()
        obs0 = obs0.copy()        # the returned array is a view of s.obs that later steps overwrite
        n = s.n_act
        if tag == "rnd":
            acts = np.random.default_rng(400).integers(0, 3, n)
        else:
            acts = np.full(n, int(tag[1]), dtype=np.int64)
        obs, rwd, dn = [], [], []
        for k in range(n):
            o, r, done, trunc, _ = s.step(np.int64(acts[k]))
            obs.append(o.copy()), rwd.append(r), dn.append([done, trunc])
        out.update({tag + "_reset_obs": obs0.copy(), tag + "_actions": acts, tag + "_obs": np.array(obs),
                    tag + "_rwd": np.array(rwd), tag + "_done": np.array(dn), tag + "_hx": s.hx.copy()})
    save("lorenz", **out)


def job_init_data():
    """The reference's shipped developed-flow initial states (init_field.dat, text '%.5e'),
    parsed exactly as each env's load() does (rayleigh.py:356-362, shkadov.py:364-368,
    sloshing.py:309-313) and stored as float64 arrays: a DATA dependency of reset()."""
    f = np.loadtxt(os.path.join(REF, "beacon", "rayleigh", "init_field.dat"))
    n = f.shape[0] // 4
    ray = np.stack([f[k * n:(k + 1) * n, :] for k in range(4)])
    f = np.loadtxt(os.path.join(REF, "beacon", "shkadov", "init_field.dat"))
    shk = np.stack([f[:, 1], f[:, 2]])
    f = np.loadtxt(os.path.join(REF, "beacon", "sloshing", "init_field.dat"))
    slo = np.zeros((2, f.shape[0] + 2))
    slo[0, 1:-1], slo[1, 1:-1] = f[:, 1], f[:, 2]
    path = os.path.join(REPO, "beacon_amd", "data", "init_fields.npz")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.savez_compressed(path, rayleigh=ray, shkadov=shk, sloshing=slo)
    print("wrote", path, ray.shape, shk.shape, slo.shape)


def job_vortex():
    mod, d = load_ref("vortex")
    out = {}
    for tag in ("zero", "rnd"):
        s = mod.vortex()
        obs0, _ = s.reset()
        obs0 = obs0.copy()
        n = s.n_act if tag == "zero" else 120
        acts = np.zeros((n, 2)) if tag == "zero" else np.random.default_rng(500).uniform(-1.0, 1.0, (n, 2))
        obs, rwd, dn = [], [], []
        for k in range(n):
            o, r, done, trunc, _ = s.step(acts[k].copy())
            obs.append(o.copy()), rwd.append(r), dn.append([done, trunc])
        out.update({tag + "_reset_obs": obs0, tag + "_actions": acts, tag + "_obs": np.array(obs),
                    tag + "_rwd": np.array(rwd), tag + "_done": np.array(dn), tag + "_hx": s.hx.copy()})
    save("vortex", **out)


def job_shkadov_separable():
    """shkadov_separable (shkadov.py:376-481): 5 round-robin resets, then 3 rounds of 5 per-jet steps."""
    mod, d = load_ref("shkadov")
    with cwd(d):
        s = mod.shkadov_separable(n_jets=5)
    s.rand_init = False
    robs = [s.reset()[0].copy() for _ in range(5)]
    acts = np.random.default_rng(600).uniform(-1.0, 1.0, (3, 5))
    seed = 6
    np.random.seed(seed)
    obs, rwd, dn, stp = [], [], [], []
    for r in range(3):
        for j in range(5):
            o, rw, done, trunc, _ = s.step(acts[r].tolist())
            obs.append(o.copy()), rwd.append(rw), dn.append([done, trunc]), stp.append(s.stp)
    save("shkadov_separable", reset_obs=np.array(robs), actions=acts, obs=np.array(obs), rwd=np.array(rwd),
         done=np.array(dn), stp=np.array(stp), noise=noise_stream(seed, s.sigma, 3 * s.ndt_act).reshape(3, s.ndt_act),
         h=s.h.copy(), q=s.q.copy())


JOBS = {
    "shkadov_separable": job_shkadov_separable,
    "vortex": job_vortex,
    "init_data": job_init_data,
    "rayleigh_default": job_rayleigh_default,
    "rayleigh_128x64": job_rayleigh_128x64,
    "rayleigh_128x64_step0": lambda: job_rayleigh_128x64_step(0),
    "rayleigh_128x64_step1": lambda: job_rayleigh_128x64_step(1),
    "mixing_a0": lambda: job_mixing(0),
    "mixing_a1": lambda: job_mixing(1),
    "mixing_a2": lambda: job_mixing(2),
    "mixing_a3": lambda: job_mixing(3),
    "mixing_synth": job_mixing_synth,
    "burgers": job_burgers,
    "shkadov": job_shkadov,
    "ctor_args": job_ctor_args,
    "sloshing": job_sloshing,
    "lorenz": job_lorenz,
}

if __name__ == "__main__":
    names = sys.argv[1:]
    if names == ["all"] or not names:
        names = list(JOBS)
    for nm in names:
        JOBS[nm]()
