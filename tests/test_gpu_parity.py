"""GPU parity: the HIP path, called through the C ABI (beacon_amd.vec -> libbeacon_hip.so),
against (a) golden vectors captured from the reference and (b) the CPU oracle on the same
seeded inputs.  float64 kernels: tight tolerance (operation order differs only in reductions,
FMA contraction and the linearised transport sweep).  float32 kernels: the tolerances stated
next to each test, measured against the float64 reference.

Nothing here reads /root/reference."""
import numpy as np
import pytest
import torch

import os

from conftest import GOLD, golden, ref_to_dev
from oracle import oracle as O

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    import beacon_amd
    from beacon_amd import envs as E
    from beacon_amd import vec as V

F64_TOL = 1e-9       # f64 kernels vs f64 reference (absolute, fields are O(1))

# float32 tolerances: every one is <= 10 x the error MEASURED on that workload in round 4 (BEACON_ERRLOG=... pytest, then
# scripts/tolerance_report.py; the measured maxima are quoted next to each entry as "# m: u v p S obs rwd").  All are absolute
# errors against the float64 reference / oracle on fields of size O(1) (|T| <= 1.25, |u|, |v| <~ 0.3 for rayleigh, <= 1 for
# mixing; rewards: the Nusselt number, O(2-5), for rayleigh, a mean of |C - 0.25| for mixing).
def f32tol(u, v, p, S, obs, rwd):
    return dict(u=u, v=v, p=p, T=S, C=S, obs=obs, rwd=rwd)


F32 = {
    # 50x50, 2 full steps + a repeat: generic m: 6.9e-7 8.6e-7 4.0e-7 1.8e-6 1.3e-6 2.6e-6; register-resident m: 3.1e-7 3.9e-7 2.1e-7 9.0e-7 6.8e-7 4.0e-6
    ("ray_default", 0): f32tol(6e-6, 6e-6, 4e-6, 1.5e-5, 1e-5, 2.5e-5),
    ("ray_default", 1): f32tol(3e-6, 3e-6, 2e-6, 8e-6, 6e-6, 3e-5),
    # 128x64, 5 timesteps from the synthetic state (velocities still tiny): m: 1.5e-8 3.7e-8 2.8e-7 2.3e-7 1.2e-7 / 6.6e-9 4.4e-9 2.0e-7 2.7e-7 7.4e-8
    ("ray_128x64", 0): f32tol(1.5e-7, 3e-7, 2.5e-6, 2e-6, 1e-6, None),
    ("ray_128x64", 1): f32tol(6e-8, 4e-8, 2e-6, 2.5e-6, 7e-7, None),
    # 100x100 rayleigh (two rows per lane), 2 x 12 timesteps: m: 7.7e-9 7.7e-9 1.5e-7 3.1e-7 2.7e-7 3.2e-6
    "ray_100x100": f32tol(7e-8, 7e-8, 1.5e-6, 3e-6, 2.5e-6, 3e-5),
    # wide domains (100 / 150 / 200 x 50), 2 x 10 timesteps: m: 4.9e-9 4.3e-9 4.2e-8 3.1e-7 2.1e-7 3.7e-6
    "ray_wide": f32tol(5e-8, 5e-8, 4e-7, 3e-6, 2e-6, 3.5e-5),
    # on-demand grids, 30 timesteps; by kernel: one row per lane m: 6.4e-9 7.6e-9 1.5e-7 3.8e-7; two rows m: 1.4e-7 1.0e-7 9.8e-7 3.7e-7;
    # Poisson-only in registers (tall / wide) m: 5.4e-7 4.1e-7 7.4e-6 5.4e-7; observations 3.5e-7, rewards 6.2e-6 over all of them
    ("ray_jit", 1): f32tol(6e-8, 7e-8, 1.5e-6, 3.8e-6, 3.5e-6, 6e-5),
    ("ray_jit", 2): f32tol(1.4e-6, 1e-6, 9e-6, 3.7e-6, 3.5e-6, 6e-5),
    ("ray_jit", 4): f32tol(5e-6, 4e-6, 7e-5, 5e-6, 3.5e-6, 6e-5),
    # mixing 100x100 from rest, 3 timesteps: m: 1.1e-7 1.2e-7 3.9e-7 2.7e-7 9.2e-8 3.2e-8
    "mix_rest": f32tol(1e-6, 1e-6, 3.5e-6, 2.5e-6, 9e-7, 3e-7),
    # mixing B = 512, one full step (250 timesteps): m: 1.3e-6 2.2e-6 1.1e-5 2.5e-6 1.2e-6 2.1e-8
    "mix_bench": f32tol(1.3e-5, 2e-5, 1e-4, 2.5e-5, 1e-5, 2e-7),
    # on-demand mixing grids, 40 timesteps: two rows per lane m: 1.2e-7 1.2e-7 3.0e-7 8.3e-7 - 4.9e-8; 100x200 m: 7.6e-7 4.8e-7 1.4e-5 9.5e-7 - 1.6e-8;
    # 200x100 m: 5.4e-8 5.3e-8 1.7e-7 8.1e-7 - 3.6e-8
    ("mix_jit", 2): f32tol(1.2e-6, 1.2e-6, 3e-6, 8e-6, None, 4.5e-7),
    ("mix_jit", "100x130"): f32tol(1.2e-6, 1.2e-6, 3e-6, 8e-6, None, 4.5e-7),
    ("mix_jit", "100x200"): f32tol(7e-6, 4.5e-6, 1.4e-4, 9e-6, None, 4.5e-7),
    ("mix_jit", "106x200"): f32tol(1.2e-6, 1.2e-6, 2.5e-6, 8.8e-6, None, 3.4e-7),
    ("mix_jit", "200x100"): f32tol(5e-7, 5e-7, 1.7e-6, 8e-6, None, 4.5e-7),
}
DEV = "cuda:0"


def dev2ref(state):
    """[B,4,ny+2,nx+2] device state -> numpy [B,4,nx+2,ny+2] float64"""
    return np.swapaxes(state.detach().cpu().numpy().astype(np.float64), -1, -2)


def maxdiff(a, b):
    """max |a - b|.  With BEACON_ERRLOG=<file> every evaluation is appended there as a JSON line (test id, line of the
    assertion, value): scripts/tolerance_report.py turns that log into the table the float32 tolerances are set from
    (tolerance <= 10 x the measured error: VERDICT r03 item 2a)."""
    d = float(np.max(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))))
    log = os.environ.get("BEACON_ERRLOG")
    if log:
        import json
        import sys
        with open(log, "a") as fh:
            fh.write(json.dumps({"test": os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0].split("::")[-1],
                                 "line": sys._getframe(1).f_lineno, "pos": sys._getframe(1).f_lasti, "err": d}) + "\n")
    return d


# ---------------------------------------------------------------------------------------------
# rayleigh
# ---------------------------------------------------------------------------------------------
def _ray_init(g):
    return np.stack([g["u_init"], g["v_init"], g["p_init"], g["T_init"]])


def _variant(env, variant):
    """0 = generic kernel, 1 = register-resident kernel (skip when the grid/dtype has none)."""
    got = env.set_variant(variant)
    if got != variant:
        env.close()
        pytest.skip("no variant-%d kernel for this configuration" % variant)
    assert env.kernel_name == ("ns2d_fast_step" if variant else "ns2d_generic_step")


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("dtype,swtol", [("f64", 0), ("f32", 2)])
def test_rayleigh_default_vs_golden(dtype, swtol, variant):
    """50x50, shipped init state, 2 action steps (400 timesteps) + a=None repeat.
    f32 tolerances: F32["ray_default", variant] (measured errors x <= 10), sweeps within +-2."""
    tol = F32["ray_default", variant] if dtype == "f32" else f32tol(F64_TOL, F64_TOL, 20 * F64_TOL, F64_TOL, F64_TOL, 50 * F64_TOL)
    g = golden("rayleigh_default")
    B = 3
    env = V.VecRayleigh(B, DEV, dtype, _ray_init(g))
    _variant(env, variant)
    obs, info = env.reset()
    assert info is None
    assert maxdiff(obs.cpu().numpy()[0], g["reset_obs"]) <= (0 if dtype == "f64" else 1e-6)
    for k in range(3):
        a = np.tile(g["actions"][k], (B, 1)) if k < 2 else None
        obs, rwd, done, trunc, _ = env.step(a)
        st = env.check_status()
        assert not st.any()
        o = obs.cpu().numpy()
        assert maxdiff(o[0], o[1]) == 0 and maxdiff(o[0], o[2]) == 0      # replicas are independent and equal
        assert maxdiff(o[0], g["step%d_obs" % k]) <= tol["obs"]
        assert maxdiff(float(rwd[0]), float(g["step%d_rwd" % k])) <= tol["rwd"]
        fields = dev2ref(env.get_state())[0]
        for i, F in enumerate("uvpT"):
            assert maxdiff(fields[i], g["step%d_%s" % (k, F)]) <= tol[F], (k, F)
        sw = env.sweeps.cpu().numpy()[0]
        assert np.max(np.abs(sw - g["itp"][k])) <= swtol
        assert maxdiff(env.actions_norm.cpu().numpy()[0], g["step%d_a_norm" % k]) <= (1e-15 if dtype == "f64" else 1e-7)
        assert [bool(done[0]), bool(trunc[0])] == list(g["step%d_done" % k])
    env.close()


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_rayleigh_128x64_vs_golden(dtype, variant):
    """BASELINE grid, seeded synthetic state, 5 timesteps (first Poisson solve: 5375 sweeps).  f32: F32["ray_128x64", variant]."""
    tol = F32["ray_128x64", variant] if dtype == "f32" else f32tol(F64_TOL, F64_TOL, 50 * F64_TOL, F64_TOL, F64_TOL, None)
    g = golden("rayleigh_128x64")
    env = V.VecRayleigh(2, DEV, dtype, None, L=2.56, H=1.28)
    env.set_ndt_act(5)
    _variant(env, variant)
    env.reset()
    st0 = np.stack([ref_to_dev(g[k]) for k in ("u0", "v0", "p0", "T0")])
    env.set_state(np.tile(st0[None], (2, 1, 1, 1)))
    obs, rwd, done, trunc, _ = env.step(np.tile(g["actions"][0], (2, 1)))
    env.check_status()
    sw = env.sweeps.cpu().numpy()[0]
    ref_sw = g["itp"][0]
    assert np.all(np.abs(sw - ref_sw) <= (0 if dtype == "f64" else np.maximum(3, 0.01 * ref_sw)))
    fields = dev2ref(env.get_state())[0]
    for i, F in enumerate("uvpT"):
        assert maxdiff(fields[i], g["step0_" + F]) <= tol[F], F
    # obs: the history slots before this step hold zeros, the last slot the new samples
    assert maxdiff(obs.cpu().numpy()[0][-96:], g["step0_obs"][-96:]) <= tol["obs"]
    # bottom ghost cells right of the last segment are never written (rayleigh.py:199-202)
    assert maxdiff(fields[3][121:129, 0], g["T0"][121:129, 0]) == 0
    env.close()


@pytest.mark.parametrize("variant", [0, 1])
def test_rayleigh_batch_vs_oracle_f64(variant):
    """8 replicas, a different action vector each, 2 full steps, against the oracle per replica."""
    g = golden("rayleigh_default")
    B = 8
    rng = np.random.default_rng(42)
    acts = rng.uniform(-1, 1, (2, B, 10))
    acts[:, 0, :] = 0.0          # uncontrolled replica
    acts[:, 1, :] *= 3.0         # saturating actions (m > 1 branch of the conditioning)
    env = V.VecRayleigh(B, DEV, "f64", _ray_init(g))
    _variant(env, variant)
    env.reset()
    oracles = [O.rayleigh(init_fields=_ray_init(g)) for _ in range(B)]
    for o in oracles:
        o.reset()
    for k in range(2):
        obs, rwd, done, trunc, _ = env.step(acts[k])
        env.check_status()
        st = dev2ref(env.get_state())
        sw = env.sweeps.cpu().numpy()
        for b, o in enumerate(oracles):
            ob, rw, dn, tr, _ = o.step(acts[k, b].tolist())
            assert maxdiff(obs[b].cpu().numpy(), ob) <= F64_TOL
            assert maxdiff(float(rwd[b]), rw) <= 1e-8
            assert maxdiff(st[b][3], o.S) <= F64_TOL and maxdiff(st[b][0], o.u) <= F64_TOL
            assert np.max(np.abs(sw[b] - o.itp)) <= 1       # a replica at the threshold may take +-1
    env.close()


def test_rayleigh_fast_schedulers_match_single_launch():
    """The fast path splits a step into [0,Q) + LPT-ordered [Q,ndt) when replicas outnumber
    the CUs; forced here on a small batch: obs, rewards, sweep counts and interior fields must
    equal the unsplit run bit for bit (only the p ghost cells, rebuilt once per launch from the
    change of their interior neighbour, may differ by one rounding)."""
    import os
    import subprocess
    import sys
    code = (
        "import sys, numpy as np, torch; sys.path.insert(0, %r)\n"
        "from beacon_amd import vec as V\n"
        "from beacon_amd.envs import packaged_init\n"
        "env = V.VecRayleigh(24, 'cuda:0', 'f64', packaged_init('rayleigh'))\n"
        "assert env.set_variant(1) == 1\n"
        "env.set_sched(*[int(x) for x in sys.argv[2].split(',')])\n"
        "env.reset()\n"
        "a = np.random.default_rng(5).uniform(-1, 1, (2, 24, 10))\n"
        "for k in range(2): obs, rwd, *_ = env.step(a[k])\n"
        "env.check_status()\n"
        "np.save(sys.argv[1], np.concatenate([obs.cpu().numpy().ravel(), rwd.cpu().numpy(),"
        " env.get_state().cpu().numpy().ravel(), env.sweeps.cpu().numpy().ravel().astype(float)]))\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    # bcn_set_sched(mode, grid, q, lpt_min_batch) per handle (no environment variable changes the scheduling)
    for tag, sched in (("split", "1,0,0,2"), ("single", "0,0,0,0"), ("ticket", "2,5,0,0")):
        path = "/tmp/bcn_lpt_%s.npy" % tag
        r = subprocess.run([sys.executable, "-c", code, path, sched], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(path))
    n_obs, B = 24 * 192, 24
    for other in (outs[0], outs[2]):      # two-launch LPT split / ticketed chunk scheduler (5 persistent WGs) vs plain
        assert np.array_equal(other[:n_obs + B], outs[1][:n_obs + B])                  # obs, rwd
        assert np.array_equal(other[-B * 200:], outs[1][-B * 200:])                    # sweeps
        st = [o[n_obs + B:-B * 200].reshape(B, 4, 52, 52) for o in (other, outs[1])]
        for f in (0, 1, 3):
            assert np.array_equal(st[0][:, f], st[1][:, f])
        assert np.array_equal(st[0][:, 2, 1:-1, 1:-1], st[1][:, 2, 1:-1, 1:-1])
        assert np.max(np.abs(st[0][:, 2] - st[1][:, 2])) < 1e-12


@pytest.mark.parametrize("cfg,dtype", [("50", "f32"), ("50", "f64"), ("128", "f32")])
def test_rayleigh_fast_identical_replicas_are_bit_identical(cfg, dtype):
    """Race / hazard detector: 96 replicas with identical inputs must stay bit-identical, sweep counts
    included (a missing wait state in front of a hand-written DPP instruction once showed up exactly
    here, and nowhere else)."""
    B = 96
    if cfg == "50":
        env = V.VecRayleigh(B, DEV, dtype, E.packaged_init("rayleigh"))
    else:
        env = V.VecRayleigh(B, DEV, dtype, np.load(os.path.join(GOLD, "rayleigh_128x64_init.npz"))["fields"], L=2.56, H=1.28)
    _variant(env, 1)
    env.reset()
    rng = np.random.default_rng(0)
    for k in range(3):
        obs, *_ = env.step(np.tile(rng.uniform(-1, 1, (1, 10)), (B, 1)))
        st, sw = env.get_state(), env.sweeps
        assert int((st != st[0:1]).flatten(1).any(1).sum()) == 0
        assert int((sw != sw[0:1]).any(1).sum()) == 0
    env.close()


def test_rayleigh_100x100_fast2_vs_oracle_and_generic():
    """ns2d_fast2 (two rows per lane, uneven column strips) in its rayleigh instantiation: L = H = 2
    -> 100x100, f32, seeded synthetic start, 2 x 12 timesteps.  Against the f64 oracle per replica
    (F32["ray_100x100"]: measured errors x <= 10), against the generic kernel, and replicas with identical inputs
    must stay bit-identical (hand-written DPP).  Sweep counts within max(4, 4 %): the stop test
    looks at increments of ~1e-6 on a phi of O(0.1), so float32 rounding moves the crossing a little
    (a start with grid-scale noise in the velocities moves it by tens of percent: phi is O(1) there
    and the all-Neumann problem is only compatible to rounding)."""
    nx = ny = 100
    rng = np.random.default_rng(7)
    init = np.zeros((4, nx + 2, ny + 2))
    y = (np.arange(ny + 2) - 0.5) / ny
    x = (np.arange(nx + 2) - 0.5) / nx
    init[3] = (0.5 - y)[None, :] + 0.1 * np.sin(4 * np.pi * x)[:, None] * np.sin(np.pi * y)[None, :] \
        + 1e-3 * rng.standard_normal((nx + 2, ny + 2))
    B, NDT = 6, 12
    acts = rng.uniform(-1, 1, (2, B, 10))
    acts[:, 4:] = acts[:, 0:1]                      # replicas 4, 5 repeat replica 0
    env = V.VecRayleigh(B, DEV, "f32", init, L=2.0, H=2.0)
    env.set_ndt_act(NDT)
    _variant(env, 1)
    env.reset()
    state0 = env.get_state().clone()
    oracles = [O.rayleigh(init_fields=init, L=2.0, H=2.0) for _ in range(4)]
    for o in oracles:
        o.cfg.ndt_act = NDT
        o.reset()
    fast = []
    for k in range(2):
        obs, rwd, _, _, _ = env.step(acts[k])
        env.check_status()
        assert env.kernel_name == "ns2d_fast2_step"
        raw = env.get_state()
        st, sw = dev2ref(raw), env.sweeps.cpu().numpy()
        for b in (4, 5):
            assert bool((raw[b] == raw[0]).all()) and np.array_equal(sw[b], sw[0])
        for b, o in enumerate(oracles):
            ob, rw, _, _, _ = o.step(acts[k, b].tolist())
            for i, F in enumerate("uvpT"):
                assert maxdiff(st[b][i], o.st[i]) <= F32["ray_100x100"][F], (k, b, F)
            assert maxdiff(obs[b].cpu().numpy(), ob) <= F32["ray_100x100"]["obs"]
            assert maxdiff(float(rwd[b]), rw) <= F32["ray_100x100"]["rwd"]
            assert np.all(np.abs(sw[b] - o.itp) <= np.maximum(4, 0.04 * o.itp)), (sw[b], o.itp)
        fast.append((raw.clone(), sw.copy()))
    assert env.set_variant(0) == 0
    env.set_state(state0)
    env.set_stp(0)
    for k in range(2):
        env.step(acts[k])
        assert env.kernel_name == "ns2d_generic_step"
        raw = env.get_state()
        for i, F in enumerate("uvpT"):      # two float32 kernels, ghost cells included (measured: p 4.6e-6)
            assert maxdiff(raw[:, i].cpu().numpy(), fast[k][0][:, i].cpu().numpy()) <= (3e-5 if F == "p" else 2 * F32["ray_100x100"][F]), (k, F)
        assert np.all(np.abs(env.sweeps.cpu().numpy() - fast[k][1]) <= np.maximum(4, 0.04 * fast[k][1]))
    env.close()


def test_ns2d_generic_other_grids_vs_oracle_f64():
    """The generic kernel on grids no fast path covers (non-square cells of the domain, other segment
    counts, dx != dy is not reachable through the reference's constructors): rayleigh L=3, H=1 ->
    150x50 with 5 segments and Ra=5e3, mixing L=1, H=2 -> 100x200; float64 against the oracle."""
    rng = np.random.default_rng(11)
    nx, ny = 150, 50
    x, y = (np.arange(nx + 2) - 0.5) / nx, (np.arange(ny + 2) - 0.5) / ny
    init = np.zeros((4, nx + 2, ny + 2))
    init[3] = (0.5 - y)[None, :] + 0.1 * np.sin(6 * np.pi * x)[:, None] * np.sin(np.pi * y)[None, :]
    B, NDT = 3, 4
    acts = rng.uniform(-1, 1, (2, B, 5))
    env = V.VecRayleigh(B, DEV, "f64", init, L=3.0, H=1.0, n_sgts=5, ra=5.0e3)
    env.set_ndt_act(NDT)
    env.set_variant(0)          # this test is about the generic kernel (the grid would get a kernel plugin)
    env.reset()
    oracles = [O.rayleigh(init_fields=init, L=3.0, H=1.0, n_sgts=5, ra=5.0e3) for _ in range(B)]
    for o in oracles:
        o.cfg.ndt_act = NDT
        o.reset()
    for k in range(2):
        obs, rwd, _, _, _ = env.step(acts[k])
        env.check_status()
        assert env.kernel_name == "ns2d_generic_step"
        st, sw = dev2ref(env.get_state()), env.sweeps.cpu().numpy()
        for b, o in enumerate(oracles):
            ob, rw, _, _, _ = o.step(acts[k, b].tolist())
            for i, F in enumerate("uvpT"):
                assert maxdiff(st[b][i], o.st[i]) <= F64_TOL * (50 if F == "p" else 1), (k, b, F)
            assert maxdiff(obs[b].cpu().numpy(), ob) <= F64_TOL and maxdiff(float(rwd[b]), rw) <= 1e-8
            assert np.max(np.abs(sw[b] - o.itp)) <= 1
    env.close()

    env = V.VecMixing(2, DEV, "f64", L=1.0, H=2.0)
    env.set_ndt_act(2)
    env.reset()
    oracles = [O.mixing(L=1.0, H=2.0) for _ in range(2)]
    for o in oracles:
        o.cfg.ndt_act = 2
        o.reset()
    a = np.array([0, 2])
    obs, rwd, _, _, _ = env.step(a)
    env.check_status()
    st, sw = dev2ref(env.get_state()), env.sweeps.cpu().numpy()
    for b, o in enumerate(oracles):
        ob, rw, _, _, _ = o.step(int(a[b]))
        for i, F in enumerate("uvpC"):
            assert maxdiff(st[b][i], o.st[i]) <= F64_TOL * (50 if F == "p" else 1), (b, F)
        assert maxdiff(obs[b].cpu().numpy(), ob) <= F64_TOL and maxdiff(float(rwd[b]), rw) <= 1e-9
        assert np.max(np.abs(sw[b] - o.itp)) <= 1
    env.close()


def test_rayleigh_bench_workload_f32_vs_f64():
    """The float32 tolerance at the metric configuration: 128x64 developed flow, random actions, two full
    action steps (2 x 200 timesteps, ~94 Jacobi sweeps each) on the float32 and the float64 register-resident
    kernels.  Measured drift (scripts/f32_vs_f64.py): observations 1-4e-6, fields <= 4e-6, rewards <= 7e-6,
    sweep counts differ by at most 2 in <= 10 % of the timesteps; asserted with a 10x margin."""
    init = np.load(os.path.join(GOLD, "rayleigh_128x64_init.npz"))["fields"]
    B = 32
    acts = np.random.default_rng(1234).uniform(-1, 1, (2, B, 10))
    envs = {dt: V.VecRayleigh(B, DEV, dt, init, L=2.56, H=1.28) for dt in ("f32", "f64")}
    for e in envs.values():
        _variant(e, 1)
        e.reset()
    for k in range(2):
        out = {}
        for dt, e in envs.items():
            obs, rwd, _, _, _ = e.step(acts[k])
            e.check_status()
            out[dt] = (obs.double().cpu(), rwd.double().cpu(), e.get_state().double().cpu(), e.sweeps.cpu().numpy())
        assert float((out["f32"][0] - out["f64"][0]).abs().max()) < 5e-5
        assert float((out["f32"][1] - out["f64"][1]).abs().max()) < 1e-4
        assert float((out["f32"][2] - out["f64"][2]).abs().max()) < 5e-5
        assert int(np.abs(out["f32"][3] - out["f64"][3]).max()) <= 4
    for e in envs.values():
        e.close()


# float32 drift over a whole episode, measured in round 4 (scripts/tolerance_report.py on this test's log; profile over the
# steps: scripts/episode_drift.py).  rayleigh: the largest observation difference of the 32 replicas is 2.5e-3, reached by ONE
# replica in a transient around step 51 (the median replica stays at 3e-5) and back at 7e-5 when the episode ends; the two
# float64 kernels (generic / register-resident: the same arithmetic in another order) drift apart with the SAME profile, from
# 4e-15 to 1e-11 at that step -- the flow amplifies rounding-level differences ~3000 x there, whatever their size, and float32
# sits at 0.45 of that curve scaled by eps32 / eps64 = 2^29.
# mixing: the float64 twin does not drift at all (4e-15 over the episode: this flow amplifies nothing), float32 accumulates
# about linearly -- 2.5e-6 after one step (test_mixing_bench_dispatch_b512_vs_oracle_and_scheduler), 1.8e-4 (median replica) /
# 8.5e-4 (worst probe of the worst replica) after 100, rewards 1e-5 -- and with fields 1e-4 apart the loosely converged solves
# (tol = 1e-4 on the increment norm, mixing.py:423) differ by up to 62 % in the sweeps of a single timestep.
# Measured -> asserted (worst replica / median replica / reward / at the end: observation, reward / sweeps / eps-scaled ratio):
#   rayleigh 2.5e-3, 2.3e-5, 2.7e-3, 7.2e-5, 1.5e-5, 0.125, 0.45;  mixing 8.5e-4, 1.8e-4, 1.05e-5, 6.2e-4, 9.2e-6, 0.625, -
EPISODE_TOL = {"rayleigh": dict(obs=1e-2, obs_median=2e-4, rwd=1e-2, obs_end=5e-4, rwd_end=1e-4, sweeps_rel=0.25, eps_scaled=4.0),
               "mixing": dict(obs=5e-3, obs_median=1e-3, rwd=1e-4, obs_end=4e-3, rwd_end=8e-5, sweeps_rel=1.0, eps_scaled=None)}


@pytest.mark.parametrize("kind", ["rayleigh", "mixing"])
def test_full_episode_drift_f32_vs_f64(kind):
    """north_star's "stated fp32 tolerance" over a WHOLE episode, not per step: 100 action steps (rayleigh.py:48-50,
    mixing.py:43-45: n_act = 100), B = 32, the float32 kernel (default stop rule) against the float64 kernel (the
    reference's arithmetic; itself within 1e-9 of the reference with equal sweep counts) on the same inputs --
    rayleigh 128x64 from the developed state with bench.py's action law, mixing 100x100 from rest with random wall
    choices.  A per-step error of 1e-6 does not stay 1e-6 over 20 000 timesteps of a forced flow; how much of the drift is
    the flow's own sensitivity shows in a float64 TWIN: the generic float64 kernel computes the same arithmetic in another
    order (rounding-level differences, injected every timestep like float32's), and float32's drift must stay within a
    stated multiple of the twin's scaled by eps32 / eps64.  Asserted: largest observation / reward difference over the
    episode (worst replica, median replica) and at its end, largest relative per-timestep sweep-count difference, the
    eps-scaled ratio, and that all runs end the episode at the same step."""
    B, N = 32, 100
    tol = EPISODE_TOL[kind]
    if kind == "rayleigh":
        init = np.load(os.path.join(GOLD, "rayleigh_128x64_init.npz"))["fields"]
        acts = np.random.default_rng(1234).uniform(-1.0, 1.0, (N, B, 10))
        envs = {dt: V.VecRayleigh(B, DEV, dt[:3], init, L=2.56, H=1.28) for dt in ("f32", "f64", "f64twin")}
    else:
        acts = np.random.default_rng(1234).integers(0, 4, (N, B))
        envs = {dt: V.VecMixing(B, DEV, dt[:3]) for dt in ("f32", "f64", "f64twin")}
    for dt, e in envs.items():
        v = 0 if dt == "f64twin" else 1
        assert e.set_variant(v) == v
        e.reset()
    eo, er, es, to = [], [], [], []
    for k in range(N):
        out = {}
        for dt, e in envs.items():
            obs, rwd, done, trunc, _ = e.step(acts[k])
            out[dt] = (obs.double().cpu().numpy(), rwd.double().cpu().numpy(), done.cpu().numpy().copy(),
                       trunc.cpu().numpy().copy(), e.sweeps.cpu().numpy().astype(np.float64))
        for dt in ("f32", "f64twin"):
            assert np.array_equal(out[dt][2], out["f64"][2]) and np.array_equal(out[dt][3], out["f64"][3])
        assert bool(out["f32"][2].all()) == (k == N - 1)
        eo.append(np.abs(out["f32"][0] - out["f64"][0]).max(axis=1))          # per replica
        er.append(float(np.abs(out["f32"][1] - out["f64"][1]).max()))
        es.append(float((np.abs(out["f32"][4] - out["f64"][4]) / np.maximum(out["f64"][4], 16.0)).max()))
        to.append(float(np.abs(out["f64twin"][0] - out["f64"][0]).max()))
    for e in envs.values():
        e.check_status()
        e.close()
    eo = np.array(eo)                                                          # [step, replica]
    got = dict(obs=float(eo.max()), obs_median=float(np.median(eo.max(axis=0))), rwd=max(er), obs_end=float(eo[-1].max()),
               rwd_end=er[-1], sweeps_rel=max(es), eps_scaled=float(eo.max()) / (2.0 ** 29 * max(max(to), 1e-300)))
    maxdiff(max(to), 0.0)                                                      # (logged: the float64 twin's own drift)
    bad = {k_: (v, tol[k_]) for k_, v in got.items() if tol[k_] is not None and not maxdiff(v, 0.0) <= tol[k_]}
    assert not bad, (bad, got)


# ---------------------------------------------------------------------------------------------
# the dispatch bench.py times: BASELINE configs[3] exactly as bench.py builds it
# ---------------------------------------------------------------------------------------------
def _bench_workload(B, nsteps, dtype):
    """bench.py's inputs: developed 128x64 state, default_rng(1234).uniform(-1, 1, (steps, B, 10))."""
    init = np.load(os.path.join(GOLD, "rayleigh_128x64_init.npz"))["fields"]
    acts = np.random.default_rng(1234).uniform(-1.0, 1.0, (nsteps, B, 10))
    env = V.VecRayleigh(B, DEV, dtype, init, L=2.56, H=1.28)
    env.reset()
    return env, init, acts


def _oracle_batch_step(init, acts, nrep, L=2.56, H=1.28):
    """One full action step of replicas 0..nrep-1 on the float64 C oracle (OpenMP over envs)."""
    import ctypes as C
    e = O.rayleigh(init=False, L=L, H=H)
    n = (e.cfg.nx + 2) * (e.cfg.ny + 2)
    st = np.zeros((nrep, 8, n))
    st[:, :4] = init.reshape(1, 4, n)
    e0 = O.rayleigh(init_fields=init, L=L, H=H)
    obs = np.tile(e0.reset()[0], (nrep, 1))          # observation history after reset(): [0, 0, 0, reset sample]
    rwd = np.zeros(nrep)
    sw = np.zeros(nrep, dtype=np.int64)
    a = np.ascontiguousarray(acts[:nrep].astype(np.float64))
    O.lib().orc_ns2d_step_batch(C.byref(e.cfg), nrep, O.dp(st.reshape(-1)), O.dp(a.reshape(-1)), a.shape[1],
                                O.dp(obs.reshape(-1)), O.dp(rwd), sw.ctypes.data_as(C.POINTER(C.c_int64)),
                                min(nrep, os.cpu_count() or 1))
    return st[:, :4].reshape(nrep, 4, e.cfg.nx + 2, e.cfg.ny + 2), obs, rwd, sw


@pytest.mark.parametrize("dtype,ftol,swtol", [("f32", 1e-5, 3), ("f64", F64_TOL, 1)])
def test_rayleigh_bench_dispatch_vs_oracle_and_reference(dtype, ftol, swtol):
    """The timed dispatch itself -- B=512, 128x64, 200 timesteps, default scheduler: `ns2d_fast_sched` -- against
    (a) the float64 C oracle on replicas 0..7 over one FULL action step and (b) the reference's own full step for
    replicas 0 and 1 (tests/golden/rayleigh_128x64_step{0,1}.npz, captured by oracle/capture/capture.py).
    float32 tolerance, stated against the float64 reference and set from scripts/f32_errors.py (round 3: u, v 4e-7,
    T and p 1.3e-6, observations 1.2e-6, reward 4e-6, every sweep count equal): fields/obs 1e-5 absolute (|T| <= 1.25,
    |u|,|v| < 0.3), p 2e-5 (200 accumulated phi), reward 2e-5, total sweeps of the step within 0.2 % and per timestep
    within 3; float64: 1e-9 with equal sweep counts (+-1 at the threshold)."""
    pf = 2 if dtype == "f32" else 50        # p: float32 measured (above); float64 keeps the generous factor of 1e-9
    B, NREP = 512, 8
    env, init, acts = _bench_workload(B, 1, dtype)
    obs, rwd, done, trunc, _ = env.step(acts[0])
    env.check_status()
    assert env.kernel_name == "ns2d_fast_sched"
    st = dev2ref(env.get_state()[:NREP])
    sw = env.sweeps.cpu().numpy()[:NREP]
    o = obs.cpu().numpy().astype(np.float64)[:NREP]
    r = rwd.cpu().numpy().astype(np.float64)[:NREP]
    ost, oobs, orwd, osw = _oracle_batch_step(init, acts[0], NREP)
    for b in range(NREP):
        for i, F in enumerate("uvpT"):
            assert maxdiff(st[b][i], ost[b][i]) <= ftol * (pf if F == "p" else 1), (b, F)
        assert maxdiff(o[b], oobs[b]) <= ftol and maxdiff(r[b], orwd[b]) <= max(1e-8, 2 * ftol)
        assert maxdiff(int(sw[b].sum()), int(osw[b])) <= max(swtol, 0.002 * osw[b] if dtype == "f32" else 0), (b, sw[b].sum(), osw[b])
    for b in (0, 1):                      # the reference itself
        g = golden("rayleigh_128x64_step%d" % b)
        assert np.array_equal(g["action"], acts[0, b])
        for i, F in enumerate("uvpT"):
            assert maxdiff(st[b][i], g[F]) <= ftol * (pf if F == "p" else 1), (b, F)
        # the capture steps from loaded fields without reset()'s get_obs: only the newest history slot is comparable
        assert maxdiff(o[b][-96:], g["obs"][-96:]) <= ftol and maxdiff(r[b], float(g["rwd"])) <= max(1e-8, 2 * ftol)
        assert np.max(np.abs(sw[b] - g["itp"])) <= swtol, (b, np.max(np.abs(sw[b] - g["itp"])))
    env.close()


@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_rayleigh_bench_dispatch_scheduler_is_bit_exact(dtype):
    """B=512 bench workload, two action steps: the ticketed chunk scheduler (mode 2, the timed dispatch) against
    one workgroup per replica in one launch (mode 0) -- obs, rewards, sweep counts and fields bit for bit (p ghost
    cells, rebuilt once per chunk from the change of their interior neighbour, to one rounding)."""
    outs = []
    for mode in (2, 0):
        env, init, acts = _bench_workload(512, 2, dtype)
        env.set_sched(mode)
        for k in range(2):
            obs, rwd, _, _, _ = env.step(acts[k])
        env.check_status()
        assert env.kernel_name == ("ns2d_fast_sched" if mode == 2 else "ns2d_fast_step")
        outs.append((obs.clone(), rwd.clone(), env.sweeps.clone(), env.get_state().clone()))
        env.close()
    a, b = outs
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    for f in (0, 1, 3):
        assert torch.equal(a[3][:, f], b[3][:, f])
    assert torch.equal(a[3][:, 2, 1:-1, 1:-1], b[3][:, 2, 1:-1, 1:-1])
    assert float((a[3][:, 2] - b[3][:, 2]).abs().max()) < (1e-4 if dtype == "f32" else 1e-12)


@pytest.mark.parametrize("dtype,swrel", [("f32", 0.02), ("f64", 0.0)])
def test_mixing_bench_dispatch_b512_vs_oracle_and_scheduler(dtype, swrel):
    """mixing-v0 at BASELINE configs[4]'s batch: B=512, 100x100, one full 250-timestep step from rest through
    `ns2d_fast2_sched` (256 persistent workgroups): replicas 0..3 (actions 0..3) against the float64 C oracle
    (float32: F32["mix_bench"] -- measured u, v, C 2.5e-6, p 1.1e-5, observations 1.2e-6, rewards 2e-8, at most two
    sweeps' difference in a timestep; sweeps within 3 or 2 %), and the whole batch bit for bit against the
    unscheduled launch."""
    B = 512
    a = (np.arange(B) % 4).astype(np.int64)
    outs = []
    for mode in (2, 0):
        env = V.VecMixing(B, DEV, dtype)
        env.set_sched(mode)
        env.reset()
        obs, rwd, _, _, _ = env.step(a)
        env.check_status()
        assert env.kernel_name == ("ns2d_fast2_sched" if mode == 2 else "ns2d_fast2_step")
        outs.append((obs.clone(), rwd.clone(), env.sweeps.clone(), env.get_state().clone()))
        env.close()
    x, y = outs
    assert torch.equal(x[0], y[0]) and torch.equal(x[1], y[1]) and torch.equal(x[2], y[2])
    for f in (0, 1, 3):
        assert torch.equal(x[3][:, f], y[3][:, f])
    assert torch.equal(x[3][:, 2, 1:-1, 1:-1], y[3][:, 2, 1:-1, 1:-1])
    st = dev2ref(x[3][:4])
    sw = x[2].cpu().numpy()
    for b in range(4):
        o = O.mixing()
        o.reset()
        ob, rw, _, _, _ = o.step(int(a[b]))
        tol = F32["mix_bench"] if dtype == "f32" else f32tol(F64_TOL, F64_TOL, 50 * F64_TOL, F64_TOL, F64_TOL, 1e-9)
        for i, F in enumerate("uvpC"):
            assert maxdiff(st[b][i], o.st[i]) <= tol[F], (b, F)
        assert maxdiff(x[0][b].cpu().numpy(), ob) <= tol["obs"] and maxdiff(float(x[1][b]), rw) <= tol["rwd"]
        assert np.all(np.abs(sw[b] - o.itp) <= np.maximum(1 if dtype == "f64" else 3, swrel * o.itp)), b
        assert torch.equal(x[0][b], x[0][b + 4])          # same action -> same replica, whatever CU ran it


def test_rayleigh_bench_episode_is_deterministic_with_staggered_resets():
    """(was scripts/soak.py + scripts/stress.py) 24 action steps of the bench configuration, B=512, episode
    counters staggered so that replicas end their episode at different steps and are auto-reset one by one through
    masks and the ticket scheduler; run twice: bit-identical, every status word 0, reset replicas restart from the
    developed state."""
    init = np.load(os.path.join(GOLD, "rayleigh_128x64_init.npz"))["fields"]
    B, N = 512, 24
    acts = torch.as_tensor(np.random.default_rng(11).uniform(-1, 1, (8, B, 10)), dtype=torch.float32, device=DEV)
    outs = []
    for run in range(2):
        env = V.VecRayleigh(B, DEV, "f32", init, L=2.56, H=1.28)
        obs0, _ = env.reset()
        obs0 = obs0.clone()
        env.set_stp((np.arange(B) * 7) % env.n_act)            # staggered episode ends
        resets, rw = 0, []
        for k in range(N):
            obs, rwd, done, trunc, _ = env.step(acts[k % 8])
            rw.append(rwd.clone())
            n = int(done.sum().item())
            if n:
                d = done.clone().bool()
                env.reset_done()
                resets += n
                assert torch.equal(env.obs[d], obs0[d])        # back to the reset observation
        st = env.check_status()
        assert (st == 0).all() and resets > 0
        assert env.kernel_name == "ns2d_fast_sched"
        outs.append((env.get_state().clone(), torch.stack(rw), env.obs.clone(), resets))
        env.close()
    assert outs[0][3] == outs[1][3]
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])


def test_convergence_plan_never_skips_a_passing_sweep():
    """The register-resident kernels evaluate the Jacobi residual only on the sweeps that can pass the test
    (ns2d_fast.hip: a proven lower bound of the norm, conv_plan 1, the float64 default; plus an extrapolation of the
    norm's observed decay, conv_plan 2, the float32 default).  "verify_conv" evaluates every sweep and raises
    BCN_ST_PLAN (4) if a sweep the plan would have skipped passes: it must never do so, and sweep counts and fields must
    equal those of the planned run and of conv_plan 0 (every sweep evaluated, as the reference does) bit for bit --
    bench workload (B=512, f32, 2 steps = 205 000 solves), the reference's 50x50 vectors (f64) and the synthetic 128x64
    state whose first solve takes 5375 sweeps."""
    def run(verify, plan=None):
        out = []
        env, init, acts = _bench_workload(512, 2, "f32")
        env.set_option("verify_conv", verify)
        if plan is not None:
            env.set_option("conv_plan", plan)
        for k in range(2):
            env.step(acts[k])
        out.append((env.status.clone(), env.sweeps.clone(), env.get_state().clone()))
        env.close()
        g = golden("rayleigh_default")
        for p in ([plan] if plan is not None else [1, 2]):
            env = V.VecRayleigh(4, DEV, "f64", _ray_init(g))
            _variant(env, 1)
            env.set_option("verify_conv", verify)
            env.set_option("conv_plan", p)
            env.reset()
            for k in range(2):
                env.step(np.tile(g["actions"][k], (4, 1)))
            assert np.array_equal(env.sweeps.cpu().numpy()[0], g["itp"][1])
            out.append((env.status.clone(), env.sweeps.clone(), env.get_state().clone()))
            env.close()
        g = golden("rayleigh_128x64")
        for p in ([plan] if plan is not None else [1, 2]):
            env = V.VecRayleigh(2, DEV, "f64", None, L=2.56, H=1.28)
            env.set_ndt_act(5)
            _variant(env, 1)
            env.set_option("verify_conv", verify)
            env.set_option("conv_plan", p)
            env.reset()
            st0 = np.stack([ref_to_dev(g[k]) for k in ("u0", "v0", "p0", "T0")])
            env.set_state(np.tile(st0[None], (2, 1, 1, 1)))
            env.step(np.tile(g["actions"][0], (2, 1)))
            assert np.array_equal(env.sweeps.cpu().numpy()[0], g["itp"][0])
            out.append((env.status.clone(), env.sweeps.clone(), env.get_state().clone()))
            env.close()
        for p in ([plan] if plan is not None else [1, 2]):     # mixing from rest: solves of up to 2466 sweeps
            env = V.VecMixing(8, DEV, "f32")
            env.set_ndt_act(40)
            env.set_option("verify_conv", verify)
            env.set_option("conv_plan", p)
            env.reset()
            env.step(np.arange(8) % 4)
            out.append((env.status.clone(), env.sweeps.clone(), env.get_state().clone()))
            env.close()
        return out
    planned, verified = run(0), run(1)
    for (st_p, sw_p, f_p), (st_v, sw_v, f_v) in zip(planned, verified):
        assert int(st_v.max()) == 0 and int(st_p.max()) == 0          # no BCN_ST_PLAN, no overflow
        assert torch.equal(sw_p, sw_v) and torch.equal(f_p, f_v)
    literal = run(0, plan=0)                                          # every sweep evaluated, as the reference does
    for i, j in ((0, 0), (1, 1), (2, 1), (3, 2), (4, 2), (5, 3), (6, 3)):
        assert torch.equal(planned[i][1], literal[j][1]) and torch.equal(planned[i][2], literal[j][2])


# workloads of the wider plan checks: name -> (constructor, driver); every driver returns the tensors to compare
def _plan_workloads():
    def jit_grid(L, H):
        def make():
            env = V.VecRayleigh(6, DEV, "f32", None, L=L, H=H)
            env.set_ndt_act(30)
            assert env.set_variant(1) == 1 and getattr(env, "_plugin", None) is not None
            return env

        def drive(env):
            x, y = (np.arange(env.nx + 2) - 0.5) / env.nx, (np.arange(env.ny + 2) - 0.5) / env.ny
            st0 = np.zeros((4, env.nx + 2, env.ny + 2))
            st0[3] = (0.5 - y)[None, :] + 0.08 * np.sin(2 * np.pi * x * L)[:, None] * np.sin(np.pi * y)[None, :]
            env.reset()
            env.set_state(np.tile(ref_to_dev(st0)[None], (6, 1, 1, 1)))
            acts = np.random.default_rng(3).uniform(-1, 1, (2, 6, 10))
            sw = []
            for k in range(2):
                env.step(acts[k])
                sw.append(env.sweeps.clone())
            return sw
        return make, drive

    def ra_sgts(ra, n_sgts):
        def make():
            g = golden("rayleigh_default")
            env = V.VecRayleigh(8, DEV, "f32", _ray_init(g), n_sgts=n_sgts, ra=ra)
            assert env.set_variant(1) == 1
            return env

        def drive(env):
            env.reset()
            env.step(np.random.default_rng(5).uniform(-1, 1, (8, n_sgts)))
            return [env.sweeps.clone()]
        return make, drive

    def mixing_full():
        def make():
            return V.VecMixing(512, DEV, "f32")

        def drive(env):
            env.reset()
            env.step(np.arange(512) % 4)
            assert env.kernel_name == "ns2d_fast2_sched"
            return [env.sweeps.clone()]
        return make, drive

    def episode():
        def make():
            init = np.load(os.path.join(GOLD, "rayleigh_128x64_init.npz"))["fields"]
            return V.VecRayleigh(512, DEV, "f32", init, L=2.56, H=1.28)

        def drive(env):
            acts = torch.as_tensor(np.random.default_rng(11).uniform(-1, 1, (8, 512, 10)), dtype=torch.float32, device=DEV)
            env.reset()
            env.set_stp((np.arange(512) * 7) % env.n_act)
            sw = []
            for k in range(24):
                obs, rwd, done, trunc, _ = env.step(acts[k % 8])
                sw.append(env.sweeps.sum(1).clone())
                if int(done.sum().item()):
                    env.reset_done()
            assert env.kernel_name == "ns2d_fast_sched"
            return sw
        return make, drive

    # (Ra = 5e3 is not a valid configuration of the REFERENCE: at dt = 0.01 its explicit scalar transport has a diffusion
    # number of 0.84 there and the float64 reference itself blows up to NaN around timestep 60 -- sweep counts 1195,
    # 15950, 43272, then 1 for ever -- so the low-Ra case is Ra = 2e4)
    return {"jit75x50": jit_grid(1.5, 1.0), "jit110x64": jit_grid(2.2, 1.28), "jit60x120": jit_grid(1.2, 2.4),
            "ra2e4_2sgts": ra_sgts(2.0e4, 2), "ra5e4_5sgts": ra_sgts(5.0e4, 5), "ra1e4_7sgts": ra_sgts(1.0e4, 7),
            "mixing_full_b512": mixing_full(), "episode24_b512": episode()}


@pytest.mark.parametrize("name", ["jit75x50", "jit110x64", "jit60x120", "ra2e4_2sgts", "ra5e4_5sgts", "ra1e4_7sgts",
                                  "mixing_full_b512", "episode24_b512"])
def test_float32_stop_rule_on_more_workloads(name):
    """The float32 default (conv_plan 3: the extrapolating plan, guarded; speculative jump for rayleigh) on workloads the
    first plan test does not reach: three on-demand grids (75x50, 110x64, 60x120), Ra = 2e4 / 5e4 / 1e4 with 2 / 5 / 7
    bottom segments, a FULL 250-timestep mixing step at B=512 through the ticket scheduler, and the 24-step episode with
    staggered auto-resets.  For each:
    (a) sweep counts, fields and observations of the default equal, BIT FOR BIT, those of conv_plan 0 (every sweep
        evaluated, as rayleigh.py:448-454 does), of the proven plan 1 and of the run without the speculative jump;
    (b) verify_conv (every sweep evaluated next to the plan) gives the same, and flags BCN_ST_PLAN only in replicas whose
        late-stop counter is non-zero, i.e. every stop the extrapolation did not foresee was caught by the guard and
        repeated under the proven plan;
    (c) the UNGUARDED extrapolation (conv_plan 2, round 2's default) never stops early, and differs from the reference's
        stop sweep only in replicas with late stops -- where there are none it is bit-identical too."""
    make, drive = _plan_workloads()[name]

    def run(**opts):
        env = make()
        for k, v in opts.items():
            env.set_option(k, v)
        sw = drive(env)
        torch.cuda.synchronize()
        st = env.status.cpu().numpy().copy()
        assert not (st & 1).any()                                           # no overflow
        out = (sw, env.get_state().clone(), env.obs.clone(), st, env.get_counters())
        env.close()
        return out

    def same(a, b):
        return all(torch.equal(x, y) for x, y in zip(a[0], b[0])) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])

    default, literal = run(), run(conv_plan=0)
    assert int(default[3].max()) == 0 and int(literal[3].max()) == 0
    assert same(default, literal), name
    for tag, opts in {"nojump": dict(spec_start=0), "proven": dict(conv_plan=1)}.items():
        got = run(**opts)
        assert int(got[3].max()) == 0 and same(default, got), (name, tag)
    ver = run(verify_conv=1)
    assert same(default, ver), name
    flagged = (ver[3] & 4) != 0
    assert not (flagged & (ver[4][:, 2] == 0)).any(), (name, "a skipped sweep passed and the guard did not see it")
    late = default[4][:, 2]
    plan2 = run(conv_plan=2)
    assert int(plan2[3].max()) == 0
    for a, b in zip(plan2[0], literal[0]):
        assert int((a < b).sum()) == 0, (name, "the extrapolating plan stopped early")
    if int(late.sum()) == 0:
        assert same(plan2, literal), name
    print("%s: late stops in the last step %d (replicas %d of %d), repeated timesteps %d"
          % (name, int(late.sum()), int((late > 0).sum()), late.size, int(default[4][:, 3].sum())))


@pytest.mark.parametrize("kind", ["rayleigh", "mixing"])
def test_conv_plan_3_repeats_late_stops_under_the_proven_plan(kind):
    """The guard itself.  "plan_overshoot" (a test hook) lengthens every skip of the extrapolating plan by 12 sweeps, so
    its evaluations land behind the reference's stop sweep: plan 2 then stops late (larger sweep counts, late stops
    counted), plan 3 must notice each of them, repeat the solve under the proven plan and return exactly the counts and
    fields of conv_plan 0."""
    def run(plan, over):
        if kind == "rayleigh":
            env, init, acts = _bench_workload(64, 1, "f32")
            env.set_ndt_act(40)
            env.reset()
            a = acts[0]
        else:
            env = V.VecMixing(8, DEV, "f32")
            env.set_ndt_act(40)
            env.reset()
            a = np.arange(8) % 4
        env.set_option("conv_plan", plan)
        env.set_option("plan_overshoot", over)
        env.step(a)
        env.check_status()
        out = (env.sweeps.clone(), env.get_state().clone(), env.get_counters())
        env.close()
        return out
    literal, plan2, plan3 = run(0, 0), run(2, 12), run(3, 12)
    assert int(plan2[2][:, 2].sum()) > 0                                                 # the hook does provoke late stops
    p2, lit = plan2[0].cpu().numpy(), literal[0].cpu().numpy()
    nlate = 0
    for b in range(lit.shape[0]):                       # a replica's FIRST differing solve is a late one (never early);
        d = np.nonzero(p2[b] != lit[b])[0]              # behind it the fields differ by a few sweeps' worth of phi
        if len(d):
            assert p2[b, d[0]] > lit[b, d[0]], (b, d[0], p2[b, d[0]], lit[b, d[0]])
            nlate += 1
    assert nlate > 0
    assert int(plan3[2][:, 2].sum()) > 0 and int(plan3[2][:, 3].sum()) >= int(plan3[2][:, 2].sum())
    assert torch.equal(plan3[0], literal[0]) and torch.equal(plan3[1], literal[1])


def test_float32_default_plan_verified_over_a_long_bench_episode():
    """the float32 default (conv_plan 3 with the slow-mode landing guard) is proven for exact arithmetic; float32
    rounding of the increments is the size of the guard's distance from tol, so the default is VERIFIED on the bench workload
    over a long stretch: 16 action steps = 3 200 timesteps x 256 replicas = 819 200 solves with every sweep evaluated next to
    the plan (verify_conv): no replica may raise BCN_ST_PLAN (a sweep the plan skipped passed the test), none may overflow, and
    sweep counts and fields must equal the planned run's bit for bit."""
    def run(verify):
        env, init, acts = _bench_workload(256, 16, "f32")
        env.set_option("verify_conv", verify)
        flags, sw = 0, []
        for k in range(16):
            env.step(acts[k])
            flags |= int(env.status.max())
            sw.append(env.sweeps.clone())
        out = (flags, torch.stack(sw), env.get_state().clone(), env.get_counters())
        env.close()
        return out
    planned, verified = run(0), run(1)
    assert planned[0] == 0 and verified[0] == 0, (planned[0], verified[0])
    assert torch.equal(planned[1], verified[1]) and torch.equal(planned[2], verified[2])
    assert int(planned[1].min()) >= 1 and int(planned[1].max()) > 60


@pytest.mark.parametrize("kind,dtype", [("rayleigh", "f32"), ("rayleigh", "f64"), ("mixing", "f32")])
def test_tall_grid_evaluation_plans_never_change_a_result(kind, dtype):
    """ns2d_fast4_impl.h (ny > 128: rayleigh 50x150, mixing 100x200) evaluates the Jacobi residual only where its plan says
    the stop test can pass.  The plans choose WHICH sweeps are evaluated, never the arithmetic: conv_plan 1 (proven bound),
    2 (extrapolated), 3 (extrapolated, late stops repeated under 1) must return the sweep counts and fields of conv_plan 0
    (every sweep, as the reference: rayleigh.py:448-454) bit for bit, with no late stop; verify_conv (every sweep evaluated
    next to the plan) must flag nothing under the proven plan; the speculative opening of a solve (spec_start) must be inert in
    the results whether it lands short or too far; and with the test hook plan_overshoot = 12 plan 2 stops late
    (counted) while plan 3 notices, repeats those solves and still returns the result of conv_plan 0."""
    def run(plan, over=0, verify=0, spec=None):
        if kind == "rayleigh":
            env = V.VecRayleigh(6, DEV, dtype, None, L=1.0, H=3.0)
            rng = np.random.default_rng(5)
            x, y = (np.arange(env.nx + 2) - 0.5) / env.nx, (np.arange(env.ny + 2) - 0.5) / env.ny
            st0 = np.zeros((4, env.nx + 2, env.ny + 2))
            st0[3] = (0.5 - y)[None, :] + 0.08 * np.sin(2 * np.pi * x)[:, None] * np.sin(np.pi * y)[None, :]
            env.set_ndt_act(30)
            env.reset()
            env.set_state(np.tile(ref_to_dev(st0)[None], (6, 1, 1, 1)))
            a = rng.uniform(-1, 1, (6, 10))
        else:
            env = V.VecMixing(4, DEV, dtype, L=1.0, H=2.0)
            env.set_ndt_act(30)
            env.reset()
            a = np.arange(4)
        assert env.set_variant(1) == 1
        env.set_option("conv_plan", plan)
        env.set_option("plan_overshoot", over)
        env.set_option("verify_conv", verify)
        if spec is not None:
            env.set_option("spec_start", spec)
        env.step(a)
        assert env.kernel_name == "ns2d_fast4_step"
        torch.cuda.synchronize()
        st = env.status.cpu().numpy().copy()
        assert not (st & 1).any()                                           # no overflow
        out = (env.sweeps.clone(), env.get_state().clone(), env.get_counters(), st)
        env.close()
        return out
    literal = run(0)
    assert int(literal[0].max()) > 20
    for plan in (1, 2, 3):
        got = run(plan)
        assert torch.equal(got[0], literal[0]) and torch.equal(got[1], literal[1]), plan
        assert int(got[2][:, 2].sum()) == 0, plan                                        # no late stop
    ver = run(1, verify=1)
    assert torch.equal(ver[0], literal[0]) and not np.any(ver[3] & 4)
    # the speculative opening of a solve (spec_start / 8 of the previous timestep's count as plain sweeps before the first
    # evaluation; the rayleigh float32 default is 7): none, the default's neighbour, and 16 -- twice the previous count, which
    # overshoots in every timestep and runs the repeat path -- under the extrapolating plans: never another result
    for spec in (0, 6, 16):
        for plan in (2, 3):
            got = run(plan, spec=spec)
            assert torch.equal(got[0], literal[0]) and torch.equal(got[1], literal[1]), (plan, spec)
            assert int(got[2][:, 2].sum()) == 0, (plan, spec)
    plan2, plan3 = run(2, over=12, spec=0), run(3, over=12, spec=0)
    assert int(plan2[2][:, 2].sum()) > 0 and int((plan2[0] > literal[0]).sum()) > 0     # the hook does provoke late stops
    assert int(plan3[2][:, 2].sum()) > 0 and int(plan3[2][:, 3].sum()) == int(plan3[2][:, 2].sum())
    assert torch.equal(plan3[0], literal[0]) and torch.equal(plan3[1], literal[1])


def test_tall_grid_overflow_status_and_replica_mask():
    """ns2d_fast4_step keeps the other kernels' contract around the solver: a Poisson solve that exceeds itmax sets the
    status bit that check_status() raises on (the reference prints and exit(1)s: rayleigh.py:451-454), also when a late
    chunk of a scheduled step meets it; a masked step leaves the skipped replicas' fields, observations and episode
    counters untouched."""
    for B, sched in ((2, 0), (288, 2)):
        env = V.VecRayleigh.__new__(V.VecRayleigh)
        V.VecRayleigh.__init__(env, B, DEV, "f32", None, L=1.0, H=3.0)
        env.itmax = 3
        env.set_ndt_act(40)
        env.set_sched(sched)
        env.reset()
        x, y = (np.arange(env.nx + 2) - 0.5) / env.nx, (np.arange(env.ny + 2) - 0.5) / env.ny
        st0 = np.zeros((4, env.nx + 2, env.ny + 2))
        st0[3] = (0.5 - y)[None, :] + 0.08 * np.sin(2 * np.pi * x)[:, None] * np.sin(np.pi * y)[None, :]
        env.set_state(np.tile(ref_to_dev(st0)[None], (B, 1, 1, 1)))
        env.step(np.tile(np.full((1, 10), 0.7) * np.array([1, -1] * 5), (B, 1)))
        assert env.kernel_name == ("ns2d_fast4_sched" if sched else "ns2d_fast4_step")
        assert int((env.status.cpu().numpy() & 1).sum()) == B
        with pytest.raises(RuntimeError, match="max number of iterations"):
            env.check_status()
        env.close()
    env = V.VecMixing(4, DEV, "f32", L=1.0, H=2.0)
    env.set_ndt_act(10)
    env.reset()
    env.step(np.arange(4))
    before, obs_b, stp_b = env.get_state().clone(), env.obs.clone(), env.get_stp().copy()
    mask = np.array([1, 0, 1, 0], dtype=np.uint8)
    env.step(np.arange(4)[::-1].copy(), mask=mask)
    env.check_status()
    assert env.kernel_name == "ns2d_fast4_step"
    after, stp_a = env.get_state(), env.get_stp()
    for b in range(4):
        same = torch.equal(after[b], before[b])
        assert same == (mask[b] == 0), b
        assert stp_a[b] == stp_b[b] + int(mask[b])
    assert torch.equal(env.obs[1], obs_b[1]) and torch.equal(env.obs[3], obs_b[3])
    env.close()


@pytest.mark.parametrize("kind", ["rayleigh", "mixing"])
def test_tall_grid_ticket_scheduler_matches_the_plain_launch(kind):
    """More replicas than CUs: ns2d_fast4_sched (persistent workgroups drawing 20-timestep chunks of any replica,
    ns2d_sched.h) must return what one workgroup per replica returns, bit for bit: two action steps at B = 320 with distinct
    actions (rayleigh 50x150 float32; mixing 100x130)."""
    B = 320
    out = {}
    for sched in (0, 2):
        if kind == "rayleigh":
            env = V.VecRayleigh(B, DEV, "f32", None, L=1.0, H=3.0)
            x, y = (np.arange(env.nx + 2) - 0.5) / env.nx, (np.arange(env.ny + 2) - 0.5) / env.ny
            st0 = np.zeros((4, env.nx + 2, env.ny + 2))
            st0[3] = (0.5 - y)[None, :] + 0.08 * np.sin(2 * np.pi * x)[:, None] * np.sin(np.pi * y)[None, :]
            env.set_ndt_act(60)
            env.reset()
            env.set_state(np.tile(ref_to_dev(st0)[None], (B, 1, 1, 1)))
            acts = np.random.default_rng(9).uniform(-1, 1, (2, B, 10))
        else:
            env = V.VecMixing(B, DEV, "f32", L=1.0, H=1.3)
            env.set_ndt_act(60)
            env.reset()
            acts = np.random.default_rng(9).integers(0, 4, (2, B))
        assert env.set_variant(1) == 1
        env.set_sched(sched)
        res = []
        for a in acts:
            obs, rwd, _, _, _ = env.step(a)
            env.check_status()
            res.append((env.sweeps.clone(), obs.clone(), rwd.clone()))
        assert env.kernel_name == ("ns2d_fast4_sched" if sched == 2 else "ns2d_fast4_step")
        out[sched] = (res, env.get_state().clone())
        env.close()
    for (s0, o0, r0), (s2, o2, r2) in zip(out[0][0], out[2][0]):
        assert torch.equal(s0, s2) and torch.equal(o0, o2) and torch.equal(r0, r2)
    assert torch.equal(out[0][1], out[2][1])
    assert int(out[0][0][1][0].max()) > int(out[0][0][1][0].min())      # the replicas do differ in their sweep counts


@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_speculative_first_evaluation_never_changes_a_result(dtype):
    """ns2d_fast_impl.h starts a Jacobi solve with spec_start/8 of the previous timestep's sweep count as double sweeps
    before the first evaluation of the residual (the norm never increases, so a first evaluation that does not pass
    proves that no earlier sweep did) and repeats the solve in the ordinary way when that first evaluation passes.
    spec_start = 16 (twice the previous count) overshoots in every timestep and so runs the repeat path; 17 is the default:
    15/16 of the previous count minus the zone in which the residual is already below the landing guard; all settings must
    give the sweep counts and fields of spec_start = 0 bit for bit (full action step of the bench workload, 200 timesteps,
    ticket scheduler and plain launch).  The float64 kernels are built without the jump: there the option must be inert."""
    ref = {}
    for spec, sched in ((0, 0), (4, 0), (7, 0), (16, 0), (17, 0), (0, 2), (6, 2), (16, 2), (17, 2)):
        env, init, acts = _bench_workload(300, 1, dtype)
        env.set_option("spec_start", spec)
        env.set_sched(sched)
        env.step(acts[0])
        env.check_status()
        got = (env.sweeps.clone(), env.get_state().clone(), env.obs.clone())
        env.close()
        if spec == 0:                    # (the p ghosts of the two launch modes differ in the last bit: chunked increments)
            ref[sched] = got
            assert int(got[0].min()) >= 1 and int(got[0].max()) > 100
        else:
            for a, b in zip(ref[sched], got):
                assert torch.equal(a, b), (spec, sched)
    assert torch.equal(ref[0][0], ref[2][0]) and torch.equal(ref[0][2], ref[2][2])


@pytest.mark.parametrize("L,H,dtype,tol", [(1.5, 1.0, "f32", 5e-5), (1.5, 1.0, "f64", F64_TOL), (1.06, 1.0, "f32", 5e-5),
                                            (2.2, 1.28, "f32", 5e-5), (2.2, 1.28, "f64", F64_TOL), (1.0, 1.4, "f32", 5e-5),
                                            (1.2, 2.4, "f32", 5e-5), (2.2, 1.3, "f32", 5e-5), (1.0, 1.4, "f64", F64_TOL),
                                            (1.0, 3.0, "f32", 5e-5), (1.0, 3.0, "f64", F64_TOL), (1.28, 4.0, "f32", 5e-5),
                                            (1.0, 2.9, "f32", 5e-5), (1.0, 2.98, "f64", F64_TOL), (6.0, 1.0, "f32", 5e-5),
                                            (1.06, 3.0, "f32", 5e-5), (1.0, 1.5, "f64", F64_TOL), (1.15, 2.14, "f64", F64_TOL)])
def test_jit_grids_rayleigh_vs_oracle(L, H, dtype, tol):
    """The reference takes any L, H (rayleigh.py:20-27).  Grids without a built-in register-resident kernel get one
    instantiated for them (beacon_amd/jit.py): 75x50 (strips of 10 columns, the last wave 5), 53x50, 110x64, and 50x70,
    60x120, 110x65 (two rows per lane; odd ny: the last lane holds one row; float64: fields in a global scratch), and
    50x150, 64x200, 50x145, 50x149 (ny > 128: ns2d_fast4_impl.h, the Poisson solve in registers with 3 / 4 rows per lane
    -- the last lane holding 1 or 2 rows where 3 does not divide ny --, the transport as a register walk along anti-diagonals),
    and 300x50 (too wide for the one-row-per-lane kernel's registers: the same hybrid with one row per lane, 16 strips of 19 columns),
    and 53x150 (14 strips of 4 columns, the last one a single column: the sweep's barrier placement for that case; 50x149 in
    float64 is the same case under the other shape of the Jacobi loop) -- 30 timesteps with distinct actions against the float64 oracle, and against
    the generic kernel on the same inputs."""
    B = 6
    env = V.VecRayleigh(B, DEV, dtype, None, L=L, H=H)
    env.set_ndt_act(30)
    assert env.set_variant(1) == 1 and getattr(env, "_plugin", None) is not None, "no kernel plugin for %dx%d" % (env.nx, env.ny)
    rng = np.random.default_rng(3)
    x, y = (np.arange(env.nx + 2) - 0.5) / env.nx, (np.arange(env.ny + 2) - 0.5) / env.ny
    st0 = np.zeros((4, env.nx + 2, env.ny + 2))      # conduction profile + a smooth perturbation (robust sweep counts)
    st0[3] = (0.5 - y)[None, :] + 0.08 * np.sin(2 * np.pi * x * L)[:, None] * np.sin(np.pi * y)[None, :]
    acts = rng.uniform(-1, 1, (B, 10))
    out = {}
    for variant in (1, 0):
        env.set_variant(variant)
        env.reset()
        env.set_state(np.tile(ref_to_dev(st0)[None], (B, 1, 1, 1)))
        obs, rwd, _, _, _ = env.step(acts)
        env.check_status()
        out[variant] = (obs.double().cpu().numpy(), rwd.double().cpu().numpy(), dev2ref(env.get_state()), env.sweeps.cpu().numpy())
        if variant == 1:
            from beacon_amd import jit
            rows = jit.choose(env.nx, env.ny, dtype == "f64", 0)["rows"]
            # (float64 with strips of unequal width -- 110x64: 7 x 14 + 12, 75x50: 7 x 10 + 5 -- runs ONE body since round 6, the last
            # strip's surplus columns dead: ns2d_fast_impl.h DEADC; until then such grids took the hybrid kernel)
            want = 4 if env.ny > 128 or env.nx > 208 else 2 if env.ny > 64 else 1
            assert rows == want
            assert env.kernel_name == {1: "ns2d_fast_step", 2: "ns2d_fast2_step", 4: "ns2d_fast4_step"}[rows]
    for b in range(B):
        o = O.rayleigh(init=False, L=L, H=H)
        o.cfg.ndt_act = 30
        o.reset_fields()
        o.st[:4] = st0
        ob, rw, _, _, _ = o.step(acts[b].tolist())
        t = F32["ray_jit", rows] if dtype == "f32" else f32tol(tol, tol, 50 * tol, tol, tol, max(1e-8, 4 * tol))
        for i, F in enumerate("uvpT"):
            assert maxdiff(out[1][2][b][i], o.st[i]) <= t[F], (b, F)
        n = 3 * env.nx_obs_pts * env.ny_obs_pts
        assert maxdiff(out[1][0][b][-n:], ob[-n:]) <= t["obs"] and maxdiff(out[1][1][b], rw) <= t["rwd"]
        assert np.all(np.abs(out[1][3][b] - o.itp) <= (1 if dtype == "f64" else np.maximum(3, 0.02 * o.itp)))
    for i, F in enumerate("uvpT"):          # the generic kernel on the same inputs: both within the tolerance of the oracle
        assert maxdiff(out[1][2][:, i], out[0][2][:, i]) <= (2 * F32["ray_jit", rows][F] if dtype == "f32" else 50 * tol), F
    env.close()


@pytest.mark.parametrize("L,H", [(1.0, 1.1), (1.0, 1.05), (1.0, 2.0), (1.0, 1.3), (2.0, 1.0), (1.06, 2.0)])
def test_jit_grid_mixing_vs_oracle(L, H):
    """mixing(L=1.0, H=1.1 / 1.05): 100x110 / 100x105 (odd ny), two rows per lane, strips of 13 columns (the last wave 9);
    mixing(L=1.0, H=2.0 / 1.3): 100x200 / 100x130, ns2d_fast4_impl.h (15 strips of 7 columns, 4 / 3 rows per lane, transport in
    two row blocks / one; L=1.06: 106x200, 16 strips of 7 columns, the last one a single column); mixing(L=2.0, H=1.0): 200x100 (wider than the two-rows-per-lane kernel's registers take): the
    same hybrid with 2 rows per lane, 16 strips of 13 columns, 4 columns per lane in the transport walk;
    40 timesteps from rest."""
    env = V.VecMixing(4, DEV, "f32", L=L, H=H)
    env.set_ndt_act(40)
    assert env.set_variant(1) == 1 and getattr(env, "_plugin", None) is not None
    env.reset()
    a = np.arange(4)
    obs, rwd, _, _, _ = env.step(a)
    env.check_status()
    assert env.kernel_name == ("ns2d_fast4_step" if env.ny > 128 or env.nx > 128 else "ns2d_fast2_step")
    st = dev2ref(env.get_state())
    sw = env.sweeps.cpu().numpy()
    for b in range(4):
        o = O.mixing(L=L, H=H)
        o.cfg.ndt_act = 40
        o.reset()
        ob, rw, _, _, _ = o.step(int(a[b]))
        t = F32["mix_jit", "%dx%d" % (env.nx, env.ny) if (env.ny > 128 or env.nx > 128) else 2]
        for i, F in enumerate("uvpC"):
            assert maxdiff(st[b][i], o.st[i]) <= t[F], (b, F)
        assert maxdiff(float(rwd[b]), rw) <= t["rwd"]
        assert np.all(np.abs(sw[b] - o.itp) <= np.maximum(3, 0.02 * o.itp)), b
    env.close()


@pytest.mark.parametrize("L,nx", [(2.0, 100), (3.0, 150), (4.0, 200)])
def test_rayleigh_wide_domains_fast_vs_oracle(L, nx):
    """Register-resident instantiations for the reference's other natural aspect ratios (nx = 50 L, ny = 50):
    float32, smooth synthetic start, 2 x 10 timesteps, against the float64 oracle (F32["ray_wide"]) and with
    identical replicas bit-identical; the sweep counts within max(4, 4 %) as for the 100x100 test."""
    ny = 50
    rng = np.random.default_rng(17)
    x, y = (np.arange(nx + 2) - 0.5) / nx, (np.arange(ny + 2) - 0.5) / ny
    init = np.zeros((4, nx + 2, ny + 2))
    init[3] = (0.5 - y)[None, :] + 0.1 * np.sin(2 * np.pi * L * x)[:, None] * np.sin(np.pi * y)[None, :] \
        + 1e-3 * rng.standard_normal((nx + 2, ny + 2))
    B, NDT = 4, 10
    acts = rng.uniform(-1, 1, (2, B, 10))
    acts[:, 3] = acts[:, 0]
    env = V.VecRayleigh(B, DEV, "f32", init, L=L, H=1.0)
    env.set_ndt_act(NDT)
    _variant(env, 1)
    env.reset()
    oracles = [O.rayleigh(init_fields=init, L=L, H=1.0) for _ in range(3)]
    for o in oracles:
        o.cfg.ndt_act = NDT
        o.reset()
    for k in range(2):
        obs, rwd, _, _, _ = env.step(acts[k])
        env.check_status()
        assert env.kernel_name == "ns2d_fast_step"
        raw = env.get_state()
        st, sw = dev2ref(raw), env.sweeps.cpu().numpy()
        assert bool((raw[3] == raw[0]).all()) and np.array_equal(sw[3], sw[0])
        for b, o in enumerate(oracles):
            ob, rw, _, _, _ = o.step(acts[k, b].tolist())
            for i, F in enumerate("uvpT"):
                assert maxdiff(st[b][i], o.st[i]) <= F32["ray_wide"][F], (k, b, F)
            assert maxdiff(obs[b].cpu().numpy(), ob) <= F32["ray_wide"]["obs"] and maxdiff(float(rwd[b]), rw) <= F32["ray_wide"]["rwd"]
            assert np.all(np.abs(sw[b] - o.itp) <= np.maximum(4, 0.04 * o.itp)), (sw[b], o.itp)
    env.close()


@pytest.mark.parametrize("L,H,n_sgts,ra", [(1.5, 1.0, 2, 2.0e4), (1.0, 1.5, 5, 5.0e4), (2.2, 1.3, 7, 1.0e4)])
def test_rayleigh_odd_configs_generic_vs_oracle_f64(L, H, n_sgts, ra):
    """Constructor-space coverage of the generic kernel: grids that are not multiples of anything (75x50, 50x75,
    110x65), segment counts that do not divide nx (the reference leaves the bottom ghost cells right of the last
    segment unwritten, rayleigh.py:199-202), other Rayleigh numbers; float64 against the oracle, 2 x 4 timesteps."""
    nx, ny = int(50 * L), int(50 * H)
    rng = np.random.default_rng(int(1000 * L + 10 * H + n_sgts))
    x, y = (np.arange(nx + 2) - 0.5) / nx, (np.arange(ny + 2) - 0.5) / ny
    init = np.zeros((4, nx + 2, ny + 2))
    init[3] = (0.5 - y)[None, :] + 0.08 * np.sin(2 * np.pi * x * L)[:, None] * np.sin(np.pi * y)[None, :]
    B, NDT = 3, 4
    acts = rng.uniform(-1, 1, (2, B, n_sgts))
    env = V.VecRayleigh(B, DEV, "f64", init, L=L, H=H, n_sgts=n_sgts, ra=ra)
    env.set_ndt_act(NDT)
    env.set_variant(0)          # this test is about the generic kernel (some of these grids would get a kernel plugin)
    env.reset()
    oracles = [O.rayleigh(init_fields=init, L=L, H=H, n_sgts=n_sgts, ra=ra) for _ in range(B)]
    for o in oracles:
        o.cfg.ndt_act = NDT
        o.reset()
    for k in range(2):
        obs, rwd, _, _, _ = env.step(acts[k])
        env.check_status()
        assert env.kernel_name == "ns2d_generic_step"
        st, sw = dev2ref(env.get_state()), env.sweeps.cpu().numpy()
        for b, o in enumerate(oracles):
            ob, rw, _, _, _ = o.step(acts[k, b].tolist())
            for i, F in enumerate("uvpT"):
                assert maxdiff(st[b][i], o.st[i]) <= F64_TOL * (50 if F == "p" else 1), (k, b, F)
            assert maxdiff(obs[b].cpu().numpy(), ob) <= F64_TOL and maxdiff(float(rwd[b]), rw) <= 1e-8
            assert np.max(np.abs(sw[b] - o.itp)) <= 1
    env.close()


def test_rayleigh_episode_end_and_overflow():
    g = golden("rayleigh_default")
    env = V.VecRayleigh(2, DEV, "f64", _ray_init(g))
    env.set_ndt_act(2)
    env.reset()
    env.set_stp([98, 99])
    _, _, done, trunc, _ = env.step(np.zeros((2, 10)))
    assert done.cpu().tolist() == [0, 1] and trunc.cpu().tolist() == [0, 1]     # stp == n_act-1 (rayleigh.py:150)
    assert env.get_stp().tolist() == [99, 100]
    env.close()
    # Poisson non-convergence: the reference prints and exit(1)s; here a status bit + RuntimeError
    env = V.VecRayleigh.__new__(V.VecRayleigh)
    V.VecRayleigh.__init__(env, 1, DEV, "f64", _ray_init(g))
    env.itmax = 3
    env.set_ndt_act(2)
    env.reset()
    env.step(np.full((1, 10), 0.7) * np.array([1, -1] * 5))
    with pytest.raises(RuntimeError, match="max number of iterations"):
        env.check_status()
    env.close()


def test_rayleigh_single_env_mirror():
    """Reference-style usage: class rayleigh, list actions normalised in place, float64 obs."""
    g = golden("rayleigh_default")
    env = E.rayleigh()
    obs, info = env.reset()
    assert obs.dtype == np.float64 and obs.shape == (192,) and info is None
    assert maxdiff(obs, g["reset_obs"]) == 0
    a = g["actions"][0].tolist()
    obs, rwd, done, trunc, info = env.step(a)
    assert isinstance(rwd, float) and isinstance(done, bool) and info is None
    assert maxdiff(a, g["step0_a_mutated"]) <= 1e-15
    assert maxdiff(obs, g["step0_obs"]) <= F64_TOL
    assert maxdiff(env.T, g["step0_T"]) <= F64_TOL
    env.close()


# ---------------------------------------------------------------------------------------------
# mixing
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype,swrel", [("f64", 0.0), ("f32", 0.02)])
@pytest.mark.parametrize("act", [0, 1, 2, 3])
@pytest.mark.parametrize("variant", [0, 1])
def test_mixing_from_rest_vs_golden(act, dtype, swrel, variant):
    """100x100 from rest, 3 timesteps; the first Poisson solve takes 2466 sweeps; generic kernel and two-rows-per-lane
    kernel (float32: fields in LDS; float64: fields in a global scratch, rhs in LDS).
    f32: sweep counts within 2 %, fields F32["mix_rest"] (measured errors x <= 10)."""
    tol = F32["mix_rest"] if dtype == "f32" else f32tol(F64_TOL, F64_TOL, 50 * F64_TOL, F64_TOL, F64_TOL, F64_TOL)
    g = golden("mixing_a%d" % act)
    env = V.VecMixing(2, DEV, dtype)
    env.set_ndt_act(3)
    assert env.set_variant(variant) == variant
    obs, _ = env.reset()
    assert maxdiff(obs.cpu().numpy()[0], g["reset_obs"]) == 0
    st = dev2ref(env.get_state())[0]
    assert maxdiff(st[3], g["reset_C"]) == 0
    obs, rwd, done, trunc, _ = env.step(np.array([act, act]))
    env.check_status()
    sw = env.sweeps.cpu().numpy()[0]
    assert np.all(np.abs(sw - g["itp"][0]) <= np.maximum(1 if dtype == "f64" else 3, swrel * g["itp"][0])), sw
    st = dev2ref(env.get_state())[0]
    for i, F in enumerate("uvpC"):
        assert maxdiff(st[i], g["step0_" + F]) <= tol[F], F
    assert maxdiff(obs.cpu().numpy()[0], g["step0_obs"]) <= tol["obs"]
    assert maxdiff(float(rwd[0]), float(g["step0_rwd"])) <= tol["rwd"]
    assert env.kernel_name == ("ns2d_fast2_step" if variant else "ns2d_generic_step")
    env.close()


def test_mixing_synth_all_actions_f64():
    g = golden("mixing_synth")
    env = V.VecMixing(5, DEV, "f64")
    env.set_ndt_act(4)
    env.reset()
    st0 = np.stack([ref_to_dev(g[k]) for k in ("u0", "v0", "p0", "C0")])
    env.set_state(np.tile(st0[None], (5, 1, 1, 1)))
    obs, rwd, _, _, _ = env.step(np.arange(5))           # action 4: all walls at rest
    env.check_status()
    st = dev2ref(env.get_state())
    sw = env.sweeps.cpu().numpy()
    for a in range(5):
        assert np.max(np.abs(sw[a] - g["a%d_itp" % a])) <= 1
        for i, F in enumerate("uvpC"):
            assert maxdiff(st[a][i], g["a%d_%s" % (a, F)]) <= F64_TOL * (50 if F == "p" else 1)
        assert maxdiff(obs[a].cpu().numpy(), g["a%d_obs" % a]) <= F64_TOL
    env.close()


def test_mixing_fast2_vs_generic_and_identical_replicas():
    """mixing 100x100 f32 on ns2d_fast2 from the synthetic developed state: 16 replicas per action
    (bit-identical within a group, sweep counts included), then the generic kernel on the same
    inputs (5e-5 on u, v, C; p 50x looser; sweeps within max(3, 1 %))."""
    g = golden("mixing_synth")
    B, NDT = 64, 6
    env = V.VecMixing(B, DEV, "f32")
    env.set_ndt_act(NDT)
    env.reset()
    st0 = np.stack([ref_to_dev(g[k]) for k in ("u0", "v0", "p0", "C0")])
    acts = np.repeat(np.arange(4), 16)
    runs = {}
    for variant in (1, 0):
        assert env.set_variant(variant) == variant
        env.set_state(np.tile(st0[None], (B, 1, 1, 1)))
        env.set_stp(0)
        out = []
        for k in range(2):
            env.step(acts)
            env.check_status()
            out.append((env.get_state().clone(), env.sweeps.cpu().numpy().copy()))
        assert env.kernel_name == ("ns2d_fast2_step" if variant else "ns2d_generic_step")
        runs[variant] = out
    for k in range(2):
        st, sw = runs[1][k]
        for a in range(4):
            grp = slice(16 * a, 16 * a + 16)
            assert bool((st[grp] == st[16 * a:16 * a + 1]).all()) and bool((sw[grp] == sw[16 * a]).all())
        gs, gw = runs[0][k]
        for i, F in enumerate("uvpC"):
            assert float((st[:, i] - gs[:, i]).abs().max()) <= 5e-5 * (50 if F == "p" else 1), (k, F)
        assert np.all(np.abs(sw - gw) <= np.maximum(3, 0.01 * gw))
    env.close()


def test_mixing_fast2_ticket_scheduler_matches_single_launch():
    """ns2d_fast2_sched (persistent workgroups drawing 10-timestep chunks, here 5 workgroups for 24
    replicas) against the plain one-workgroup-per-replica launch: obs, rewards, sweep counts and
    interior fields bit for bit (p ghost cells: one float32 rounding, see the rayleigh test above)."""
    import subprocess
    import sys
    code = (
        "import sys, numpy as np, torch; sys.path.insert(0, %r)\n"
        "from beacon_amd import vec as V\n"
        "env = V.VecMixing(24, 'cuda:0', 'f32')\n"
        "env.set_ndt_act(60)\n"
        "env.set_sched(*[int(x) for x in sys.argv[3].split(',')])\n"
        "env.reset()\n"
        "a = np.random.default_rng(5).integers(0, 4, (2, 24))\n"
        "for k in range(2): obs, rwd, *_ = env.step(a[k])\n"
        "env.check_status()\n"
        "assert env.kernel_name == sys.argv[2], env.kernel_name\n"
        "np.save(sys.argv[1], np.concatenate([obs.cpu().numpy().ravel(), rwd.cpu().numpy(),"
        " env.get_state().cpu().numpy().ravel(), env.sweeps.cpu().numpy().ravel().astype(np.float32)]))\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for tag, kname, sched in (("single", "ns2d_fast2_step", "0,0,0,0"), ("ticket", "ns2d_fast2_sched", "2,5,0,0")):
        path = "/tmp/bcn_mix_%s.npy" % tag
        r = subprocess.run([sys.executable, "-c", code, path, kname, sched], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(path))
    B, ndt = 24, 60
    n_head = B * 192 + B
    assert np.array_equal(outs[0][:n_head], outs[1][:n_head])                         # obs, rwd
    assert np.array_equal(outs[0][-B * ndt:], outs[1][-B * ndt:])                     # sweeps
    st = [o[n_head:-B * ndt].reshape(B, 4, 102, 102) for o in outs]
    for f in (0, 1, 3):
        assert np.array_equal(st[0][:, f], st[1][:, f])
    assert np.array_equal(st[0][:, 2, 1:-1, 1:-1], st[1][:, 2, 1:-1, 1:-1])
    assert np.max(np.abs(st[0][:, 2] - st[1][:, 2])) < 1e-5


def test_mixing_full_steps_f32_vs_f64():
    """The float32 tolerance of mixing over full action steps (2 x 250 timesteps from rest, random actions):
    ns2d_fast2 (float32) against the float64 generic kernel.  Measured (scripts/mix_drift.py): observations
    3e-5 / 8e-5 after one / two steps, u, v, C <= 3e-4, p <= 8e-4, rewards 1e-6; the sweep count differs
    (by <= 9) in ~40 % of the timesteps because tol = 1e-4 is reached after a handful of sweeps.  Asserted
    with a margin of ~5x."""
    B = 8
    acts = np.random.default_rng(5).integers(0, 4, (2, B))
    envs = {dt: V.VecMixing(B, DEV, dt) for dt in ("f32", "f64")}
    for e in envs.values():
        e.reset()
    for k in range(2):
        out = {}
        for dt, e in envs.items():
            obs, rwd, _, _, _ = e.step(acts[k])
            e.check_status()
            out[dt] = (obs.double().cpu(), rwd.double().cpu(), e.get_state().double().cpu(), e.sweeps.cpu().numpy())
        assert float((out["f32"][0] - out["f64"][0]).abs().max()) < 4e-4
        assert float((out["f32"][1] - out["f64"][1]).abs().max()) < 1e-5
        for i, F in enumerate("uvpC"):
            assert float((out["f32"][2][:, i] - out["f64"][2][:, i]).abs().max()) < (4e-3 if F == "p" else 1.5e-3), (k, F)
        assert int(np.abs(out["f32"][3] - out["f64"][3]).max()) <= 30
    assert envs["f32"].kernel_name.startswith("ns2d_fast2") and envs["f64"].kernel_name.startswith("ns2d_fast2")
    for e in envs.values():
        e.close()


# ---------------------------------------------------------------------------------------------
# burgers
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype,tol", [("f64", 1e-12), ("f32", 1e-4)])
def test_burgers_vs_golden(dtype, tol):
    """Two seeded episodes run as two replicas of one batch (200 and 25 steps).
    f32 tolerance: 1e-4 absolute on obs/reward as a hard per-step bound, and the measured error profile asserted
    explicitly so that a change of the float32 arithmetic cannot hide under it: mean over the 200 steps <= 1e-5
    (measured 4e-6) and at most 3 steps above 5e-5 (measured: one peak of 4-6e-5 at step 78, with the two-reciprocal
    and the one-reciprocal form of the limiter alike); 1e-3 on the full field after 12400 timesteps (the van Leer
    ratio amplifies rounding at the downstream shocks)."""
    ftol = tol if dtype == "f64" else 1e-3        # full field after 12 400 timesteps: measured 1.8e-4
    ftol1 = tol if dtype == "f64" else 3e-5       # ... after 1 550 timesteps (second replica): measured 3.3e-6
    rtol = tol if dtype == "f64" else 2e-5        # rewards: measured 2.0e-6
    errs = []
    g = golden("burgers")
    env = V.VecBurgers(2, DEV, dtype)
    obs, _ = env.reset()
    assert maxdiff(obs.cpu().numpy()[0], g["s0_reset_obs"]) == 0
    n0, n1 = len(g["s0_actions"]), len(g["s1_actions"])
    for k in range(n0):
        a = np.array([g["s0_actions"][k, 0], g["s1_actions"][min(k, n1 - 1), 0]])
        nz = np.array([g["s0_noise"][k], g["s1_noise"][min(k, n1 - 1)]])
        obs, rwd, done, trunc, _ = env.step(a, nz)
        errs.append(maxdiff(obs[0].cpu().numpy(), g["s0_obs"][k]))
        assert errs[-1] <= tol
        assert maxdiff(float(rwd[0]), g["s0_rwd"][k]) <= rtol
        if k < n1:
            assert maxdiff(obs[1].cpu().numpy(), g["s1_obs"][k]) <= tol
        if k == n1 - 1:
            st = env.get_state().cpu().numpy()[1]
            for i, f in enumerate(("u", "up", "upp")):
                assert maxdiff(st[i], g["s1_" + f]) <= ftol1
    assert bool(done[0]) and bool(trunc[0])              # 200th step ends the episode
    if dtype == "f32":
        assert np.mean(errs) <= 1e-5 and int((np.array(errs) > 5e-5).sum()) <= 3, (np.mean(errs), np.max(errs))
    st = env.get_state().cpu().numpy()[0]
    for i, f in enumerate(("u", "up", "upp")):
        assert maxdiff(st[i], g["s0_" + f]) <= ftol
    env.close()


@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_inlet_noise_drawn_inside_the_step_kernel(dtype):
    """step() without a noise tensor: the kernel draws uniform(-sigma, sigma) itself (bcn_set_noise: Philox4x32-10 keyed by
    the seed, the GLOBAL replica index, the replica's step count and the timestep).  burgers keeps its inlet value
    u[0] = u_target + noise in the state (burgers.py:137), shkadov h[0] = 1 + noise of the last timestep (shkadov.py:204):
    the law (range, mean, variance), independence across replicas and steps, reproducibility from the seed, the
    replica offset of a sharded batch, and fresh values when a captured graph replays."""
    B = 1024
    def inlet(env):
        return (env.get_state()[:, 0, 0].double() - env.u_target).cpu().numpy()
    env = V.VecBurgers(B, DEV, dtype, seed=5)
    env.reset()
    zero = torch.zeros(B, dtype=env.tdtype, device=DEV)
    env.step(zero)
    n1 = inlet(env)
    env.step(zero)
    n2 = inlet(env)
    s = env.sigma
    for n in (n1, n2):
        assert np.abs(n).max() <= s * (1 + 1e-6) and abs(n.mean()) < 4 * s / np.sqrt(3 * B)
        assert 0.9 * s / np.sqrt(3) < n.std() < 1.1 * s / np.sqrt(3) and len(np.unique(n)) > 0.99 * B
    assert abs(np.corrcoef(n1, n2)[0, 1]) < 0.15 and abs(np.corrcoef(n1[:-1], n1[1:])[0, 1]) < 0.15
    env.close()
    same = V.VecBurgers(B, DEV, dtype, seed=5)
    same.reset(); same.step(zero)
    assert np.array_equal(inlet(same), n1)                       # the seed reproduces the stream ...
    same.set_noise_seed(6); same.reset(); same.step(zero)
    assert not np.array_equal(inlet(same), n1)                   # ... another seed gives another
    same.close()
    half = V.VecBurgers(B // 2, DEV, dtype, seed=5)              # the upper shard of a batch split over two ranks
    half.set_noise_seed(5, B // 2)
    half.reset(); half.step(zero[:B // 2])
    assert np.array_equal(inlet(half), n1[B // 2:])
    half.close()
    env = V.VecBurgers(64, DEV, dtype, seed=1)
    env.reset()
    g = env.capture(torch.zeros((2, 64), dtype=env.tdtype, device=DEV), None, n_steps=2)
    g.replay(); a = inlet(env)
    g.replay(); b = inlet(env)
    assert not np.array_equal(a, b)                              # the draw counter lives on the device: a replay draws anew
    env.close()
    env = V.VecShkadov(256, DEV, dtype, None, n_jets=5, seed=2)
    env.reset()
    env.step(torch.zeros((256, 5), dtype=env.tdtype, device=DEV))
    h0 = (env.get_state()[:, 0, 0].double() - 1.0).cpu().numpy()
    assert np.abs(h0).max() <= env.sigma * (1 + 1e-6) and h0.std() > 0.4 * env.sigma and len(np.unique(h0)) > 250
    env.close()


def test_burgers_nx512_vs_oracle_and_mirror():
    """BASELINE grid N=512 (the reference hard-codes 500) against the oracle; plus the
    reference-style class drawing its noise from numpy's global stream."""
    B = 4
    rng = np.random.default_rng(3)
    env = V.VecBurgers(B, DEV, "f64", nx=512)
    env.reset()
    ors = [O.burgers(nx=512) for _ in range(B)]
    for o in ors:
        o.reset()
    for k in range(10):
        a, nz = rng.uniform(-1, 1, B), rng.uniform(-0.1, 0.1, B)
        obs, rwd, _, _, _ = env.step(a, nz)
        for b, o in enumerate(ors):
            ob, rw, _, _, _ = o.step([a[b]], nz[b])
            assert maxdiff(obs[b].cpu().numpy(), ob) <= 1e-12 and maxdiff(float(rwd[b]), rw) <= 1e-12
        st = env.get_state().cpu().numpy()
        for b, o in enumerate(ors):       # N = 512 fills the wave exactly: the kernel without per-cell masks (FIT), bit for bit
            assert np.array_equal(st[b, 0], o.u) and np.array_equal(st[b, 1], o.up) and np.array_equal(st[b, 2], o.upp), (k, b)
    assert env.kernel_name == "burgers_step_k"
    env.close()
    # float32, N = 512 / 256: the packed kernel (burgers_step_pk_k: two cells per v_pk instruction, observations and reward from
    # the registers) against the one-wave kernel it replaces (option one_wave = 2) and the float64 oracle over 30 action steps.
    # Measured (scripts/burgers_pk_diff.py): the two kernels differ in the last bit of some fused multiply-adds, 2e-7 after 5
    # steps, 3.6e-6 after 25-40; both sit at the SAME distance from the oracle (5.6e-7 after one step, 3.8e-5 after 30)
    # (500: the reference's own grid; 200, 130: four cells per lane; 497: the last cell is a lane's FIRST cell, its copy comes from the
    # lane below -- the packed kernel with per-lane masks)
    # (coarse grids under random forcing blow up after 15-30 action steps, in the oracle too: eight steps there)
    for nx in (512, 256, 500, 200, 497, 130):
        B = 6
        envs = []
        for ow in (1, 2):
            e = V.VecBurgers(B, DEV, "f32", nx=nx)
            e.set_option("one_wave", ow)
            e.reset()
            envs.append(e)
        ors = [O.burgers(nx=nx) for _ in range(B)]
        for o in ors:
            o.reset()
        for k in range(30 if nx >= 256 else 8):
            a, nz = rng.uniform(-1, 1, B), rng.uniform(-0.1, 0.1, B)
            outs = [e.step(a, nz) for e in envs]
            st = [e.get_state() for e in envs]
            assert maxdiff(st[0].cpu().numpy(), st[1].cpu().numpy()) <= 1e-5 and maxdiff(outs[0][0].cpu().numpy(), outs[1][0].cpu().numpy()) <= 1e-5, (nx, k)
            assert maxdiff(outs[0][1].cpu().numpy(), outs[1][1].cpu().numpy()) <= 2e-6
            assert torch.equal(outs[0][2], outs[1][2]) and torch.equal(outs[0][3], outs[1][3])
            for b, o in enumerate(ors):
                ob, rw, _, _, _ = o.step([a[b]], nz[b])
                assert maxdiff(outs[0][0][b].cpu().numpy(), ob) <= 2e-4 and abs(float(outs[0][1][b]) - rw) <= 2e-4, (nx, k, b)
        for e in envs:
            e.close()
    g = golden("burgers")
    e = E.burgers()
    e.reset()
    np.random.seed(1)
    for k in range(5):
        obs, rwd, done, trunc, info = e.step(g["s1_actions"][k].tolist())
        assert maxdiff(obs, g["s1_obs"][k]) <= 1e-12
    e.close()


# ---------------------------------------------------------------------------------------------
# shkadov
# ---------------------------------------------------------------------------------------------
SHK_F32_TOL0, SHK_F32_GROWTH, SHK_F32_CAP = 1.1e-5, 1.4, 0.1


def shkadov_tol(dtype, k, reward=False):
    """Tolerance after k action steps.  float64: the kernel keeps the reference's operation order without FMA contraction,
    so observations and fields are bit-identical to the reference at every step (measured 0.0 over the 30-step fixtures);
    1e-12 is left for the rewards (reductions).  float32 (the kernel works on h - 1, q - 1: env1d_impl.inc): the wavy film
    amplifies rounding differences -- measured on the fixtures 1-2e-6 after one step, 8e-6 / 1.3e-5 after six (default grids /
    N = 4096), 6e-5 after ten, 1.5e-3 after twenty, 1.2e-2 after thirty -- so the bound follows that growth, 1.1e-5 x 1.4^k
    (5-10 x the measured error at every horizon), and is CAPPED at 0.1 (|h - 1| of the developed film is ~1): from there on
    a trajectory comparison says nothing, and test_shkadov_f32_episode_statistics_match_f64 takes over.  Rewards are means of
    (h - 1)^2 over 500 cells: measured 4e-10 (first step) to 3e-7 (thirtieth): 4e-9 x 1.2^k."""
    if dtype == "f64":
        return 1e-12
    if reward:
        return 4e-9 * 1.2 ** k
    return min(SHK_F32_TOL0 * SHK_F32_GROWTH ** k, SHK_F32_CAP)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("tag,kw,init", [("j5", dict(n_jets=5), True), ("j10", dict(n_jets=10), True),
                                         ("n4096", dict(L0=699.2, n_jets=10), False)])
def test_shkadov_vs_golden(tag, kw, init, dtype):
    """Reference episodes (30 steps at the default grids, 6 at N = 4096) with the reference's own noise stream: see
    shkadov_tol for the bounds."""
    g = golden("shkadov")
    init_fields = np.stack([g[tag + "_h_init"], g[tag + "_q_init"]]) if init else None
    env = V.VecShkadov(2, DEV, dtype, init_fields, **kw)
    obs, _ = env.reset()
    assert maxdiff(obs.cpu().numpy()[0], g[tag + "_reset_obs"]) <= (0 if dtype == "f64" else 1e-6)
    n = len(g[tag + "_actions"])
    for k in range(n):
        tol = shkadov_tol(dtype, k + 1)
        a = np.tile(g[tag + "_actions"][k], (2, 1))
        nz = np.tile(g[tag + "_noise"][k], (2, 1))
        obs, rwd, done, trunc, _ = env.step(a, nz)
        assert maxdiff(obs[0].cpu().numpy(), g[tag + "_obs"][k]) <= tol, k
        assert maxdiff(float(rwd[0]), g[tag + "_rwd"][k]) <= shkadov_tol(dtype, k + 1, reward=True)
        assert not bool(done[0])
    st = env.get_state().cpu().numpy()[1]
    tol = shkadov_tol(dtype, n)
    assert maxdiff(st[0], g[tag + "_h"]) <= tol and maxdiff(st[1], g[tag + "_q"]) <= tol
    env.close()


@pytest.mark.parametrize("L0,n_jets", [(10.0, 1), (30.0, 2), (150.0, 5)])
def test_shkadov_packed_step_variants_match_the_scalar_step_and_the_oracle(L0, n_jets):
    """The packed float32 timestep of shkadov_step_k has four wave variants (interior; holding the first thread; the last
    thread; both, where the array is ONE wave) besides the jet flag.  nx = 200: one wave with both ends; nx = 350: not a
    multiple of 4, the scalar step runs whatever the option (the control of this test); nx = 1100, the reference's default:
    five waves with cells, three idle.  Against option one_wave = 2 (the scalar step in the same exchange layout) and the
    float64 oracle, from the flat film with inlet noise and random jets, ten action steps."""
    rng = np.random.default_rng(17)
    B = 3
    envs = []
    for ow in (1, 2):
        e = V.VecShkadov(B, DEV, "f32", None, L0=L0, n_jets=n_jets, jet_pos=L0)
        e.set_option("one_wave", ow)
        e.reset()
        envs.append(e)
    o = O.shkadov(L0=L0, n_jets=n_jets, jet_pos=L0, init_fields=np.ones((2, envs[0].nx)))    # the flat film
    o.rand_init = False
    o.reset()
    nx, ndt = envs[0].nx, envs[0].ndt_act
    for k in range(10):
        a, nz = rng.uniform(-1, 1, (B, n_jets)), rng.uniform(-5e-4, 5e-4, (B, ndt))
        outs = [e.step(a, nz) for e in envs]
        st = [e.get_state().double().cpu().numpy() for e in envs]
        assert maxdiff(st[0][:, :2], st[1][:, :2]) <= 5e-6 and maxdiff(st[0][:, 2:], st[1][:, 2:]) <= 5e-4, (nx, k)
        assert maxdiff(outs[0][0].cpu().numpy(), outs[1][0].cpu().numpy()) <= 5e-6
        ob, rw, dn, tr, _ = o.step(a[0].tolist(), nz[0])
        # h within the bound of the reference episodes; q -- the flux, which random full-amplitude jets drive five times harder than
        # those episodes do -- no further from the oracle than the scalar step is (measured: 1.4e-4 after seven steps, both)
        assert maxdiff(st[0][0, 0], o.h) <= shkadov_tol("f32", k + 1), (nx, k)
        assert maxdiff(st[0][0, 1], o.q) <= max(1.5 * maxdiff(st[1][0, 1], o.q), shkadov_tol("f32", k + 1)), (nx, k)
        assert maxdiff(outs[0][0][0].double().cpu().numpy(), ob) <= 5 * shkadov_tol("f32", k + 1)
    for e in envs:
        e.close()


# measured (round 4): see the assertions of the two tests below
# developed N = 4096 film, float32 against the float64 oracle: observations 4.6e-6 at the third step, rewards 1.3e-9; wave
# amplitude of the 1024 films (below); episode statistics of float32 against float64 over 64 replicas: mean of the returns
# 3e-4 sigma apart, their standard deviations 4e-5 apart, per-step batch-mean reward 1e-3 sigma
SHK_DEV = dict(f32_step=(1.5e-5, 1.6), f32_rwd=1e-8, amp_min=0.3, stat_mean=3e-3, stat_std=4e-4, stat_step=1e-2)


def test_shkadov_n4096_b1024_from_a_developed_film():
    """BASELINE configs[2] from a DEVELOPED film (shkadov/init.py:13-27: n_warmup = 4000 uncontrolled action steps under
    inlet noise, then dump): B = 1024 replicas at N = 4096 warmed up on the device with the kernel's own inlet noise
    (every replica its own stream), then stepped with non-zero jets -- the limiter branches and the noise amplification
    that the flat film never exercises.  Replicas 0..3 are checked against the float64 oracle over three action steps from
    the developed state (float32 kernel: measured growth; float64 kernel from the same state: bit-identical fields), the
    whole batch for finiteness, waviness, and the blow-up rule (shkadov.py:176-180) against the returned flags."""
    B, NJ = 1024, 10
    env = V.VecShkadov(B, DEV, "f32", None, L0=699.2, n_jets=NJ, seed=11)
    assert env.nx == 4096
    env.reset()
    zero = torch.zeros((B, NJ), dtype=env.tdtype, device=DEV)
    st0 = env.warmup(env.n_warmup_ref, zero).clone()               # [B, 4, nx]: h, q, rhsh, rhsq
    assert bool(torch.isfinite(st0).all())
    amp = (st0[:, 0] - 1.0).abs().amax(dim=1)
    maxdiff(float(amp.min()), 0.0), maxdiff(float(amp.mean()), 0.0)        # (logged)
    assert float(amp.min()) >= SHK_DEV["amp_min"], float(amp.min())        # every film is wavy (the reference's: 0.99) ...
    assert float((st0[0, 0] - st0[1, 0]).abs().max()) > 1e-3                # ... and its own
    rng = np.random.default_rng(21)
    acts = rng.uniform(-1, 1, (3, B, NJ))
    noise = rng.uniform(-env.sigma, env.sigma, (3, B, env.ndt_act))
    e64 = V.VecShkadov(4, DEV, "f64", None, L0=699.2, n_jets=NJ)
    e64.reset()
    e64.set_state(st0[:4].double())
    oracles = []
    for b in range(4):
        o = O.shkadov(init=False, L0=699.2, n_jets=NJ)
        o.w[:] = st0[b].double().cpu().numpy()
        oracles.append(o)
    for k in range(3):
        obs, rwd, done, trunc, _ = env.step(acts[k], noise[k])
        o64, r64, _, _, _ = e64.step(acts[k, :4], noise[k, :4])
        st64 = e64.get_state().cpu().numpy()
        for b, o in enumerate(oracles):
            ob, rw, dn, _, _ = o.step(acts[k, b].tolist(), noise[k, b])
            assert maxdiff(st64[b, 0], o.h) == 0 and maxdiff(st64[b, 1], o.q) == 0        # float64 kernel: bit-identical
            assert maxdiff(o64[b].cpu().numpy(), ob) == 0 and maxdiff(float(r64[b]), rw) <= 1e-12
            t = SHK_DEV["f32_step"][0] * SHK_DEV["f32_step"][1] ** k
            assert maxdiff(obs[b].double().cpu().numpy(), ob) <= t, (k, b)
            assert maxdiff(float(rwd[b]), rw) <= SHK_DEV["f32_rwd"] and bool(done[b]) == bool(dn)
    # twenty more steps with random jets on the whole batch: the flags follow the rule, nothing silently goes NaN
    for k in range(20):
        obs, rwd, done, trunc, _ = env.step(rng.uniform(-1, 1, (B, NJ)))
        h = env.get_state()[:, 0]
        blow = ((h < -25.0) | (h > 25.0) | ~torch.isfinite(h)).any(dim=1)
        assert torch.equal(blow, (env.status & 2).bool()) and bool((done.bool() >= blow).all())
        assert bool((rwd[blow] == -1.0).all()) and bool(torch.isfinite(obs[~blow]).all())
        if bool(blow.any()):
            env.reset(mask=blow)
    env.close(); e64.close()


def test_shkadov_f32_episode_statistics_match_f64():
    """Where trajectories cannot be compared (the film is chaotic: shkadov_tol), distributions can: 64 replicas of the
    10-jet default grid from the packaged developed film, 100 action steps with per-replica random jets and inlet noise,
    the same inputs in float32 and float64.  Compared: mean and standard deviation over the replicas of the episode return,
    in units of the float64 standard deviation, and the per-step batch-mean reward."""
    B, NJ, N = 64, 10, 100
    init = E.packaged_init("shkadov")
    rng = np.random.default_rng(33)
    acts = rng.uniform(-1, 1, (N, B, NJ))
    ret, mean_r = {}, {}
    for dt in ("f32", "f64"):
        env = V.VecShkadov(B, DEV, dt, init, n_jets=NJ)
        noise = np.random.default_rng(34).uniform(-env.sigma, env.sigma, (N, B, env.ndt_act))
        env.reset()
        tot = torch.zeros((B,), dtype=torch.float64, device=DEV)
        per = []
        for k in range(N):
            obs, rwd, done, trunc, _ = env.step(acts[k], noise[k])
            assert not bool(done.any())
            tot += rwd.double()
            per.append(float(rwd.double().mean()))
        ret[dt], mean_r[dt] = tot.cpu().numpy(), np.array(per)
        env.close()
    s64 = float(ret["f64"].std())
    assert s64 > 0
    assert maxdiff(ret["f32"].mean() / s64, ret["f64"].mean() / s64) <= SHK_DEV["stat_mean"]
    assert maxdiff(ret["f32"].std() / s64, 1.0) <= SHK_DEV["stat_std"]
    assert maxdiff(mean_r["f32"] / s64 * N, mean_r["f64"] / s64 * N) <= SHK_DEV["stat_step"]


def test_shkadov_blowup_and_rand_init_mirror():
    env = V.VecShkadov(2, DEV, "f64", None, n_jets=5)
    env.reset()
    st = env.get_state().cpu().numpy()
    st[1, 0] = 1.0 + 30.0 * np.exp(-((np.arange(env.nx) - 400) / 20.0) ** 2)
    env.set_state(st)
    obs, rwd, done, trunc, _ = env.step(np.zeros((2, 5)), np.zeros((2, 50)))
    assert done.cpu().tolist() == [0, 1] and trunc.cpu().tolist() == [0, 0]
    assert float(rwd[1]) == -1.0 and env.status.cpu().tolist() == [0, 2]      # shkadov.py:176-180
    env.close()
    # reference-style class: reset() runs random.randint(0,400) uncontrolled steps (:119-123),
    # count from python's `random`, inlet noise from numpy's global stream
    import random
    g = golden("shkadov")
    e = E.shkadov(n_jets=5)
    random.seed(3)
    np.random.seed(9)
    obs, _ = e.reset()
    n = int(g["rand_n"])
    assert e.stp == 0
    # exactly n*ndt_act draws were consumed from the numpy stream
    nxt = np.random.uniform(-e.sigma, e.sigma, 1)[0]
    np.random.seed(9)
    stream = np.random.uniform(-e.sigma, e.sigma, n * e.ndt_act + 1)
    assert nxt == stream[-1]
    # the same n uncontrolled steps driven by hand give the same state, bit for bit
    env = V.VecShkadov(1, DEV, "f64", E.packaged_init("shkadov"), n_jets=5)
    env.reset()
    for i in range(n):
        o2, _, _, _, _ = env.step(None, stream[i * 50:(i + 1) * 50].reshape(1, 50))
    assert maxdiff(o2[0].cpu().numpy(), obs) == 0
    assert maxdiff(env.get_state().cpu().numpy()[0, 0], e.h) == 0
    # against the reference itself only loosely: the wavy film is chaotic -- rounding-level
    # differences (FMA contraction) grow ~1.35x per action step (measured), 1e-15 -> 1e-2 over
    # these 121 steps; the 30-step episodes above pin the arithmetic at 1e-10
    assert maxdiff(obs, g["rand_reset_obs"]) <= 0.1
    env.close()
    e.close()


# ---------------------------------------------------------------------------------------------
# sloshing
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype,tol", [("f64", 1e-12), ("f32", 5e-5)])
def test_sloshing_vs_golden(dtype, tol):
    g = golden("sloshing")
    env = V.VecSloshing(2, DEV, dtype, np.stack([g["h_init"], g["q_init"]]))
    obs, _ = env.reset()
    assert maxdiff(obs.cpu().numpy()[0], g["reset_obs"]) <= (0 if dtype == "f64" else 1e-7)
    for k in range(len(g["actions"])):
        obs, rwd, done, trunc, _ = env.step(np.tile(g["actions"][k], 2))
        assert maxdiff(obs[1].cpu().numpy(), g["obs"][k]) <= tol, k                    # f32 measured 7.6e-6
        assert maxdiff(float(rwd[1]), g["rwd"][k]) <= (tol if dtype == "f64" else 2.5e-7)  # f32 measured 2.9e-8
    st = env.get_state().cpu().numpy()[0]
    ftol = (tol, tol, 100 * tol, 100 * tol) if dtype == "f64" else (1.5e-5, 5e-5, 1.5e-3, 5e-3)   # f32 measured 1.7e-6 7.2e-6 1.7e-4 6.2e-4
    for i, f in enumerate(("h", "q", "rhsh", "rhsq")):
        assert maxdiff(st[i], g[f]) <= ftol[i]
    env.close()
    # warm-up from rest with the reference's excitation signal (sloshing/init.py)
    env = V.VecSloshing(1, DEV, dtype, None)
    env.reset()
    t = 0.0
    for _ in range(env.n_warmup):
        env.step(np.array([env.signal(t, env.dt_act)]))
        t += env.dt_act
    st = env.get_state().cpu().numpy()[0]
    wtol = (tol, tol) if dtype == "f64" else (6e-6, 3.5e-5)                              # f32 measured 6.2e-7, 3.8e-6
    assert maxdiff(st[0], g["warm_h"]) <= wtol[0] and maxdiff(st[1], g["warm_q"]) <= wtol[1]
    env.close()


@pytest.mark.parametrize("L,nsteps", [(1.7, 6), (4.0, 4)])
def test_sloshing_other_lengths_vs_oracle_f64(L, nsteps):
    """Tank lengths other than the packaged one (nx = 80 L = 136 / 320: odd thread counts, two waves per
    replica), started from a smooth synthetic free surface; float64 against the oracle."""
    nx = int(80 * L)
    x = (np.arange(nx + 2) - 0.5) / nx
    init = np.zeros((2, nx + 2))
    init[0] = 1.0 + 0.05 * np.cos(np.pi * x)
    B = 3
    rng = np.random.default_rng(int(10 * L))
    acts = rng.uniform(-1, 1, (nsteps, B))
    env = V.VecSloshing(B, DEV, "f64", init, L=L)
    env.reset()
    oracles = [O.sloshing(init_fields=init, L=L) for _ in range(B)]
    for o in oracles:
        o.reset()
    for k in range(nsteps):
        obs, rwd, done, _, _ = env.step(acts[k])
        st = env.get_state().cpu().numpy()
        for b, o in enumerate(oracles):
            ob, rw, dn, _, _ = o.step([acts[k, b]])
            assert maxdiff(obs[b].cpu().numpy(), ob) <= 1e-12 and maxdiff(float(rwd[b]), rw) <= 1e-12
            assert maxdiff(st[b][0], o.h) <= 1e-12 and maxdiff(st[b][1], o.q) <= 1e-12
            assert bool(done[b]) == bool(dn)
    env.close()


def test_env1d_float64_bit_identical_to_oracle():
    """The float64 1D kernels follow the reference's operation order and are built without FMA contraction
    (env1d_f64.hip): fields and observations equal the oracle's -- which equals the reference bit for bit
    (tests/test_oracle.py) -- EXACTLY; only the rewards, which are reductions, differ in the last bits."""
    rng = np.random.default_rng(1)
    env, o = V.VecBurgers(1, DEV, "f64"), O.burgers()
    env.reset(); o.reset()
    for k in range(20):
        a, n = rng.uniform(-1, 1), rng.uniform(-0.1, 0.1)
        obs, rwd, _, _, _ = env.step(np.array([a]), np.array([n]))
        ob, rw, _, _, _ = o.step([a], n)
        assert np.array_equal(env.get_state().cpu().numpy()[0, 0], o.u) and np.array_equal(obs[0].cpu().numpy(), ob)
        assert maxdiff(float(rwd[0]), rw) <= 1e-13
    env.close()
    init = E.packaged_init("sloshing")
    env, o = V.VecSloshing(1, DEV, "f64", init), O.sloshing(init_fields=init)
    env.reset(); o.reset()
    for k in range(20):
        a = rng.uniform(-1, 1)
        obs, rwd, _, _, _ = env.step(np.array([a]))
        ob, rw, _, _, _ = o.step([a])
        st = env.get_state().cpu().numpy()[0]
        assert np.array_equal(st[0], o.h) and np.array_equal(st[1], o.q) and np.array_equal(obs[0].cpu().numpy(), ob)
        assert maxdiff(float(rwd[0]), rw) <= 1e-13
    env.close()
    init = E.packaged_init("shkadov")
    env, o = V.VecShkadov(1, DEV, "f64", init), O.shkadov(init_fields=init)
    o.rand_init = False
    env.reset(); o.reset()
    for k in range(8):
        a, nz = rng.uniform(-1, 1, 5), rng.uniform(-5e-4, 5e-4, 50)
        obs, rwd, _, _, _ = env.step(a[None], nz[None])
        ob, rw, _, _, _ = o.step(a.tolist(), nz)
        st = env.get_state().cpu().numpy()[0]
        assert np.array_equal(st[0], o.h) and np.array_equal(st[1], o.q) and np.array_equal(obs[0].cpu().numpy(), ob)
        assert maxdiff(float(rwd[0]), rw) <= 1e-13
    env.close()


def test_mixing_parallel_transport_passes_match_the_ordered_sweep():
    """float32 mixing solves the ordered part of the scalar transport -- the reference's in-place sweep reads the NEW west and south
    values (mixing.py:478-497): a lower-triangular system -- by its Neumann series, one parallel pass of all waves per term, as many
    terms as leave rho^(M+1) <= 2^-27 (ns2d_fast2_impl.h; option transport_iter = the most passes allowed, 0 = the ordered sweep).
    Over a full action step with four different lid actions: velocities, pressure and sweep counts are the SAME BITS (the scalar is
    passive), the scalar agrees with the ordered sweep to 2e-6 (measured 8e-7 after 250 timesteps: float32 rounding of two
    different summation orders; the truncation is 7e-9 per timestep) and stays as close to the float64 oracle; with fewer passes
    allowed than the measured rho needs (transport_iter = 3) the kernel takes the ordered sweep itself: the same bits."""
    acts = np.array([0, 1, 2, 3, 1, 3])
    out = {}
    for ti in (0, 24, 3):
        env = V.VecMixing(6, DEV, "f32")
        env.set_option("transport_iter", ti)
        env.reset()
        obs, rwd, _, _, _ = env.step(acts)
        env.step(acts[::-1].copy())
        env.check_status()
        assert env.kernel_name.startswith("ns2d_fast2")
        out[ti] = (env.get_state().clone(), env.sweeps.clone(), env.obs.clone(), env.rwd.clone())
        env.close()
    for k in (0, 1, 2):
        assert torch.equal(out[0][0][:, k], out[24][0][:, k])          # u, v, p
    assert torch.equal(out[0][1], out[24][1])
    d = (out[0][0][:, 3] - out[24][0][:, 3]).abs().max().item()
    assert 0.0 < d <= 2e-6, d                                          # the passes did run, and agree
    assert maxdiff(out[0][2].cpu().numpy(), out[24][2].cpu().numpy()) <= 2e-6 and maxdiff(out[0][3].cpu().numpy(), out[24][3].cpu().numpy()) <= 2e-6
    for a, b in zip(out[0], out[3]):
        assert torch.equal(a, b)                                       # too few passes allowed: the ordered sweep, bit for bit
    o = O.mixing()
    o.reset()
    o.step(int(acts[0]))
    o.step(int(acts[-1]))
    d_par, d_ord = maxdiff(dev2ref(out[24][0])[0][3], o.st[3]), maxdiff(dev2ref(out[0][0])[0][3], o.st[3])
    assert d_par <= d_ord + 2e-6 and d_par <= 4 * F32["mix_bench"]["C"], (d_par, d_ord)   # (two action steps: 500 timesteps)


# ---- the reference's non-geometric constructor arguments on the register-resident kernels -------------------
# (tag, the tag whose start state / reset() it shares, constructor arguments, action)
MIX_CTOR = [("mix_re50_pe1e3_a0", "mix_re50_pe1e3_a0", dict(re=50.0, pe=1.0e3), 0),
            ("mix_re50_pe1e3_a3", "mix_re50_pe1e3_a0", dict(re=50.0, pe=1.0e3), 3),
            ("mix_re200_pe1e5_a1", "mix_re200_pe1e5_a1", dict(re=200.0, pe=1.0e5, side=0.3, C0=2.0), 1),
            ("mix_re200_pe1e5_a2", "mix_re200_pe1e5_a1", dict(re=200.0, pe=1.0e5, side=0.3, C0=2.0), 2),
            ("mix_re400_pe2e3_a0", "mix_re400_pe2e3_a0", dict(re=400.0, pe=2.0e3, side=0.62, C0=0.5), 0)]
RAY_CTOR = [("ray_sgts5_ra5e4", dict(n_sgts=5, ra=5.0e4)), ("ray_sgts12_ra8e3_50x75", dict(n_sgts=12, ra=8.0e3, H=1.5)),
            ("ray_sgts3_ra2e5", dict(n_sgts=3, ra=2.0e5))]
# float32, four / five timesteps from the seeded states of tests/golden/ctor_args.npz (measured maxima in the comment of each test)
F32["mix_ctor"] = f32tol(1e-6, 1e-6, 4e-6, 2.5e-6, 2.5e-6, 3e-7)
# m (mixing, 5 cases): 3.3e-7 1.8e-7 6.0e-7 8.2e-7 2.4e-7 4.3e-8; m (rayleigh 50x50, 2 cases): 8.7e-9 1.4e-8 3.3e-8 1.9e-7 8.4e-8 2.6e-6
F32["ray_ctor"] = f32tol(8e-8, 1.4e-7, 3.3e-7, 1.9e-6, 8e-7, 2.6e-5)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("tag,t0,kw,act", MIX_CTOR, ids=[c[0] for c in MIX_CTOR])
def test_mixing_constructor_arguments_vs_reference(tag, t0, kw, act, dtype):
    """mixing(re, pe, side, C0) (mixing.py:21-34) on ns2d_fast2 (set_variant(1)): reset() -- patch side / C0, reward level -- and
    four timesteps from the seeded state, against what the REFERENCE itself returned (tests/golden/ctor_args.npz): float64 1e-9
    with its sweep counts (first solve 2472: the lid speed u_max = re nu / L enters through the boundary conditions), float32
    F32["mix_ctor"] with counts within max(3, 2 %).  re = 200 / 400 double / quadruple the lid speed, and with it the spectral
    radius the float32 kernel's parallel transport passes are counted from."""
    g = golden("ctor_args")
    B = 3
    env = V.VecMixing(B, DEV, dtype, **kw)
    env.set_ndt_act(4)
    assert env.set_variant(1) == 1
    obs0 = env.reset()
    obs0 = (obs0[0] if isinstance(obs0, tuple) else obs0).double().cpu().numpy()
    t = F32["mix_ctor"] if dtype == "f32" else f32tol(F64_TOL, F64_TOL, 50 * F64_TOL, F64_TOL, F64_TOL, F64_TOL)
    st = dev2ref(env.get_state())
    assert maxdiff(st[0][3], g[t0 + "_reset_C"]) <= (0 if dtype == "f64" else 1e-7)
    n = 3 * env.nx_obs_pts * env.ny_obs_pts
    assert maxdiff(obs0[0][-n:], g[t0 + "_reset_obs"][-n:]) <= t["obs"]
    st0 = np.stack([g["%s_%s0" % (t0, f)] for f in "uvpC"])
    env.set_state(np.tile(ref_to_dev(st0)[None], (B, 1, 1, 1)))
    obs, rwd, _, _, _ = env.step(np.full(B, act))
    env.check_status()
    assert env.kernel_name.startswith("ns2d_fast2")
    st, sw = dev2ref(env.get_state()), env.sweeps.cpu().numpy()
    for b in range(B):
        for i, F in enumerate("uvpC"):
            assert maxdiff(st[b][i], g["%s_%s" % (tag, F)]) <= t[F], (b, F)
        assert maxdiff(obs[b].double().cpu().numpy()[-n:], g[tag + "_obs"][-n:]) <= t["obs"]
        assert maxdiff(float(rwd[b]), float(g[tag + "_rwd"])) <= t["rwd"]
        assert np.all(np.abs(sw[b] - g[tag + "_itp"]) <= (1 if dtype == "f64" else np.maximum(3, 0.02 * g[tag + "_itp"]))), (sw[b], g[tag + "_itp"])
    env.close()


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("tag,kw", RAY_CTOR, ids=[c[0] for c in RAY_CTOR])
def test_rayleigh_constructor_arguments_vs_reference(tag, kw, dtype):
    """rayleigh(n_sgts, ra) (rayleigh.py:20-27) on the register-resident kernels (set_variant(1): ns2d_fast at 50x50, ns2d_fast2
    -- two rows per lane, odd ny, strips of 9 and 5 columns: the wrong kernel of round 5 -- at 50x75): five timesteps from the
    seeded state against what the REFERENCE returned; 12 and 3 segments on 50 cells leave the last bottom ghosts unwritten."""
    g = golden("ctor_args")
    B = 3
    env = V.VecRayleigh(B, DEV, dtype, None, **kw)
    env.set_ndt_act(5)
    assert env.set_variant(1) == 1
    env.reset()
    # (50x75 runs the two-rows-per-lane kernel: the tolerances of that family's on-demand grids)
    t = (F32["ray_ctor"] if env.ny <= 64 else F32["ray_jit", 2]) if dtype == "f32" else f32tol(F64_TOL, F64_TOL, 50 * F64_TOL, F64_TOL, F64_TOL, 50 * F64_TOL)
    st0 = np.stack([g["%s_%s0" % (tag, f)] for f in "uvpT"])
    env.set_state(np.tile(ref_to_dev(st0)[None], (B, 1, 1, 1)))
    a = np.tile(g[tag + "_action"][None], (B, 1))
    obs, rwd, _, _, _ = env.step(a)
    env.check_status()
    assert env.kernel_name == ("ns2d_fast2_step" if env.ny > 64 else "ns2d_fast_step")
    assert maxdiff(env.actions_norm.double().cpu().numpy()[0], g[tag + "_a_norm"]) <= (1e-15 if dtype == "f64" else 1e-7)
    st, sw = dev2ref(env.get_state()), env.sweeps.cpu().numpy()
    n = 3 * env.nx_obs_pts * env.ny_obs_pts
    for b in range(B):
        for i, F in enumerate("uvpT"):
            assert maxdiff(st[b][i], g["%s_%s" % (tag, F)]) <= t[F], (b, F)
        assert maxdiff(obs[b].double().cpu().numpy()[-n:], g[tag + "_obs"][-n:]) <= t["obs"]
        assert maxdiff(float(rwd[b]), float(g[tag + "_rwd"])) <= t["rwd"]
        assert np.all(np.abs(sw[b] - g[tag + "_itp"]) <= (1 if dtype == "f64" else np.maximum(3, 0.02 * g[tag + "_itp"]))), (sw[b], g[tag + "_itp"])
    env.close()


@pytest.mark.parametrize("scale,passes", [(45.0, True), (80.0, False)])
def test_mixing_transport_passes_beyond_twelve_terms_and_the_fallback(scale, passes):
    """The float32 mixing kernel counts its parallel transport passes from the measured rho = max(|aW| + |aS|) (mixing.py:478-497
    as a Neumann series; include/beacon_hip.h: "transport_iter").  At the reference's defaults rho = 0.2 -> 12 terms.  Here the
    seeded velocity field (max |u| + |v| = 0.069) is scaled (a) by 45: rho = 0.31 -> 16 terms -- with at most 12 allowed the kernel
    falls back to the ordered sweep (same bits as transport_iter 0), with the default 24 the passes run and agree with the ordered
    sweep and the float64 oracle; (b) by 80: rho = 0.55 -> 31 terms: the default itself falls back -- the bits of the ordered sweep."""
    g = golden("mixing_synth")
    st0 = np.stack([g["u0"] * scale, g["v0"] * scale, g["p0"], g["C0"]])
    out = {}
    for ti in (0, 12, 24):
        env = V.VecMixing(4, DEV, "f32")
        env.set_ndt_act(3)
        env.set_option("transport_iter", ti)
        env.reset()
        env.set_state(np.tile(ref_to_dev(st0)[None], (4, 1, 1, 1)))
        env.step(np.array([4, 4, 4, 4]))          # walls at rest (mixing.py:233): the seeded field alone sets rho
        env.check_status()
        out[ti] = (env.get_state().clone(), env.sweeps.clone())
        env.close()
    assert torch.equal(out[12][0], out[0][0]) and torch.equal(out[12][1], out[0][1])      # more than 12 needed: the ordered sweep
    same = torch.equal(out[24][0], out[0][0])
    assert same != passes, "rho of this state: the passes %s" % ("did not run" if passes else "ran")
    o = O.mixing()
    o.cfg.ndt_act = 3
    o.reset()
    o.st[:4] = st0
    o.step(np.int64(4))
    ref = dev2ref(out[24][0])[0]
    assert maxdiff(ref[3], o.st[3]) <= 2.5e-6
    for i in range(3):
        assert torch.equal(out[24][0][:, i], out[0][0][:, i])                                # velocities, pressure: the same bits


def test_sloshing_packed_float32_kernel_matches_the_unpacked_one_and_the_oracle():
    """float32 sloshing at the reference's grid runs sloshing_step_pk_k (one wave, two cells per v_pk instruction, fluxes per face,
    walls as selects); option one_wave = 2 selects the kernel it replaces.  40 action steps with random actions from the packaged
    developed state: measured (scripts/burgers_pk_diff.py) both kernels sit at the same distance from the float64 oracle -- h 2e-7
    after one step, 3e-6 after 40; q 6e-7 / 1.2e-5 -- and differ from each other by as much; the right-hand-side arrays are flux
    differences times 1 / dx = 80 (values up to 25: an ulp of the flux is 2e-6, 1.5e-4 after the division), so their bound is 2e-3."""
    # other tank lengths (nx = 160, 240: the far wall in another lane and cell), from rest, against the unpacked kernel
    rng = np.random.default_rng(5)
    for L in (2.0, 3.0):
        pair = []
        for ow in (1, 2):
            e = V.VecSloshing(4, DEV, "f32", None, L=L)
            e.set_option("one_wave", ow)
            e.reset()
            pair.append(e)
        for k in range(12):
            a = rng.uniform(-1, 1, 4)
            outs = [e.step(a) for e in pair]
            st = [e.get_state().double().cpu().numpy() for e in pair]
            assert maxdiff(st[0][:, :2], st[1][:, :2]) <= 4e-5 and maxdiff(st[0][:, 2:], st[1][:, 2:]) <= 2e-3, (L, k)
            assert maxdiff(outs[0][1].cpu().numpy(), outs[1][1].cpu().numpy()) <= 1e-6
        for e in pair:
            e.close()
    init = E.packaged_init("sloshing")
    rng = np.random.default_rng(3)
    B = 6
    envs = []
    for ow in (1, 2):
        e = V.VecSloshing(B, DEV, "f32", init)
        e.set_option("one_wave", ow)
        e.reset()
        envs.append(e)
    ors = [O.sloshing(init_fields=init) for _ in range(B)]
    for o in ors:
        o.reset()
    for k in range(40):
        a = rng.uniform(-1, 1, B)
        outs = [e.step(a) for e in envs]
        st = [e.get_state().double().cpu().numpy() for e in envs]
        assert maxdiff(st[0][:, :2], st[1][:, :2]) <= 4e-5 and maxdiff(st[0][:, 2:], st[1][:, 2:]) <= 2e-3, k
        assert maxdiff(outs[0][0].cpu().numpy(), outs[1][0].cpu().numpy()) <= 4e-5 and maxdiff(outs[0][1].cpu().numpy(), outs[1][1].cpu().numpy()) <= 1e-6
        assert torch.equal(outs[0][2], outs[1][2]) and torch.equal(outs[0][3], outs[1][3])
        for b, o in enumerate(ors):
            ob, rw, dn, tr, _ = o.step([a[b]])
            assert maxdiff(st[0][b, 0], o.h) <= 2e-5 and maxdiff(st[0][b, 1], o.q) <= 6e-5, (k, b)
            assert maxdiff(outs[0][0][b].double().cpu().numpy(), ob) <= 6e-5 and abs(float(outs[0][1][b]) - rw) <= 1e-6
            assert bool(outs[0][2][b]) == bool(dn) and bool(outs[0][3][b]) == bool(tr)
    for e in envs:
        e.close()


def test_sloshing_blowup_flag():
    env = V.VecSloshing(2, DEV, "f64", None)
    env.reset()
    st = env.get_state().cpu().numpy()
    st[1, 0, :100] = 2.5                      # h > 2 h_max -> done, not truncated (sloshing.py:156)
    env.set_state(st)
    obs, rwd, done, trunc, _ = env.step(np.zeros(2))
    assert done.cpu().tolist() == [0, 1] and trunc.cpu().tolist() == [0, 0]
    assert float(rwd[1]) != -10.0             # the -10 blow-up reward is dead code in the reference
    env.close()


@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_nan_state_is_flagged_as_blowup(dtype):
    """A NaN film height must end the episode with BCN_ST_BLOWUP in both precisions: the float32 limiters
    (v_med3_f32 / v_max_f32) drop a NaN operand instead of propagating it, and `h < -25 or h > 25`
    (shkadov.py:176) is false for NaN, so the magnitude test alone would let a poisoned replica run on."""
    env = V.VecShkadov(2, DEV, dtype, None, n_jets=5)
    env.reset()
    st = env.get_state().cpu().numpy()
    st[1, 0, 300] = np.nan
    env.set_state(st)
    obs, rwd, done, trunc, _ = env.step(np.zeros((2, 5)), np.zeros((2, 50)))
    assert done.cpu().tolist() == [0, 1] and trunc.cpu().tolist() == [0, 0]
    assert env.status.cpu().tolist() == [0, 2] and float(rwd[1]) == -1.0
    env.close()
    env = V.VecSloshing(2, DEV, dtype, None)
    env.reset()
    st = env.get_state().cpu().numpy()
    st[1, 0, 50] = np.nan
    env.set_state(st)
    obs, rwd, done, trunc, _ = env.step(np.zeros(2))
    assert done.cpu().tolist() == [0, 1] and env.status.cpu().tolist() == [0, 2]
    env.close()


# ---------------------------------------------------------------------------------------------
# full-size properties (BASELINE configs): size-independent invariants
# ---------------------------------------------------------------------------------------------
def test_rayleigh_fullsize_properties():
    """B=512, 128x64, f32: replica independence, permutation equivariance, discrete
    incompressibility after the corrector, bounded temperature, segment-mean-free actions."""
    B = 512
    g = golden("rayleigh_128x64")
    env = V.VecRayleigh(B, DEV, "f32", None, L=2.56, H=1.28)
    env.set_ndt_act(8)
    env.reset()
    st0 = np.stack([ref_to_dev(g[k]) for k in ("step0_u", "step0_v", "step0_p", "step0_T")])
    state = torch.as_tensor(st0, dtype=torch.float32, device=DEV)[None].repeat(B, 1, 1, 1)
    env.set_state(state)
    rng = np.random.default_rng(0)
    acts = rng.uniform(-1, 1, (B, 10))
    acts[B // 2:] = acts[:B // 2]              # second half repeats the first
    obs, rwd, done, trunc, _ = env.step(acts)
    env.check_status()
    o = obs.cpu().numpy()
    assert np.array_equal(o[:B // 2], o[B // 2:])
    st = env.get_state()
    assert torch.equal(st[:B // 2], st[B // 2:])
    u, v, T = st[:, 0].double(), st[:, 1].double(), st[:, 3].double()
    div = (u[:, 1:-1, 2:] - u[:, 1:-1, 1:-1]) / env.dx + (v[:, 2:, 1:-1] - v[:, 1:-1, 1:-1]) / env.dy
    assert float(div.abs().max()) < 5e-2       # Jacobi stops at sum(dphi^2) <= 1e-8, not at div = 0
    assert float(T[:, 1:-1, 1:-1].max()) < 1.3 and float(T[:, 1:-1, 1:-1].min()) > -0.6
    an = env.actions_norm.cpu().numpy()
    assert np.abs(an.mean(axis=1)).max() < 1e-6 and np.abs(an).max() <= 0.75 + 1e-6
    sw = env.sweeps.cpu().numpy()
    assert sw.min() >= 1 and sw.max() < 10000
    # the generic kernel on the same inputs: same physics to float32 rounding
    if env.set_variant(0) == 0 and env.kernel_name == "ns2d_generic_step":
        o1, s1 = obs.clone(), st.clone()
        env.set_state(state)
        env.set_stp(0)
        obs, _, _, _, _ = env.step(acts)
        assert float((obs[:, -96:] - o1[:, -96:]).abs().max()) < 5e-5
        assert float((env.get_state()[:, 3] - s1[:, 3]).abs().max()) < 5e-5
        assert int((env.sweeps.cpu() - torch.as_tensor(sw)).abs().max()) <= 3
    env.close()


def test_burgers_shkadov_fullsize_properties():
    env = V.VecBurgers(1024, DEV, "f32", nx=512)
    env.reset()
    obs, rwd, _, _, _ = env.step(np.zeros(1024), np.zeros(1024))
    assert float(rwd.abs().max()) == 0.0 and float((obs - 0.5).abs().max()) == 0.0   # steady state preserved
    obs, rwd, _, _, _ = env.step(np.ones(1024), np.full(1024, 0.05))
    assert torch.equal(obs[0], obs[1023]) and float(rwd.max()) < 0.0
    env.close()
    env = V.VecShkadov(1024, DEV, "f32", None, L0=699.2, n_jets=10)
    assert env.nx == 4096
    env.reset()
    obs, rwd, done, _, _ = env.step(np.zeros((1024, 10)), np.zeros((1024, 50)))
    st = env.get_state()
    assert float((st[:, 0] - 1).abs().max()) < 1e-5 and not bool(done.any())        # flat film is steady
    env.close()


# ---------------------------------------------------------------------------------------------
# "next" rows of SURVEY 8f: masks / auto-reset, on-device random warm-up, init generation,
# dump()/load() text formats, shkadov_separable
# ---------------------------------------------------------------------------------------------
def test_masked_step_and_auto_reset():
    g = golden("burgers")
    env = V.VecBurgers(4, DEV, "f64")
    env.reset()
    a, nz = np.array([0.5, -0.5, 0.25, 0.0]), np.array([0.01, -0.02, 0.03, 0.0])
    env.step(a, nz)
    st1 = env.get_state().clone()
    obs1, stp1 = env.obs.clone(), env.get_stp()
    env.step(a, nz, mask=np.array([1, 0, 1, 0]))            # replicas 1 and 3 must not move
    st2 = env.get_state()
    assert torch.equal(st2[1], st1[1]) and torch.equal(st2[3], st1[3])
    assert not torch.equal(st2[0], st1[0]) and not torch.equal(st2[2], st1[2])
    assert env.get_stp().tolist() == [stp1[0] + 1, stp1[1], stp1[2] + 1, stp1[3]]
    assert torch.equal(env.obs[1], obs1[1])
    # episode end -> reset_done() re-initialises exactly the finished replicas
    env.set_stp([199, 5, 199, 7])
    _, _, done, _, _ = env.step(a, nz)
    assert done.cpu().tolist() == [1, 0, 1, 0]
    keep = env.get_state().clone()
    obs, _ = env.reset_done()
    st = env.get_state()
    assert float((st[0] - 0.5).abs().max()) == 0 and float((st[2] - 0.5).abs().max()) == 0
    assert torch.equal(st[1], keep[1]) and torch.equal(st[3], keep[3])
    assert env.get_stp().tolist() == [0, 6, 0, 8]
    assert maxdiff(obs[0].cpu().numpy(), g["s0_reset_obs"]) == 0
    env.close()


def test_shkadov_device_random_warmup():
    """reset_random(): every replica gets its own number of uncontrolled steps (shkadov.py:119-123)."""
    init = E.packaged_init("shkadov")
    env = V.VecShkadov(4, DEV, "f64", init, n_jets=5, seed=3)
    n = torch.tensor([0, 2, 5, 3])
    nz = np.random.default_rng(1).uniform(-5e-4, 5e-4, (5, 50))
    env.reset()
    # reference path by hand, replica by replica, with the SAME noise per step index
    want = []
    for b in range(4):
        e1 = V.VecShkadov(1, DEV, "f64", init, n_jets=5)
        e1.reset()
        for i in range(int(n[b])):
            e1.step(None, nz[i][None])
        want.append((e1.get_state()[0].clone(), e1.obs[0].clone()))
        e1.close()
    env.reset()
    for i in range(5):
        env.step(None, np.tile(nz[i], (4, 1)), mask=(n > i))
    env.set_stp(0)
    st = env.get_state()
    for b in range(4):
        assert torch.equal(st[b], want[b][0]) and torch.equal(env.obs[b], want[b][1])
    obs, _ = env.reset_random(rand_steps=6)                 # device-drawn counts
    assert int(env.n_rand.max()) <= 6 and env.get_stp().tolist() == [0, 0, 0, 0]
    env.close()


def test_rayleigh_warmup_dump_load_roundtrip(tmp_path):
    """init generation (beacon/rayleigh/init.py) + the reference's text formats."""
    e = E.rayleigh(init=False, n_sgts=1)
    e.vec.set_ndt_act(20)
    e.reset()
    e.vec.warmup(5)                                        # develops nothing from exact rest, but runs the path
    e.close()
    e = E.rayleigh()
    e.reset()
    e.step(np.linspace(-1, 1, 10).tolist())
    f, a = str(tmp_path / "field.dat"), str(tmp_path / "act.dat")
    e.dump(f, a)
    blk = np.loadtxt(f)
    assert blk.shape == (4 * 52, 52)                       # 4 stacked (nx+2) x (ny+2) blocks (rayleigh.py:344-353)
    assert np.allclose(blk[3 * 52:], e.T, rtol=2e-5, atol=1e-9)      # '%.5e'
    e2 = E.rayleigh()
    e2.load(f)
    obs, _ = e2.reset()
    assert np.allclose(e2.T, e.T, rtol=2e-5, atol=1e-9) and np.allclose(obs[-48:], e.vec.obs[0].cpu().numpy()[-48:], rtol=2e-5, atol=1e-9)
    e.close(), e2.close()
    s = E.sloshing(init=False)
    s.reset()
    s.warmup()                                             # excitation warm-up of sloshing/init.py
    g = golden("sloshing")
    assert maxdiff(s.h, g["warm_h"]) <= 1e-12
    fs = str(tmp_path / "slosh.dat")
    s.dump(fs)
    s2 = E.sloshing(init=False)
    s2.load(fs)
    s2.reset()
    assert np.allclose(s2.h[1:-1], s.h[1:-1], rtol=2e-5)
    s.close(), s2.close()


def test_rayleigh_init_generation_on_the_gpu_matches_the_cpu_made_fixture():
    """SURVEY 8f-1 (rayleigh/init.py:13-28): the developed 128x64 state the bench starts from, generated ON THE GPU --
    VecRayleigh(n_sgts=1).develop(): seeded perturbed conduction state, n_warmup = 100 zero-action steps = 20 000
    timesteps through the float64 register-resident kernel -- against tests/golden/rayleigh_128x64_init.npz, which
    oracle/make_init.py produced with the float64 C oracle from the same start state.  The flow is a steady attractor
    (Ra = 1e4), so rounding differences do not grow: fields and the whole Nusselt history agree to 1e-7 (the float64
    kernel's own tolerance per step is 1e-9)."""
    from oracle import make_init
    z = np.load(os.path.join(GOLD, "rayleigh_128x64_init.npz"))
    env = V.VecRayleigh(2, DEV, "f64", None, L=float(z["L"]), H=float(z["H"]), n_sgts=1)
    _variant(env, 1)
    _, T0 = make_init.start_state(float(z["L"]), float(z["H"]), int(z["seed"]))
    assert np.array_equal(env.perturbed_conduction_state(int(z["seed"]))[3], T0)       # same start as the fixture's
    fields = env.develop(int(z["n_steps"]), int(z["seed"]))
    assert env.kernel_name == "ns2d_fast_step"
    nu = -env.warmup_rwd.double().cpu().numpy()
    assert nu.shape == (int(z["n_steps"]), 2) and np.array_equal(nu[:, 0], nu[:, 1])
    assert maxdiff(nu[:, 0], z["nu"]) <= 1e-7, maxdiff(nu[:, 0], z["nu"])
    for i, F in enumerate("uvpT"):
        assert maxdiff(fields[i], z["fields"][i]) <= (5e-6 if F == "p" else 1e-7), (F, maxdiff(fields[i], z["fields"][i]))
    assert (env.get_stp() == 0).all()
    env.close()
    # the generated state is what the constructor takes: reset() on it reproduces the fixture's reset observation
    a = V.VecRayleigh(1, DEV, "f64", fields, L=float(z["L"]), H=float(z["H"]))
    b = V.VecRayleigh(1, DEV, "f64", z["fields"], L=float(z["L"]), H=float(z["H"]))
    assert maxdiff(a.reset()[0].cpu().numpy(), b.reset()[0].cpu().numpy()) <= 1e-7
    a.close(), b.close()


def test_shkadov_separable_mirror():
    g = golden("shkadov_separable")
    e = E.shkadov_separable(n_jets=5)
    e.rand_init = False
    for k in range(5):
        obs, info = e.reset()
        assert obs.shape == (10,) and info is None and maxdiff(obs, g["reset_obs"][k]) == 0
    np.random.seed(6)
    k = 0
    for r in range(3):
        for j in range(5):
            obs, rwd, done, trunc, _ = e.step(g["actions"][r].tolist())
            assert maxdiff(obs, g["obs"][k]) <= 1e-12 and maxdiff(rwd, g["rwd"][k]) <= 1e-14
            assert [done, trunc] == g["done"][k].tolist() and e.stp == g["stp"][k]
            k += 1
    assert maxdiff(e.h, g["h"]) <= 1e-12
    e.close()


def test_render_writes_the_reference_layout(tmp_path, monkeypatch):
    """render() of the single-env mirrors (host side, matplotlib): same directories and file names as the
    reference's render() methods, dumps readable back with load()."""
    pytest.importorskip("matplotlib")
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("MPLBACKEND", "Agg")
    rng = np.random.default_rng(0)

    env = E.rayleigh(dtype="f32")
    env.reset()
    env.vec.set_ndt_act(3)
    env.reset()
    for k in range(2):
        env.step(rng.uniform(-1, 1, 10).tolist())
        env.render()
    for f in ("render/temperature/0.png", "render/temperature/1.png", "render/field/field_1.dat", "render/action/a_1.dat",
              "render/nu.dat"):
        assert os.path.getsize(tmp_path / f) > 0, f
    assert np.loadtxt(tmp_path / "render/nu.dat").shape == (2, 2)
    T = env.T
    env.load(str(tmp_path / "render/field/field_1.dat"))
    env.reset()
    assert maxdiff(env.T, T) < 1e-4            # '%.5e' text round trip
    env.close()

    os.rename(tmp_path / "render", tmp_path / "render_rayleigh")
    env = E.mixing(dtype="f32")
    env.vec.set_ndt_act(2)
    env.reset()
    env.step(2)
    env.render()
    assert os.path.getsize(tmp_path / "render/concentration/0.png") > 0
    assert np.loadtxt(tmp_path / "render/field/field_0.dat").shape == (4 * 102, 102)
    env.close()

    os.rename(tmp_path / "render", tmp_path / "render_mixing")
    for make, act, files in ((lambda: E.burgers(dtype="f32"), [0.3], ("gif/0.png", "fields/0.dat")),
                             (lambda: E.sloshing(dtype="f32"), [0.2], ("height/0.png", "field/field_0.dat", "action/jet_0.dat")),
                             (lambda: E.shkadov(dtype="f32"), [0.1] * 5, ("height/0.png", "field/field_0.dat", "action/jet_0.dat"))):
        env = make()
        if hasattr(env, "rand_init"):
            env.rand_init = False
        env.reset()
        env.step(act)
        env.render()
        for f in files:
            assert os.path.getsize(tmp_path / "render" / f) > 0, f
        env.close()
        os.rename(tmp_path / "render", tmp_path / ("render_%d" % len(os.listdir(tmp_path))))


def test_c_abi_error_behaviour():
    """Errors cross the C ABI as return codes + bcn_last_error() text and surface as BeaconHipError (the
    reference prints and exits): out-of-range configurations at create time and NULL handles
    (the Poisson overflow status word is covered by test_rayleigh_episode_end_and_overflow)."""
    from beacon_amd import _lib
    with pytest.raises(_lib.BeaconHipError, match="burgers cfg out of range"):
        V.VecBurgers(2, DEV, "f32", nx=10000)          # > 8192 cells
    with pytest.raises(_lib.BeaconHipError):
        V.VecRayleigh(0, DEV, "f32", None)             # batch must be positive
    lib = _lib.load()
    assert lib.bcn_batch(None) == 0 and lib.bcn_destroy(None) == 0            # NULL handle: inert, like free(NULL)
    assert lib.bcn_get_state(None, None, 0, None) == 1 and b"null" in lib.bcn_last_error()   # BCN_ERR_ARG
    # the sized counters call writes exactly words_per_replica words per replica (a caller built against an older header
    # that allocated 2 words per replica is not overrun), the unsized one BCN_COUNTER_WORDS; the API version is exported
    import ctypes as C
    assert lib.bcn_api_version() == _lib.API_VERSION
    env = V.VecRayleigh(3, DEV, "f32", E.packaged_init("rayleigh"))
    env.set_ndt_act(4)
    env.reset()
    env.step(np.zeros((3, 10)))
    full = env.get_counters()
    assert full.shape == (3, _lib.COUNTER_WORDS) and int(full[:, 1].min()) > 0
    for words in (2, 6):
        buf = (C.c_uint64 * (3 * words + 1))(*([0xdeadbeef] * (3 * words + 1)))
        assert lib.bcn_get_counters_n(env.h, buf, words, env._stream()) == 0
        got = np.frombuffer(buf, dtype=np.uint64)
        assert int(got[-1]) == 0xdeadbeef                                  # nothing behind the requested words
        got = got[:-1].reshape(3, words)
        assert np.array_equal(got[:, :min(words, 4)], full[:, :min(words, 4)]) and not got[:, 4:].any()
    assert lib.bcn_get_counters_n(env.h, buf, 0, env._stream()) == 1
    env.close()


def test_plain_c_consumer_of_the_c_abi(tmp_path):
    """include/beacon_hip.h + libbeacon_hip.so from a plain C program (gcc, HIP runtime API for the device
    buffers; no Python, no torch in the process): burgers, 4 replicas, float64, 3 steps, against the oracle."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None or not os.path.isdir("/opt/rocm/include"):
        pytest.skip("gcc / ROCm headers not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "burgers_smoke")
    libdir = os.path.join(root, "beacon_amd")
    cmd = ["gcc", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(root, "include"),
           os.path.join(root, "tests", "c_abi", "burgers_smoke.c"), "-o", exe, "-L" + libdir, "-lbeacon_hip",
           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "kernel ok" in r.stdout, (r.stdout[-500:], r.stderr[-1000:])
    got_obs, got_rwd = {}, {}
    for line in r.stdout.splitlines():
        t = line.split()
        if t[0] == "obs":
            got_obs[(int(t[1]), int(t[2]))] = np.array([float(x) for x in t[3:]])
        elif t[0] == "rwd":
            got_rwd[(int(t[1]), int(t[2]))] = float(t[3])
    for b in range(4):
        o = O.burgers()
        o.reset()
        for s_ in range(3):
            ob, rw, _, _, _ = o.step([0.25 * (b - 1.5) * (s_ + 1)], 0.02 * (b + 1) - 0.01 * s_)
            assert maxdiff(got_obs[(s_, b)], ob) <= 1e-12 and abs(got_rwd[(s_, b)] - rw) <= 1e-12


# ---------------------------------------------------------------------------------------------
# HIP graphs: step() only enqueues work, so a trainer can record it (VecEnv.capture)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["burgers", "sloshing", "shkadov", "rayleigh"])
def test_step_captured_in_a_hip_graph_replays_bit_identically(name):
    rng = np.random.default_rng(11)
    n, B = 3, 4

    def make():
        if name == "burgers":
            return V.VecBurgers(B, DEV, "f32", nx=512)
        if name == "sloshing":
            return V.VecSloshing(B, DEV, "f32")
        if name == "shkadov":
            return V.VecShkadov(B, DEV, "f32", E.packaged_init("shkadov"), n_jets=5)
        env = V.VecRayleigh(B, DEV, "f32", E.packaged_init("rayleigh"))
        env.set_ndt_act(20)
        return env

    eager, rec = make(), make()
    ashape = {"burgers": (2 * n, B), "sloshing": (2 * n, B), "shkadov": (2 * n, B, 5), "rayleigh": (2 * n, B, 10)}[name]
    acts = torch.as_tensor(rng.uniform(-1, 1, ashape), dtype=eager.tdtype, device=DEV)
    noise = None
    if name == "burgers":
        noise = torch.as_tensor(rng.uniform(-0.1, 0.1, (2 * n, B)), dtype=eager.tdtype, device=DEV)
    if name == "shkadov":
        noise = torch.as_tensor(rng.uniform(-5e-4, 5e-4, (2 * n, B, eager.ndt_act)), dtype=eager.tdtype, device=DEV)
    eager.reset()
    want = []
    for k in range(2 * n):
        obs, rwd, done, trunc, _ = eager.step(acts[k], None if noise is None else noise[k])
        want.append((obs.clone(), rwd.clone(), done.clone()))
    eager.check_status()
    rec.reset()
    a_in = acts[:n].clone()
    z_in = None if noise is None else noise[:n].clone()
    g = rec.capture(a_in, z_in, n_steps=n)                    # recording launches nothing
    for rep in range(2):                                      # second replay: the episode goes on with new inputs
        a_in.copy_(acts[rep * n:(rep + 1) * n])
        if z_in is not None:
            z_in.copy_(noise[rep * n:(rep + 1) * n])
        obs_seq, rwd_seq, done_seq, _ = g.replay()
        torch.cuda.synchronize()
        for k in range(n):
            o, r, d = want[rep * n + k]
            assert torch.equal(obs_seq[k], o) and torch.equal(rwd_seq[k], r) and torch.equal(done_seq[k], d), (name, rep, k)
    rec.check_status()
    assert torch.equal(rec.get_state(), eager.get_state())
    eager.close(); rec.close()


# ---------------------------------------------------------------------------------------------
# SURVEY 8e on real kernels: two ranks (two processes) sharing the one GPU of the box, device tensors over gloo
# (the nccl backend needs one device per rank: it runs only in the driver's multi-GPU bench)
# ---------------------------------------------------------------------------------------------
_GPU_WORKER = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
from beacon_amd import vec as V
from beacon_amd.dist import ShardedVecEnv
from beacon_amd.envs import packaged_init
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
dev, Bl = "cuda:0", 3


def episode(env, acts, G, sharded, rank, stp0, lo=0, hi=None):
    # one fixed sequence of trainer calls (VERDICT r03 item 1b): staggered episode ends, a single-env reset with a
    # mask, a masked step, auto-resets; `sharded`: payloads matter on rank 0 only, every rank makes the same calls
    log = []
    root = (not sharded) or rank == 0
    def rec(*t):
        if root:
            log.append([x.clone() for x in t if x is not None and torch.is_tensor(x)])
    o, _ = env.reset(); rec(o)
    (env.env if sharded else env).set_stp(stp0[lo:hi])
    for k in range(acts.shape[0]):
        if k == 2:
            m = torch.zeros(G, dtype=torch.uint8); m[1::3] = 1
            o, _ = env.reset(mask=m if root else True); rec(o)
        if k == 3:
            m = (torch.arange(G) %% 4 != 0)
            o, r, d, t, _ = env.step(acts[k] if root else None, mask=m if root else True)
        else:
            o, r, d, t, _ = env.step(acts[k] if root else None)
        rec(o, r, d, t)
        if k %% 2 == 1:
            o, _ = env.reset_done(); rec(o)
    return log


def same(a, b):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert len(x) == len(y)
        for u, v in zip(x, y):
            assert torch.equal(u, v)

rng = np.random.default_rng(5)
for name in ("rayleigh", "burgers"):
    def make(B):
        if name == "rayleigh":
            e = V.VecRayleigh(B, dev, "f32", packaged_init("rayleigh")); e.set_ndt_act(20); return e
        return V.VecBurgers(B, dev, "f32", nx=512)
    env = make(Bl)
    senv = ShardedVecEnv(env)
    o0, _ = senv.reset()
    shape = (3, world * Bl, 10) if name == "rayleigh" else (3, world * Bl)
    acts = torch.as_tensor(rng.uniform(-1, 1, shape), dtype=torch.float32)
    noise = None if name == "rayleigh" else torch.zeros((Bl,), dtype=torch.float32, device=dev)
    outs = [senv.step(acts[k] if rank == 0 else None, noise) for k in range(3)]
    if rank == 0:
        ref = make(world * Bl)
        ref.reset()
        assert torch.equal(o0, ref.obs), name
        nz = None if noise is None else torch.zeros((world * Bl,), dtype=torch.float32, device=dev)
        for k in range(3):
            obs, rwd, done, trunc, _ = ref.step(acts[k], nz)
            o, r, d, t, _ = outs[k]
            assert o.shape == obs.shape and torch.equal(o, obs) and torch.equal(r, rwd) and torch.equal(d, done), (name, k)
        assert int(senv.gather_status().max()) == 0
        ref.close()
    else:
        assert all(x[0] is None for x in outs)
    senv.close()
# staggered resets, masks and auto-reset through the sharded env, gather overlapped (double-buffered outputs)
G = world * Bl
acts = torch.as_tensor(rng.uniform(-1, 1, (6, G, 10)), dtype=torch.float32)
stp0 = np.array([97, 98, 99, 96, 99, 98], dtype=np.int32)
def make(B):
    e = V.VecRayleigh(B, dev, "f32", packaged_init("rayleigh")); e.set_ndt_act(10); return e
for overlap in (False, True):
    senv = ShardedVecEnv(make(Bl), overlap=overlap)
    got = episode(senv, acts, G, True, rank, stp0, senv.lo, senv.hi)
    if rank == 0:
        ref = make(G)
        want = episode(ref, acts, G, False, 0, stp0)
        assert sum(int(x[2].sum()) for x in want if len(x) == 4) >= 5       # episodes did end, at different steps
        same(got, want)
        ref.close()
    senv.close()
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


_RCCL_WORKER = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
from beacon_amd import vec as V
from beacon_amd.dist import ShardedVecEnv
from beacon_amd.envs import packaged_init
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)


def episode(env, acts, G, sharded, rank, stp0, lo=0, hi=None):
    # one fixed sequence of trainer calls (VERDICT r03 item 1b): staggered episode ends, a single-env reset with a
    # mask, a masked step, auto-resets; `sharded`: payloads matter on rank 0 only, every rank makes the same calls
    log = []
    root = (not sharded) or rank == 0
    def rec(*t):
        if root:
            log.append([x.clone() for x in t if x is not None and torch.is_tensor(x)])
    o, _ = env.reset(); rec(o)
    (env.env if sharded else env).set_stp(stp0[lo:hi])
    for k in range(acts.shape[0]):
        if k == 2:
            m = torch.zeros(G, dtype=torch.uint8); m[1::3] = 1
            o, _ = env.reset(mask=m if root else True); rec(o)
        if k == 3:
            m = (torch.arange(G) %% 4 != 0)
            o, r, d, t, _ = env.step(acts[k] if root else None, mask=m if root else True)
        else:
            o, r, d, t, _ = env.step(acts[k] if root else None)
        rec(o, r, d, t)
        if k %% 2 == 1:
            o, _ = env.reset_done(); rec(o)
    return log


def same(a, b):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert len(x) == len(y)
        for u, v in zip(x, y):
            assert torch.equal(u, v)

dev, B = "cuda:0", 6
rng = np.random.default_rng(5)
for name in ("rayleigh", "burgers"):
    def make():
        if name == "rayleigh":
            e = V.VecRayleigh(B, dev, "f32", packaged_init("rayleigh")); e.set_ndt_act(20); return e
        return V.VecBurgers(B, dev, "f32", nx=512)
    senv = ShardedVecEnv(make(), always_collective=True)
    assert senv.sh.collective and senv.sh.world == 1
    o0 = senv.reset()[0].clone()       # (outputs are views of the receive buffer: valid until the next step, as VecEnv's)
    acts = torch.as_tensor(rng.uniform(-1, 1, (3, B, 10) if name == "rayleigh" else (3, B)), dtype=torch.float32)
    noise = None if name == "rayleigh" else torch.zeros((B,), dtype=torch.float32, device=dev)
    outs = [tuple(x.clone() if x is not None else None for x in senv.step(acts[k], noise)) for k in range(3)]
    ref = make()
    ref.reset()
    assert torch.equal(o0, ref.obs), name
    for k in range(3):
        obs, rwd, done, trunc, _ = ref.step(acts[k], noise)
        o, r, d, t, _ = outs[k]
        assert o.is_cuda and o.shape == obs.shape and torch.equal(o, obs) and torch.equal(r, rwd) and torch.equal(d, done), (name, k)
    assert int(senv.gather_status().max()) == 0
    ref.close(); senv.close()
# the gather on a side stream beside the next step (double-buffered outputs): step_async, results taken one step late
def make():
    e = V.VecRayleigh(B, dev, "f32", packaged_init("rayleigh")); e.set_ndt_act(10); return e
acts = torch.as_tensor(rng.uniform(-1, 1, (8, B, 10)), dtype=torch.float32, device=dev)
for late, own_stream in ((1, False), (2, True)):
    # results taken `late` steps late (2 = as late as the ring of three output buffers allows), on the step stream or on a
    # consumer stream of their own (the next gather into that receive area must wait for the consumer's reads)
    senv = ShardedVecEnv(make(), always_collective=True, overlap=True)
    assert senv.overlap and senv._side is not None and len(senv.env.out_bufs) == 3
    ref = make(); ref.reset(); senv.reset()
    cons = torch.cuda.Stream(device=dev) if own_stream else torch.cuda.current_stream(dev)
    pend, outs = [], []
    def take():
        with torch.cuda.stream(cons):
            outs.append([x.clone() for x in pend.pop(0).wait()[:4]])
    for k in range(8):
        pend.append(senv.step_async(acts[k]))
        if len(pend) == late + 1:
            take()
    while pend:
        take()
    torch.cuda.synchronize()
    for k in range(8):
        o, r, d, t, _ = ref.step(acts[k])
        assert torch.equal(outs[k][0], o) and torch.equal(outs[k][1], r) and torch.equal(outs[k][2], d), (late, k)
    ref.close(); senv.close()
# masks, single-env resets and auto-reset through the sharded env == the plain env, bit for bit
stp0 = np.array([97, 98, 99, 96, 99, 98], dtype=np.int32)
for overlap in (False, True):
    senv = ShardedVecEnv(make(), always_collective=True, overlap=overlap)
    got = episode(senv, acts, B, True, 0, stp0)
    ref = make()
    want = episode(ref, acts, B, False, 0, stp0)
    assert sum(int(x[2].sum()) for x in want if len(x) == 4) >= 5
    same(got, want)
    ref.close(); senv.close()
torch.cuda.synchronize()
dist.barrier()
dist.destroy_process_group()
print("rccl ok", torch.cuda.nccl.version())
"""


def test_one_rank_through_rccl(tmp_path):
    """The production backend on the box's one GPU: a world of ONE rank whose scatter and packed gather are forced through
    torch.distributed's "nccl" backend (= RCCL on ROCm) instead of being skipped -- communicator set-up, a send / receive to
    itself per collective, device buffers.  Two ranks cannot share a GPU under RCCL (that is the gloo test below); more
    than one device is the driver's bench.  Results must equal the plain env's bit for bit."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "rccl_worker.py"
    script.write_text(_RCCL_WORKER % root)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29741", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert p.returncode == 0 and "rccl ok" in p.stdout, p.stdout[-3000:]


def test_bench_multi_gpu_path_over_rccl_with_one_rank():
    """`bench.py --force-dist`: the N > 1 code path of the bench (process group on the "nccl" backend with the device id,
    barriers and the max-reduction around the timed region, the packed gather inside it, the strong-scaling loop) with a
    world of ONE rank on the box's one GPU, self-launched and under torch.distributed.run as the driver starts it."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = [os.path.join(root, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "2", "--warmup", "1", "--no-cpu", "--no-secondary"]
    for cmd in ([sys.executable] + common,
                [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                 "--master-port", "29757"] + common):
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
        r = subprocess.run(cmd, env=dict(env, HSA_ENABLE_IPC_MODE_LEGACY="0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        d = json.loads(lines[0])
        assert d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 1000 and d["config"]["kernel"] == "ns2d_fast_sched"
        assert d["strong"]["global_batch"] == 512 and d["strong"]["value"] > 1000


def test_two_ranks_share_the_gpu_over_gloo(tmp_path):
    """ShardedVecEnv end to end with the real kernels: rank 0 scatters the global actions, every rank steps its shard on
    the GPU, one packed gather brings obs / rwd / status / done / trunc back -- bit-identical to one process stepping the
    global batch (replicas are independent)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "gpu_worker.py"
    script.write_text(_GPU_WORKER % root)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29733", WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert "rank %d ok" % r in o


# ---- on-demand kernels: first-use self-check ---------------------------------------------
def _fuzz_ids():
    from beacon_amd import jit
    return ["%s-%dx%d-%s%s" % (("rayleigh", "mixing")[c[3]], *jit._grid(c[0], c[1], c[3]), "f64" if c[2] else "f32", "-args" if c[4] else "")
            for c in jit.fuzz_cases()]


@pytest.mark.parametrize("idx", range(37), ids=_fuzz_ids())
def test_jit_fuzzed_grids_agree_with_the_generic_kernel(idx):
    """37 random domain sizes (seeded; the reference takes any L, H: rayleigh.py:20-27, mixing.py:20-28) across every
    kernel family -- one / two rows per lane and the hybrid, float32 and float64 -- each through the comparison that guards a
    plugin's first use (beacon_amd/jit.py: six timesteps, plain launch and ticket scheduler, against the generic kernel):
    float64 1e-9 with sweep counts within 1, float32 2e-4 with counts within max(3, 2 %).  The last 15 (round 6) have ODD ny
    in every (family, precision, env) class and draw the reference's other constructor arguments too: n_sgts, ra / re, pe,
    side, C0.  (The plugins are compiled by __graft_entry__.build().)"""
    from beacon_amd import jit
    L, H, f64, kind, kw = jit.fuzz_cases()[idx]
    dt = "f64" if f64 else "f32"
    mk = (lambda B: V.VecRayleigh(B, DEV, dt, None, L=L, H=H, **kw)) if kind == 0 else (lambda B: V.VecMixing(B, DEV, dt, L=L, H=H, **kw))
    env = mk(2)             # the constructor itself runs the self-check of a plugin that has no verdict yet
    p = getattr(env, "_plugin", None)
    assert p is not None and p.verified is True, "no verified plugin for this grid"
    env.close()
    ok, rep = jit.compare_with_generic(mk, kind, f64)       # and once more here, whatever the marker files say
    assert ok, rep


@pytest.mark.parametrize("L,H,live,R", [(2.2, 1.28, 12, 14), (1.5, 1.0, 5, 10), (1.04, 1.0, 3, 7), (2.5, 1.1, 13, 16)])
def test_float64_one_body_kernels_with_dead_columns(L, H, live, R):
    """float64 one-row-per-lane kernels whose strips are not all equally wide run ONE body since round 6: the last strip is R columns
    wide like the others, `live` of them exist, the others are dead (ns2d_fast_impl.h: DEADC) -- round 4's two-body kernel
    of 110x64 computed one wrong word per timestep.  12 timesteps with other constructor arguments (n_sgts = 7, ra = 3e4), plain
    launch and ticket scheduler, against the generic kernel: 1e-9 with sweep counts within 1 (measured: 7e-16, equal)."""
    from beacon_amd import jit
    nx, ny = int(50 * L), int(50 * H)
    m = jit.choose(nx, ny, True, 0)
    assert m["rows"] == 1 and m["gf"] == 2 and m["R"] == R and nx - (m["nw"] - 1) * R == live
    mk = lambda B: V.VecRayleigh(B, DEV, "f64", None, L=L, H=H, n_sgts=7, ra=3.0e4)
    env = mk(2)
    assert env._plugin is not None and env._plugin.verified is True and env.kernel_name in ("ns2d_fast_step", "ns2d_generic_step")
    env.close()
    ok, rep = jit.compare_with_generic(mk, 0, True, ndt=12, batch=4)
    assert ok, rep


def test_jit_no_plugin_of_the_test_grids_was_refused():
    """Every grid of jit.TEST_GRIDS and of the fuzz list: its plugin is built, attaches, and carries a PASSED self-check on
    this box (a refused one is already an error through JitWarning; this is the inventory -- no `.bad` marker anywhere in
    beacon_amd/_jit/ except the deliberately broken plugin's own, which its test removes)."""
    from beacon_amd import jit
    for nx, ny, f64, kind in jit.TEST_GRIDS + [k for k in jit.fuzz_grid_keys() if k not in jit.TEST_GRIDS]:
        n = 50.0 if kind == 0 else 100.0
        L, H = [c / n if int(n * (c / n)) == c else (c + 0.5) / n for c in (nx, ny)]
        dt = "f64" if f64 else "f32"
        env = V.VecRayleigh(2, DEV, dt, None, L=L, H=H) if kind == 0 else V.VecMixing(2, DEV, dt, L=L, H=H)
        assert (env.nx, env.ny) == (nx, ny)
        p = getattr(env, "_plugin", None)
        assert p is not None and p.verified is True, "%dx%d %s kind %d: no verified plugin" % (nx, ny, dt, kind)
        env.close()
    bad = [f for f in os.listdir(jit.JIT_DIR) if f.endswith(".bad")]
    assert not bad, bad


def test_jit_self_check_refuses_a_broken_plugin_and_keeps_the_generic_kernel():
    """A deliberately wrong plugin (csrc/jit/ns2d_jit.hip built with -DBCN_JIT_BREAK: the kernel runs with 1.5 dt) must not be
    attached: the first env that asks for it compares it with the generic kernel, warns, leaves a `.bad` marker, and steps on
    the generic kernel with the generic kernel's results; a later env does not even run the comparison."""
    import warnings
    from beacon_amd import jit

    class Broken(V.VecRayleigh):
        _plugin_defs = {"BCN_JIT_BREAK": 1}
    path = jit.build_plugin(75, 50, True, 0, extra_defs=Broken._plugin_defs)
    assert path is not None, "the broken test plugin was not built (run __graft_entry__.build())"
    for mark in (".ok", ".bad"):
        if os.path.exists(path + mark):
            os.remove(path + mark)
    jit._LOADED.pop((75, 50, True, 0, tuple(sorted(Broken._plugin_defs.items()))), None)
    try:
        with pytest.warns(UserWarning, match="DISAGREES with the generic kernel"):
            env = Broken(2, DEV, "f64", None, L=1.5, H=1.0)
        assert getattr(env, "_plugin", None) is None and os.path.exists(path + ".bad")
        ref = V.VecRayleigh(2, DEV, "f64", None, L=1.5, H=1.0)
        a = np.random.default_rng(1).uniform(-1, 1, (2, 10))
        for e in (env, ref):
            e.set_ndt_act(4)
            if e is ref:
                assert e.set_variant(0) == 0          # (set_ndt_act re-creates the handle: the variant is chosen behind it)
            e.reset()
            e.set_state(jit._seeded_rayleigh_state(e))
            e.step(a)
            e.check_status()
        assert env.kernel_name == "ns2d_generic_step"
        assert torch.equal(env.get_state(), ref.get_state()) and torch.equal(env.sweeps, ref.sweeps)
        env.close()
        ref.close()
        # the verdict is remembered: a later env (a later process: the cache of loaded plugins cleared) skips the plugin at once
        jit._LOADED.clear()
        with pytest.warns(UserWarning, match="failed its self-check earlier"):
            env2 = Broken(2, DEV, "f64", None, L=1.5, H=1.0)
        assert getattr(env2, "_plugin", None) is None
        env2.close()
        # the sound plugin of the same grid is unaffected
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            good = V.VecRayleigh(2, DEV, "f64", None, L=1.5, H=1.0)
        assert good._plugin.verified is True
        good.close()
    finally:
        for mark in (".ok", ".bad"):
            if os.path.exists(path + mark):
                os.remove(path + mark)
        jit._LOADED.clear()


# ---- the stop rule on an adversarial right-hand side -------------------------------------------
@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_adversarial_residual_growth_behind_the_stop_sweep_never_moves_the_stop(dtype):
    """tests/golden/rayleigh_adversarial_50x50.npz: the first solve passes the reference's stop test at sweep 5, and the
    residual is ABOVE tol again for sweeps 6..15 (the reference's norm d'(I + G)d is not monotone: it can grow by up to 1.030,
    scripts/weighted_norm_bound.py).  A plan that skips sweep 5 and evaluates in that stretch finds "not converged" and, if it
    trusted that, would stop up to twelve sweeps late.  Every evaluation plan of the register-resident kernel must return the
    reference's sweep 5 and the reference's fields -- as it stands and with the test hook that lengthens every skip
    (plan_overshoot 0..12: the landings then fall on sweeps 5..16); the guarded plan (3, the float32 default) does so because a
    landing that does not clear BCN_CONV_GUARD * tol is repeated under the proven plan (counters [2], [3]); the UNGUARDED plan 2
    under the same hook stops late, which is what the guard is for."""
    g = golden("rayleigh_adversarial_50x50")
    want = int(g["stop_sweep"])
    B = 4
    ref_state = ref_to_dev(g["final_state"])

    def run(opts, variant=1):
        env = V.VecRayleigh(B, DEV, dtype, None, ra=float(g["ra"]))
        env.set_ndt_act(1)
        assert env.set_variant(variant) == variant
        for k_, v_ in opts.items():
            env.set_option(k_, v_)
        env.reset()
        env.set_state(np.tile(ref_to_dev(g["state"])[None], (B, 1, 1, 1)))
        env.step(np.zeros((B, 10)))
        st = env.check_status()
        out = (env.sweeps[:, 0].cpu().numpy().copy(), env.get_state().double().cpu().numpy(), env.get_counters(), st, env.kernel_name)
        env.close()
        return out
    tol_f = 1e-9 if dtype == "f64" else 2e-6
    base = run({"conv_plan": 0})
    assert base[4].startswith("ns2d_fast") and (base[0] == want).all(), (base[4], base[0])
    assert maxdiff(base[1][0], ref_state) <= tol_f
    gen = run({}, variant=0)
    assert (gen[0] == want).all()
    flagged = 0
    for plan in (1, 3):
        for ov in range(0, 13):
            r = run({"conv_plan": plan, "plan_overshoot": ov})
            assert (r[0] == want).all(), (plan, ov, r[0])
            assert np.array_equal(r[1], base[1]), (plan, ov)          # bit for bit the every-sweep result
            flagged += int(r[2][:, 2].sum()) if plan == 3 else 0
    assert flagged > 0                                                # the guard did catch unverified landings
    for plan in (2, 3):
        r = run({"conv_plan": plan, "verify_conv": 1})
        assert (r[0] == want).all() and not (r[3] & 4).any(), (plan, r[0], r[3])
    # the hole the guard closes: the same hook under the unguarded plan stops late for some skip lengths
    late = [int(run({"conv_plan": 2, "plan_overshoot": ov})[0][0]) for ov in (4, 6, 8, 12)]
    assert max(late) > want, late
    # ... and so does the guarded plan when it is handed slow-mode constants that are NOT this grid's (a cutoff of 1e-4 with
    # bound 1 claims that everything but the constant has decayed after one sweep: the landing guard collapses to 1.001 tol)
    late = []
    for ov in (4, 6, 8, 12):
        env = V.VecRayleigh(B, DEV, dtype, None, ra=float(g["ra"]))
        env.set_ndt_act(1)
        assert env.set_variant(1) == 1 and len(env.slow_mode_bound()) == 2          # 50x50: built in
        env.set_slow_mode_bound([(1.0e-4, 1.0)])
        env.set_option("conv_plan", 3)
        env.set_option("plan_overshoot", ov)
        env.reset()
        env.set_state(np.tile(ref_to_dev(g["state"])[None], (B, 1, 1, 1)))
        env.step(np.zeros((B, 10)))
        late.append(int(env.sweeps[0, 0]))
        env.close()
    if dtype == "f32":               # (the float64 50x50 kernel evaluates sweep by sweep: its guard is BCN_CONV_GUARD whatever the constants)
        assert max(late) > want, late


def test_slow_mode_landing_guard_never_changes_a_result():
    """conv_plan 3 verifies a landing against min(BCN_CONV_GUARD, the slow-mode guard) * tol (include/beacon_hip.h:
    bcn_set_slow_mode_bound; beacon_amd/stoprule.py).  With the grid's constants (built in for 128x64; computed at creation for a
    JIT grid), without them (guard 1.035 alone) and with every sweep evaluated (conv_plan 0) the sweep counts and fields of a
    full action step of the bench workload are the same bits, no landing is left unverified -- and the constants pay: fewer
    cycles inside the Jacobi loop than without them."""
    got = {}
    for name in ("slow", "global", "literal"):
        env, init, acts = _bench_workload(300, 1, "f32")
        have = env.slow_mode_bound()
        assert [c for c, _ in have] == [0.9, 0.8] and all(1.0 <= b < 1.01 for _, b in have), have
        if name == "global":
            env.set_slow_mode_bound([])
            assert env.slow_mode_bound() == []
        if name == "literal":
            env.set_option("conv_plan", 0)
        env.step(acts[0])
        env.check_status()
        c = env.get_counters()
        got[name] = (env.sweeps.clone(), env.get_state().clone(), env.obs.clone(), float(c[:, 0].sum()), int(c[:, 2].sum()))
        env.close()
    for name in ("global", "literal"):
        for a, b in zip(got["slow"][:3], got[name][:3]):
            assert torch.equal(a, b), name
    assert got["slow"][4] == 0 and got["global"][4] == 0            # plan 3 leaves no unverified stop behind
    assert got["slow"][3] < 0.99 * got["global"][3] < 0.99 * got["literal"][3], [got[k][3] for k in got]
    # a grid whose constants the host computes (75x50: beacon_amd/stoprule.py, cached beside its kernel plugin)
    ref = None
    for plan in (3, 0):
        env = V.VecRayleigh(6, DEV, "f32", None, L=1.5, H=1.0)
        env.set_ndt_act(30)
        assert env.set_variant(1) == 1 and len(env.slow_mode_bound()) == 2, env.slow_mode_bound()
        env.set_option("conv_plan", plan)
        x, y = (np.arange(env.nx + 2) - 0.5) / env.nx, (np.arange(env.ny + 2) - 0.5) / env.ny
        st0 = np.zeros((4, env.nx + 2, env.ny + 2))
        st0[3] = (0.5 - y)[None, :] + 0.08 * np.sin(2 * np.pi * x * 1.5)[:, None] * np.sin(np.pi * y)[None, :]
        env.reset()
        env.set_state(np.tile(ref_to_dev(st0)[None], (6, 1, 1, 1)))
        env.step(np.random.default_rng(3).uniform(-1, 1, (6, 10)))
        env.check_status()
        out = (env.sweeps.clone(), env.get_state().clone())
        env.close()
        if ref is None:
            ref = out
        else:
            assert torch.equal(ref[0], out[0]) and torch.equal(ref[1], out[1])


# ---- the two bindings of the C ABI -------------------------------------------------------------
def test_2d_env_observations_are_views_of_the_envs_history_array():
    """rayleigh.py:243-262 / mixing.py:237-258: get_obs() refills the env's own `self.obs` [4, 3, nx_obs_pts, ny_obs_pts] and returns
    np.reshape(self.obs, [-1]) -- a view, so an observation kept from reset() shows the next step's values (SURVEY 8b "Ownership").
    The drop-in mirrors do the same."""
    for env, a in ((E.rayleigh(dtype="f32"), [0.1] * 10), (E.mixing(dtype="f32"), np.int64(2))):
        env.vec.set_ndt_act(3)
        o0, _ = env.reset()
        assert o0.dtype == np.float64 and o0.shape == (env.n_obs_tot,) and np.shares_memory(o0, env.obs)
        kept = o0.copy()
        o1 = env.step(a)[0]
        assert np.shares_memory(o1, env.obs) and np.shares_memory(o0, o1)
        assert np.array_equal(o0, o1) and not np.array_equal(o0, kept)      # the kept handle moved with the env
        n = 3 * env.nx_obs_pts * env.ny_obs_pts
        assert np.array_equal(o1[2 * n:3 * n], kept[3 * n:])                 # history shifted by one slot (newest last)
        env.close()


def test_torch_ops_refuse_short_or_foreign_tensors():
    """every tensor a torch.ops.beacon.* call receives is checked against the handle -- element count (obs, rwd, done,
    trunc, status, sweeps [B][ndt_act], actions, noise), dtype and device -- BEFORE the library sees a pointer: a short buffer is an
    error, not an out-of-bounds write on the device.  Meta kernels are registered (the ops return nothing)."""
    from beacon_amd import torch_ext
    ops = torch_ext.load()
    assert ops is not None
    env = V.VecBurgers(4, DEV, "f32", nx=512)
    env.reset()
    h = env.h.value
    good = dict(obs=env.obs, rwd=env.rwd, done=env.done, trunc=env.trunc, status=env.status)
    a = torch.zeros(4, device=DEV)

    def call(**kw):
        t = dict(good, **kw)
        ops.burgers_step(h, kw.get("actions", a), kw.get("noise", None), t["obs"], t["rwd"], t["done"], t["trunc"], t["status"])
    call()
    for bad in (dict(done=torch.zeros(3, dtype=torch.uint8, device=DEV)), dict(status=torch.zeros(2, dtype=torch.int32, device=DEV)),
                dict(actions=torch.zeros(3, device=DEV)), dict(noise=torch.zeros(5, device=DEV)), dict(rwd=torch.zeros(4, device=DEV, dtype=torch.float64)),
                dict(trunc=torch.zeros(4, dtype=torch.uint8)), dict(obs=torch.zeros(env.obs.numel() - 1, device=DEV))):
        with pytest.raises(RuntimeError):
            call(**bad)
    env.close()
    env = V.VecMixing(2, DEV, "f32")
    env.set_ndt_act(3)
    env.use_torch_ops(True)
    env.reset()
    with pytest.raises(RuntimeError):      # sweeps is [B][ndt_act]
        ops.mixing_step(env.h.value, None, env.obs, env.rwd, env.done, env.trunc, env.status, torch.zeros((2, 2), dtype=torch.int32, device=DEV))
    env.close()
    m = torch.empty(4, device="meta")
    ops.burgers_reset(0, m)                # a Meta kernel exists: nothing to compute, nothing dereferenced


def test_torch_ops_and_ctypes_bindings_step_every_env_bit_identically():
    """reset() / step() through torch.ops.beacon.* (csrc/torch/beacon_torch.cpp: one dispatcher call, stream read in C++) and
    through ctypes call the same bcn_* entry points: three steps, a masked reset and another step of every env family give
    the same bits -- observations, rewards, done flags, sweep counts, solver state."""
    from beacon_amd import torch_ext
    assert torch_ext.load() is not None, "the torch extension is not built (run __graft_entry__.build())"
    rng = np.random.default_rng(11)
    B = 6
    cases = [
        ("rayleigh", lambda: V.VecRayleigh(B, DEV, "f32", E.packaged_init("rayleigh")), lambda k: rng.uniform(-1, 1, (B, 10)), None),
        ("mixing", lambda: V.VecMixing(B, DEV, "f32"), lambda k: rng.integers(0, 4, (B,)), None),
        ("burgers", lambda: V.VecBurgers(B, DEV, "f64", nx=512), lambda k: rng.uniform(-1, 1, (B,)), lambda: rng.uniform(-0.1, 0.1, (B,))),
        ("shkadov", lambda: V.VecShkadov(B, DEV, "f32", E.packaged_init("shkadov"), n_jets=5), lambda k: rng.uniform(-1, 1, (B, 5)),
         lambda: rng.uniform(-5e-4, 5e-4, (B, 50))),
        ("sloshing", lambda: V.VecSloshing(B, DEV, "f64", E.packaged_init("sloshing")), lambda k: rng.uniform(-1, 1, (B,)), None),
    ]
    for name, mk, act, nz in cases:
        acts = [act(k) for k in range(4)]
        noise = [nz() if nz else None for _ in range(4)]
        outs = []
        for use_ops in (True, False):
            env = mk()
            assert env.use_torch_ops(use_ops) == use_ops
            if name in ("rayleigh", "mixing"):
                env.set_ndt_act(12)
                env.use_torch_ops(use_ops)
            log = [env.reset()[0].clone()]
            for k in range(3):
                o, r, d, t, _ = env.step(acts[k], noise[k]) if nz else env.step(acts[k])
                log += [o.clone(), r.clone(), d.clone(), t.clone()]
            m = torch.tensor([1, 0, 1, 0, 0, 1], dtype=torch.uint8, device=DEV)
            log.append(env.reset(mask=m)[0].clone())
            o, r, d, t, _ = env.step(acts[3], noise[3]) if nz else env.step(acts[3])
            log += [o.clone(), r.clone(), env.get_state()]
            if hasattr(env, "sweeps"):
                log.append(env.sweeps.clone())
            env.check_status()
            env.close()
            outs.append(log)
        assert len(outs[0]) == len(outs[1])
        for x, y in zip(*outs):
            assert torch.equal(x, y), name
    # wrong dtypes / shapes are refused by the op itself, with the library untouched
    env = V.VecMixing(B, DEV, "f32")
    with pytest.raises(RuntimeError, match="dtype"):
        env._ops["mixing_step"](env.h.value, None, env.obs.double(), env.rwd, env.done, env.trunc, env.status, env.sweeps)
    with pytest.raises(RuntimeError, match="elements"):
        env._ops["mixing_reset"](env.h.value, env.obs[:2].contiguous())
    env.close()


def test_builtin_register_resident_kernels_pass_the_plugin_self_check():
    """beacon_amd.jit.selftest(): every register-resident kernel built into the library (128x64, 50x50, 100/150/200x50,
    100x100 rayleigh (float32) and mixing; float32 and float64 where built) through the comparison that guards an on-demand kernel's
    first use -- the check to repeat after rebuilding the library with another toolchain."""
    from beacon_amd import jit
    res = jit.selftest(DEV)
    assert len(res) == 10
    for name, (ok, rep) in res.items():
        assert ok, (name, rep)
        assert "ns2d_fast" in rep and "generic" not in rep.split(";")[0], (name, rep)     # the fast kernels did run
