"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol the header
declares, derived env parameters match the reference constructors, the
host-only lorenz env matches the golden episodes, and the replica-sharding collectives work
across two processes (gloo)."""
import ctypes
import json
import os
import re
import subprocess
import time
import sys

import numpy as np
import pytest

from conftest import ROOT, golden


def test_library_exports_every_declared_symbol():
    from beacon_amd import _lib, build
    hdr = open(os.path.join(ROOT, "include", "beacon_hip.h")).read()
    declared = set(re.findall(r"BCN_API\s+[\w\s\*]+?\b(bcn_\w+)\s*\(", hdr))
    assert len(declared) >= 30
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    if build.hipcc() is None and not os.path.exists(build.LIB):
        pytest.skip("no hipcc and no prebuilt library")
    L = _lib.load()            # resolves every symbol; no compute call without a GPU
    for name in declared:
        assert hasattr(L, name)
    assert b"gfx950" in L.bcn_version()
    hdr_ver = int(re.search(r"#define BCN_API_VERSION (\d+)", hdr).group(1))
    assert L.bcn_api_version() == hdr_ver == _lib.API_VERSION          # header, library and binding agree
    raw = ctypes.CDLL(_lib.lib_path())
    for name in declared:
        getattr(raw, name)


def test_product_path_has_no_cpu_fallback():
    import torch
    from beacon_amd import vec
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        vec.VecRayleigh(2)
    # and the product package never imports the oracle
    for root, _, files in os.walk(os.path.join(ROOT, "beacon_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def _derive(cls, *a, **k):
    from beacon_amd import vec
    c = getattr(vec, cls)
    return c._derive(c.__new__(c), *a, **k)


def test_derived_parameters_match_reference_defaults():
    g = golden("rayleigh_default")
    r = _derive("VecRayleigh")
    assert [r.nx, r.ny, r.ndt_act, r.n_act, r.n_sgts, r.nx_sgts, r.nx_obs_pts, r.ny_obs_pts, r.nx_obs,
            r.ny_obs] == g["params"].tolist()
    assert np.array_equal(np.array([r.L, r.H, r.dx, r.dy, r.dt, r.pr, r.ra, r.Tc, r.Th, r.C]), g["fparams"])
    g = golden("rayleigh_128x64")
    r = _derive("VecRayleigh", 2.56, 1.28)
    assert [r.nx, r.ny, 5, r.n_act, r.n_sgts, r.nx_sgts, r.nx_obs_pts, r.ny_obs_pts, r.nx_obs,
            r.ny_obs] == g["params"].tolist()
    assert np.array_equal(np.array([r.L, r.H, r.dx, r.dy, r.dt, r.pr, r.ra, r.Tc, r.Th, r.C]), g["fparams"])
    assert r.n_obs_tot == 384
    g = golden("mixing_a0")
    m = _derive("VecMixing")
    assert [m.nx, m.ny, 3, m.n_act, m.nx_obs_pts, m.ny_obs_pts, m.nx_obs, m.ny_obs] == g["params"].tolist()
    assert np.array_equal(np.array([m.L, m.H, m.dx, m.dy, m.dt, m.re, m.pe, m.u_max, m.side, m.C0]), g["fparams"])
    assert m.ndt_act == int(g["full_ndt_act"]) == 250 and (m.i_min, m.i_max, m.j_min, m.j_max) == (25, 75, 25, 75)
    g = golden("burgers")
    b = _derive("VecBurgers")
    assert [b.nx, b.ndt_act, b.n_act, b.ctrl_pos, b.n_obs_pts] == g["params"].tolist()
    assert np.array_equal(np.array([b.L, b.dx, b.dt, b.amp, b.sigma, b.u_target]), g["fparams"])
    g = golden("shkadov")
    for tag, kw in (("j5", dict(n_jets=5)), ("j10", dict(n_jets=10)), ("n4096", dict(L0=699.2, n_jets=10))):
        s = _derive("VecShkadov", **kw)
        assert [s.nx, s.ndt_act, s.n_act, s.n_jets, s.jet_pos, s.jet_hw, s.jet_space, s.l_obs, s.l_rwd,
                s.n_obs, s.n_interp] == g[tag + "_params"].tolist()
        assert np.array_equal(np.array([s.L, s.dx, s.dt, s.delta, s.sigma, s.jet_amp, s.eps]), g[tag + "_fparams"])
    g = golden("sloshing")
    s = _derive("VecSloshing")
    assert [s.nx, s.ndt_act, s.n_act, s.n_interp, s.n_obs] == g["params"].tolist()
    assert np.array_equal(np.array([s.L, s.dx, s.dt, s.g, s.amp, s.alpha]), g["fparams"])


def test_packaged_init_fields():
    from beacon_amd.envs import packaged_init
    g = golden("rayleigh_default")
    ray = packaged_init("rayleigh")
    assert np.array_equal(ray, np.stack([g["u_init"], g["v_init"], g["p_init"], g["T_init"]]))
    g = golden("shkadov")
    shk = packaged_init("shkadov")
    assert np.array_equal(shk[0][:1100], g["j5_h_init"]) and np.array_equal(shk[1][:1350], g["j10_q_init"])
    g = golden("sloshing")
    slo = packaged_init("sloshing")
    assert np.array_equal(slo[0], g["h_init"]) and np.array_equal(slo[1], g["q_init"])


def test_lorenz_host_env_matches_reference_episodes():
    import beacon_amd
    g = golden("lorenz")
    for tag in ("a0", "a1", "a2", "rnd"):
        e = beacon_amd.lorenz()
        o0, info = e.reset()
        assert info is None and np.array_equal(o0, g[tag + "_reset_obs"])
        for k, a in enumerate(g[tag + "_actions"]):
            o, r, d, t, info = e.step(np.int64(a))
            assert np.array_equal(o, g[tag + "_obs"][k]) and r == g[tag + "_rwd"][k]
            assert [d, t] == g[tag + "_done"][k].tolist() and info is None
        assert np.array_equal(np.array(e.hx), g[tag + "_hx"])
        assert e.action_space.n == 3 and e.observation_space.shape == (6,)


def test_shard_bounds():
    from beacon_amd.dist import shard_bounds
    assert [shard_bounds(512, 8, r) for r in (0, 3, 7)] == [(0, 64), (192, 256), (448, 512)]
    with pytest.raises(ValueError):
        shard_bounds(10, 4, 0)


_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from beacon_amd.dist import ReplicaSharder, ShardedVecEnv, shard_bounds
from beacon_amd.vec import out_layout, unpack_outputs
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
B = 6
sh = ReplicaSharder(B)
assert sh.global_batch == B * world
lo, hi = shard_bounds(sh.global_batch, world, rank)
full = torch.arange(sh.global_batch * 3, dtype=torch.float32).reshape(sh.global_batch, 3)
mine = sh.scatter_actions(full if rank == 0 else None, torch.empty((B, 3), dtype=torch.float32))
assert torch.equal(mine, full[lo:hi]), (rank, mine)
# packed per-step outputs: ONE collective per step, receive buffer reused across steps
buf = torch.zeros((out_layout(B, 3, 4)["bytes"],), dtype=torch.uint8)
obs, rwd, status, done, trunc = unpack_outputs(buf, B, 3, torch.float32)
for step in range(3):
    obs[:] = mine * 2 + step
    rwd[:] = mine.sum(1)
    status[:] = rank
    done[:] = (torch.arange(B) + rank).to(torch.uint8)
    trunc[:] = step
    g = sh.gather_outputs(buf, 3, torch.float32)
    if rank == 0:
        g_obs, g_rwd, g_status, g_done, g_trunc = g
        assert torch.equal(g_obs, full * 2 + step) and torch.equal(g_rwd, full.sum(1))
        assert g_status.tolist() == [0] * B + [1] * B and g_trunc.tolist() == [step] * (2 * B)
        assert g_done.shape == (sh.global_batch,) and g_done[B:].tolist() == [(i + 1) %% 256 for i in range(B)]
    else:
        assert g is None


class Env:                                               # CPU stand-in with the VecEnv surface
    action_is_int, n_actions, batch, obs_dim, tdtype, device = False, 3, B, 3, torch.float32, torch.device("cpu")
    def __init__(self):
        self.out_buf = torch.zeros((out_layout(B, 3, 4)["bytes"],), dtype=torch.uint8)
        self.obs, self.rwd, self.status, self.done, self.trunc = unpack_outputs(self.out_buf, B, 3, torch.float32)
        self.gen = torch.Generator()
    def reset(self, mask=None):
        self.obs.fill_(-1.0); return self.obs, None
    def step(self, a, noise=None, mask=None):
        self.obs[:] = a + 1; self.rwd[:] = a.sum(1); self.done.fill_(1)
        return self.obs, self.rwd, self.done, self.trunc, None


senv = ShardedVecEnv(Env(), seed=5)
assert senv.env.gen.initial_seed() == 5 + lo           # noise streams differ per rank
assert senv.gather_status() is None                    # cached copy of the last step's words: none yet
e9 = Env(); e9.seed = 9; e9.gen.manual_seed(9)
assert ShardedVecEnv(e9).env.gen.initial_seed() == 9 + lo   # default: the env's own seed, offset by the shard
o, _ = senv.reset()
assert (o is None) == (rank != 0) and (rank != 0 or o.shape == (2 * B, 3))
o, r, d, t, _ = senv.step(full if rank == 0 else None)
if rank == 0:
    assert torch.equal(o, full + 1) and torch.equal(r, full.sum(1)) and d.tolist() == [1] * (2 * B)
else:
    assert o is None
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_replica_sharding_collectives_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29713", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert "rank %d ok" % r in o


_WORKER2 = r"""
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import torch
import torch.distributed as dist
from beacon_amd.dist import ShardedVecEnv
from cpu_env import CpuVecEnv
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
B, NA = 6, 4                                          # replicas per rank, episode length
G = B * world
gen = torch.Generator().manual_seed(3)
acts = torch.rand((14, G, 3), generator=gen)
stp0 = (torch.arange(G) %% 4).to(torch.int32)            # staggered episode ends


def script(env, sharded):
    # one fixed sequence of trainer calls; `sharded`: the env is a ShardedVecEnv (payloads matter on rank 0 only)
    log = []
    root = (not sharded) or rank == 0
    def rec(*t):
        if root:
            log.append([x.clone() for x in t if x is not None and torch.is_tensor(x)])
    o, _ = env.reset(); rec(o)
    if sharded:                                        # staggered episode ends: set behind the reset on both sides
        env.env.stp[:] = stp0[env.lo:env.hi]
    else:
        env.stp[:] = stp0
    for k in range(12):
        if k == 5:                                     # the trainer resets single envs (rayleigh.py:89-99)
            m = torch.zeros(G, dtype=torch.uint8); m[[1, 4, 7, 10]] = 1
            o, _ = env.reset(mask=m if root else True); rec(o)
        if k == 8:                                     # a masked step: rows of skipped replicas keep their values
            m = (torch.arange(G) %% 3 != 0)
            o, r, d, t, _ = env.step(acts[k] if root else None, mask=m if root else True)
        else:
            o, r, d, t, _ = env.step(acts[k] if root else None)
        rec(o, r, d, t)
        if k %% 2 == 1:
            o, _ = env.reset_done(); rec(o)
    return log


# reference: ONE process stepping the global batch
ref = CpuVecEnv(G, n_act=NA)
ref_log = script(ref, False)
assert sum(int(x[2].sum()) for x in ref_log if len(x) == 4) >= 20      # episodes did end, at different steps
for overlap in (False, True):
    loc = CpuVecEnv(B, n_act=NA)
    senv = ShardedVecEnv(loc, overlap=overlap)
    assert senv.overlap == overlap and (len(loc.out_bufs) == senv.NBUF) == overlap
    got = script(senv, True)
    if rank == 0:
        assert len(got) == len(ref_log)
        for a, b in zip(got, ref_log):
            assert len(a) == len(b)
            for x, y in zip(a, b):
                assert torch.equal(x, y), (overlap, x, y)
    else:
        assert got == []
# step_async: two gathers in flight, results consumed one step late, buffers rotate
loc = CpuVecEnv(B, n_act=NA); senv = ShardedVecEnv(loc, overlap=True)
ref = CpuVecEnv(G, n_act=NA); ref.reset(); senv.reset()
pend, outs = [], []
for k in range(6):
    pend.append(senv.step_async(acts[k] if rank == 0 else None))
    if len(pend) == 2:
        outs.append([None if x is None else x.clone() for x in pend.pop(0).wait()[:4]])
outs.append([None if x is None else x.clone() for x in pend.pop(0).wait()[:4]])
for k in range(6):
    o, r, d, t, _ = ref.step(acts[k])
    if rank == 0:
        assert torch.equal(outs[k][0], o) and torch.equal(outs[k][1], r) and torch.equal(outs[k][2], d)
    else:
        assert outs[k][0] is None
assert (senv.gather_status() is not None) == (rank == 0)
senv.close()
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def _run_sharded_script(tmp_path, world, extra_env=None):
    script = tmp_path / "worker2.py"
    script.write_text(_WORKER2 % (ROOT, ROOT))
    s = __import__("socket").socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), OMP_NUM_THREADS="1")
    env.update(extra_env or {})
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert "rank %d ok" % r in o


def test_sharded_env_masks_auto_reset_and_overlapped_gather_gloo_world2(tmp_path):
    """ShardedVecEnv.reset(mask) / step(mask=) / reset_done() and the double-buffered, asynchronous
    gather (step_async) over two processes -- every output of a 12-step script with staggered episode ends, single-env
    resets, a masked step and auto-resets equals, bit for bit, ONE process stepping the global batch."""
    _run_sharded_script(tmp_path, 2)


def test_sharded_env_masks_auto_reset_and_overlapped_gather_gloo_world8(tmp_path):
    """the same script of trainer calls at the world size of the metric (8 ranks, 48 replicas), with the
    call-shape check of ShardedVecEnv switched on (BEACON_DIST_CHECK=1: every rank is in the same call, with a mask
    placeholder wherever rank 0 passes a mask)."""
    _run_sharded_script(tmp_path, 8, {"BEACON_DIST_CHECK": "1"})


def test_sharded_env_rejects_bad_buffer_counts_and_mismatched_calls(tmp_path):
    """BEACON_NBUF / nbuf are validated in the constructor; a rank that passes mask=None where rank 0 passes a mask
    is caught by check_calls instead of meeting rank 0 in the wrong collective."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from beacon_amd.dist import ShardedVecEnv
    from cpu_env import CpuVecEnv
    with pytest.raises(ValueError, match="at least 2"):
        ShardedVecEnv(CpuVecEnv(4), nbuf=0)
    os.environ["BEACON_NBUF"] = "three"
    try:
        with pytest.raises(ValueError, match="not an integer"):
            ShardedVecEnv(CpuVecEnv(4))
    finally:
        del os.environ["BEACON_NBUF"]
    assert ShardedVecEnv(CpuVecEnv(4), nbuf=4).NBUF == 4 and ShardedVecEnv(CpuVecEnv(4)).NBUF == 3
    worker = tmp_path / "worker3.py"
    worker.write_text(r"""
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import torch, torch.distributed as dist
from beacon_amd.dist import ShardedVecEnv
from cpu_env import CpuVecEnv
rank = int(os.environ["RANK"])
dist.init_process_group("gloo")
senv = ShardedVecEnv(CpuVecEnv(4), check_calls=True)
senv.reset()
try:
    senv.step(torch.zeros((8, 3)) if rank == 0 else None, mask=torch.ones(8, dtype=torch.uint8) if rank == 0 else None)
    print("rank", rank, "no error")
except RuntimeError as e:
    assert "disagree on the call" in str(e), e
    print("rank", rank, "caught")
dist.destroy_process_group()
""" % (ROOT, ROOT))
    s = __import__("socket").socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(worker)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and "rank %d caught" % r in o, o


def test_bench_self_launch_and_scalings_on_cpu_stub():
    """`python bench.py --gpus 2` with no launcher: the parent starts one process per rank, rank 0 prints ONE JSON
    line; weak scaling doubles the global batch, strong scaling shards the batch of 512.  Run with the CPU stand-in
    env over gloo (the line is marked data=stub: launcher / sharding / gather plumbing only)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    for scaling, per_gpu, glob in (("weak", 16, 32), ("strong", 8, 16)):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                            "--batch", "16", "--scaling", scaling, "--stub", "--backend", "gloo"],
                           env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout
        d = json.loads(lines[0])
        assert d["n_gpus"] == 2 and d["scaling"] == scaling and d["data"] == "stub" and d["steps"] == 3
        assert d["config"]["global_batch"] == glob and ("%d replicas per GPU" % per_gpu) in d["config"]["workload"]
        assert abs(d["value"] - glob * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]
        if scaling == "weak":      # the weak line also carries the strong-scaling reading of the same --batch
            st = d["strong"]
            assert st["global_batch"] == 16 and st["replicas_per_gpu"] == 8 and st["steps"] == 3 and st["value"] > 0
        else:
            assert "strong" not in d


def test_bench_world8_rehearsal_on_cpu_stub_carries_eight_rank_records():
    """`bench.py --gpus 8 --stub --backend gloo`, weak and strong: ONE line whose `ranks` block holds one
    record per rank (distinct processes, contiguous replica ranges covering the global batch, own clocks) and a `topology`
    block -- what makes the driver's N = 8 line self-evidencing (there with device uuids / PCI ids and the RCCL version)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    for scaling, per_gpu, glob in (("weak", 16, 128), ("strong", 2, 16)):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1",
                            "--batch", "16", "--scaling", scaling, "--stub", "--backend", "gloo"],
                           env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout
        d = json.loads(lines[0])
        assert d["n_gpus"] == 8 and d["scaling"] == scaling and d["config"]["global_batch"] == glob
        rk = d["ranks"]
        assert [x["rank"] for x in rk] == list(range(8)) and len({x["pid"] for x in rk}) == 8
        assert [x["global_replica_range"] for x in rk] == [[i * per_gpu, (i + 1) * per_gpu] for i in range(8)]
        assert all(x["replicas"] == per_gpu and x["ms_per_step_own_clock"] > 0 and x["jacobi_sweeps_timed"] > 0 for x in rk)
        assert d["topology"] == {"world_size": 8, "backend": "gloo", "launcher": "bench.py self-launch"}
        # the job's clock is the slowest rank's: no rank's own clock is above it
        assert max(x["ms_per_step_own_clock"] for x in rk) <= d["ms_per_step"] * (1 + 1e-9)
        if scaling == "weak":
            assert d["strong"]["global_batch"] == 16 and d["strong"]["replicas_per_gpu"] == 2


def test_bench_force_dist_runs_the_collectives_with_one_rank_on_cpu_stub():
    """`--gpus 1 --force-dist` (the one-GPU rehearsal of the N > 1 path: tests/test_gpu_parity.py runs it over RCCL): a
    world of one rank still builds the process group and sends its barriers, reductions and the packed gather through the
    backend; here with the CPU stand-in env over gloo."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "3", "--warmup", "1",
                        "--batch", "16", "--stub", "--backend", "gloo"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["data"] == "stub" and d["config"]["global_batch"] == 16
    assert d["strong"]["global_batch"] == 16 and d["strong"]["replicas_per_gpu"] == 16


def test_bench_launcher_stops_every_rank_when_one_dies():
    """A rank that dies before the rendezvous (BENCH_FAULT_RANK: os._exit(3) in front of init_process_group) must not
    leave its peer waiting there: the self-launching parent polls all children, terminates the survivors and exits
    non-zero within seconds -- far below the rendezvous timeout, which is set to 120 s here."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["BENCH_FAULT_RANK"] = "1"
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch", "16", "--stub", "--backend", "gloo", "--dist-timeout", "120"],
                       env=env, capture_output=True, text=True, timeout=100)
    dt = time.time() - t0
    assert r.returncode != 0, r.stdout + r.stderr
    assert dt < 60, dt
    assert "rank 1 exited with code 3" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]      # no line from a half-dead job


def test_vortex_host_env_matches_reference_episodes():
    import beacon_amd
    g = golden("vortex")
    for tag in ("zero", "rnd"):
        e = beacon_amd.vortex()
        o0, info = e.reset()
        assert info is None and np.array_equal(o0, g[tag + "_reset_obs"])
        for k, a in enumerate(g[tag + "_actions"]):
            o, r, d, t, info = e.step(a)
            assert np.array_equal(o, g[tag + "_obs"][k]) and r == g[tag + "_rwd"][k]
            assert [d, t] == g[tag + "_done"][k].tolist()
        assert np.array_equal(np.array(e.hx), g[tag + "_hx"])
    assert e.action_space.shape == (2,) and e.observation_space.shape == (8,)


def test_lorenz_vortex_render_and_dump(tmp_path, monkeypatch):
    """Host-only envs: render() writes the reference's png/ layout, dump() its text columns."""
    pytest.importorskip("matplotlib")
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("MPLBACKEND", "Agg")
    from beacon_amd.lorenz import lorenz
    from beacon_amd.vortex import vortex
    env = lorenz()
    env.reset()
    for k in range(3):
        env.step(np.int64(k % 3))
        env.render()
    assert os.path.getsize(tmp_path / "png/gif/2.png") > 0
    d = np.loadtxt(tmp_path / "png/lorenz.dat")
    assert d.shape == (4, 4) and abs(d[-1, 0] - 3 * env.dt_act) < 1e-12 and np.allclose(d[-1, 1:], env.x, rtol=1e-5)
    os.rename(tmp_path / "png", tmp_path / "png_lorenz")
    env = vortex()
    env.reset()
    env.step([0.2, -0.4])
    env.render()
    assert os.path.getsize(tmp_path / "png/gif/0.png") > 0
    env.dump(str(tmp_path / "vortex.dat"))
    d = np.loadtxt(tmp_path / "vortex.dat")
    assert d.shape == (1 + env.ndt_act, 7) and np.allclose(d[-1, 5:], [env.kmod, env.kphase], rtol=1e-5)


_SPACES_CHECK = r"""
import sys, numpy as np
import gymnasium
from gymnasium import spaces as gsp
from beacon_amd import spaces, vec, lorenz, vortex
assert spaces.have_gymnasium()
def mk(cls, *a, **k):
    c = getattr(vec, cls)
    return c._derive(c.__new__(c), *a, **k)._make_spaces()
r = mk("VecRayleigh", 2.56, 1.28)
assert isinstance(r.action_space, gsp.Box) and isinstance(r.observation_space, gsp.Box)
assert r.action_space.low == -0.75 and r.action_space.high == 0.75 and r.action_space.shape == (10,)   # rayleigh.py:75-78
assert r.action_space.dtype == np.float32 and r.observation_space.shape == (384,)
assert np.array_equal(r.observation_space.high, np.ones(384)) and np.array_equal(r.observation_space.low, -np.ones(384))
m = mk("VecMixing")
assert isinstance(m.action_space, gsp.Discrete) and m.action_space.n == 4 and m.observation_space.shape == (192,)
b = mk("VecBurgers")
assert np.array_equal(b.observation_space.low, np.zeros(5)) and np.array_equal(b.observation_space.high, np.ones(5))
assert b.action_space.shape == (1,) and b.action_space.low == -1.0
s = mk("VecShkadov", n_jets=10)
assert s.action_space.shape == (10,) and s.observation_space.shape == (100,)
sl = mk("VecSloshing")
assert sl.observation_space.shape == (100,) and isinstance(sl.action_space, gsp.Box)
lz = lorenz()
assert isinstance(lz.action_space, gsp.Discrete) and lz.action_space.n == 3 and isinstance(lz.observation_space, gsp.Box)
vx = vortex()
assert isinstance(vx.action_space, gsp.Box) and vx.action_space.shape == (2,)
assert np.array_equal(vx.observation_space.high, np.ones(vx.n_obs) * 1.0e-4)
print("spaces ok")
"""


def test_mirrors_carry_real_gymnasium_spaces_when_gymnasium_is_importable():
    """rayleigh.py:75-86, mixing.py:61-70, ...: with a `gymnasium` on sys.path (here the capture stand-in package, the only
    one this container has) the mirrors' spaces are that package's Box / Discrete, built with the reference's arguments;
    without it, beacon_amd.spaces' own stand-ins with the same attributes."""
    stubs = os.path.join(ROOT, "oracle", "capture", "stubs")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, stubs, os.environ.get("PYTHONPATH", "")]))
    out = subprocess.run([sys.executable, "-c", _SPACES_CHECK], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "spaces ok" in out.stdout, out.stderr[-2000:]
    from beacon_amd import spaces
    try:
        import gymnasium  # noqa: F401
    except ImportError:
        assert not spaces.have_gymnasium()
        bx = spaces.box(-0.75, 0.75, (10,))
        assert isinstance(bx, spaces.Box) and bx.shape == (10,) and bx.low.dtype == np.float32 and bx.high[3] == np.float32(0.75)
        sb = spaces.sym_box(1.0, 7)
        assert sb.shape == (7,) and np.array_equal(sb.low, -np.ones(7, np.float32)) and sb.contains(np.zeros(7))
        d = spaces.discrete(4)
        assert d.n == 4 and d.contains(3) and not d.contains(4) and 0 <= d.sample(np.random.default_rng(0)) < 4


def test_torch_extension_builds_loads_and_registers_every_op():
    """north_star's "thin PyTorch-ROCm C-ABI extension" (VERDICT r04 item 8): csrc/torch/beacon_torch.cpp builds with g++ against
    the torch headers, links libbeacon_hip.so, and registers one torch.library op per (env, reset / step) -- for the CUDA (= ROCm)
    dispatch key ONLY: there is no CPU implementation to fall back to."""
    import shutil
    import torch
    from beacon_amd import build, torch_ext, vec
    if (shutil.which("g++") is None and torch_ext.stale()) or (build.hipcc() is None and not os.path.exists(build.LIB)):
        pytest.skip("no compiler and no prebuilt extension")
    path = torch_ext.build_ext()
    assert path and os.path.exists(path) and not torch_ext.stale()
    ops = torch_ext.load()
    assert ops is not None and vec._op_table() is not None
    for name in vec._OPS:
        schema = str(getattr(ops, name).default._schema)
        assert schema.startswith("beacon::%s(int handle" % name) and schema.endswith("-> ()"), schema
        assert "Tensor(a!)" in schema                                       # outputs are written in place
    assert len(vec._OPS) == 10
    with pytest.raises((NotImplementedError, RuntimeError)):                # CUDA key only: CPU tensors find no kernel
        ops.mixing_reset(0, torch.zeros(4))
