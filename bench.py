#!/usr/bin/env python3
"""Headline benchmark: aggregate env steps/s of rayleigh-v0 (BASELINE configs[3]: 128x64 grid, batch 512).

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong]

N > 1 without a launcher: this process touches no GPU and starts N children (one per GPU, env RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT); under `python -m torch.distributed.run
--nproc-per-node N ... bench.py --gpus N` the ranks are used as launched.  One process per GPU, RCCL
("nccl") only for the trainer-facing gather.

One "step" = one batched env.step(): 200 solver timesteps (BCs, predictor, Jacobi pressure Poisson to the
reference's tolerance, corrector, ordered scalar transport) + obs + reward for every replica, in ONE HIP
launch per GPU; with N > 1 the per-step gather of the packed outputs (obs, rwd, status, done, trunc: one
collective) to rank 0 is inside the timed region.
  --scaling weak   (default) 512 replicas PER GPU                     -> "scaling": "weak"
  --scaling strong the global batch of 512 sharded, 512/N per GPU     -> "scaling": "strong"
Inputs are synthetic and resident in HBM before the timed region: developed-flow initial state from
tests/golden/rayleigh_128x64_init.npz (float64 oracle warm-up), actions from
numpy.random.default_rng(1234).uniform(-1, 1, (steps, global batch, 10)), distinct per replica and step.
Prints ONE JSON line on rank 0."""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec (MI355X_MICROARCH.md)
N_CU, SIMD_PER_CU, CLK_HZ = 256, 4, 2.4e9
VALU_ISSUE_PEAK = N_CU * SIMD_PER_CU * CLK_HZ / 2.0   # wave64 VALU instructions/s: one per 2 cycles per SIMD-32
MIN_VALU_PER_CELL_SWEEP = 7      # e+w, +n, +s, ghost fma, cx fma, difference, residual fma


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=512, help="replicas per GPU (weak) / global batch (strong)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--variant", type=int, default=-1, help="-1 = best available, 0 = generic kernel")
    ap.add_argument("--sched", type=int, default=-1, help="scheduler mode of the fast kernels (bcn_set_sched)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline legs")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary configurations")
    ap.add_argument("--zero-actions", action="store_true",
                    help="diagnostic: uncontrolled steady flow (1 Jacobi sweep per timestep) -> non-Poisson cost")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (gloo with --stub)")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal on a one-GPU box: with --gpus 1, run the process group, barriers, reductions and the packed "
                         "gather of the N > 1 path through the backend all the same (a world of one rank over RCCL)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="N > 1 path: the packed gather as a blocking collective between two steps (round 3) instead of on a side "
                         "stream beside the next step (double-buffered outputs)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal on a one-GPU box: every rank uses cuda:0 (with --backend gloo; nccl needs one device per rank)")
    ap.add_argument("--L", type=float, default=2.56, help="domain length (nx = 50 L; rayleigh.py:26)")
    ap.add_argument("--H", type=float, default=1.28, help="domain height (ny = 50 H; rayleigh.py:27)")
    ap.add_argument("--gen-init", action="store_true",
                    help="develop the initial state on the GPU (VecRayleigh.develop: rayleigh/init.py's 100 uncontrolled steps, "
                         "float64) instead of loading the CPU-made fixture; implied for any grid other than 128x64")
    ap.add_argument("--dist-timeout", type=float, default=300.0,
                    help="seconds a rank waits in the rendezvous / a collective before it gives up (init_process_group timeout)")
    ap.add_argument("--no-strong", action="store_true",
                    help="N > 1, --scaling weak: skip the extra strong-scaling loop (global batch sharded) reported under `strong`")
    ap.add_argument("--stub", action="store_true",
                    help="CPU stand-in env (no GPU, no kernels): exercises launcher, sharding and gather only; "
                         "its line is marked data=stub and is not a measurement")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------
# self-launch: the parent never touches the GPU (no torch import, no HIP call)
# ------------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_children(n, argv, poll_s=0.1, grace_s=5.0):
    """Start one fresh process per rank and watch ALL of them: the first rank that ends with a non-zero code takes
    the others down (SIGTERM, SIGKILL after `grace_s`) and the parent returns that code at once -- a rank that died
    at init_process_group must not leave its peers waiting in the rendezvous until somebody's timeout.  The children
    are the exact PIDs started here (never a pattern), and they are ended on every way out of this function."""
    port = free_port()
    procs = []
    try:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
        rc = 0
        live = list(procs)
        while live and rc == 0:
            time.sleep(poll_s)
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0:
                    rc = abs(code) or 1
                    sys.stderr.write("bench.py: rank %d exited with code %d; stopping the other ranks\n"
                                     % (procs.index(p), code))
                    break
        return rc
    finally:
        alive = [p for p in procs if p.poll() is None]
        for p in alive:
            p.terminate()
        t_end = time.time() + grace_s
        for p in alive:
            try:
                p.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()


# ------------------------------------------------------------------------------------------------
# accounting
# ------------------------------------------------------------------------------------------------
def committed_profile(kernel_name, key, dtype="f32", grid_tag=None):
    """Per-dispatch counters from the committed rocprofv3 PMC passes (profiles/*_summary.json, produced by
    scripts/prof.sh: separate --pmc passes; HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 as the MI355X guide
    prescribes for gfx950).  Latest file wins; None if absent."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_summary.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        for name, c in d.get("pmc", {}).items():
            real = "<double" if dtype == "f64" else "<float"
            if grid_tag and "<" in name and grid_tag not in name:
                continue                         # another grid's instantiation of the same kernel template
            typed = "<float" in name or "<double" in name           # (burgers_step_pk_k<8> carries no element type: float32 only)
            if name.startswith(kernel_name) and (real in name or not typed) and key in c and "reset" not in name:
                best = {"value": c[key], "source": os.path.basename(f), "sweeps_per_dispatch": c.get("sweeps_per_dispatch")}
    return best


def algorithmic_bytes(nx, ny, sweeps, esz):
    """per interior cell 20 values per timestep + 3 values per Jacobi sweep."""
    cells = nx * ny
    return float(cells) * esz * (20.0 * sweeps.shape[1] * sweeps.shape[0] + 3.0 * float(sweeps.sum()))


def cpu_legs(state0, acts, cfg_kw, seconds=14.0):
    """CPU baselines on a bounded sample of the SAME work the GPU timed: the first `cores` replicas, starting
    from the GPU's state after the warm-up steps, with the actions of the first timed steps.
      port : the float64 scalar-C oracle (restatement of the reference's numba loops), OpenMP over envs
      numpy: the vectorised-NumPy restatement (oracle/numpy_port.py), one env on one core"""
    import ctypes as C
    import numpy as np
    from oracle import oracle as O
    from oracle import numpy_port as NP
    cores = os.cpu_count() or 1
    nenv = min(cores, state0.shape[0])
    e = O.rayleigh(init=False, **cfg_kw)
    n = (e.cfg.nx + 2) * (e.cfg.ny + 2)
    st = np.zeros((nenv, 8, n))
    st[:, :4] = np.swapaxes(state0[:nenv], -1, -2).reshape(nenv, 4, n)      # device [j][i] -> reference [i][j]
    obs = np.zeros((nenv, e.n_obs_tot))
    rwd = np.zeros(nenv)
    sw = np.zeros(nenv, dtype=np.int64)
    L = O.lib()
    done_steps, tot_sw, t0 = 0, 0, time.perf_counter()
    while done_steps < acts.shape[0]:
        a = np.ascontiguousarray(acts[done_steps, :nenv].astype(np.float64))
        L.orc_ns2d_step_batch(C.byref(e.cfg), nenv, O.dp(st.reshape(-1)), O.dp(a.reshape(-1)), a.shape[1],
                              O.dp(obs.reshape(-1)), O.dp(rwd), sw.ctypes.data_as(C.POINTER(C.c_int64)), cores)
        done_steps += 1
        tot_sw += int(sw.sum())
        if time.perf_counter() - t0 > seconds:
            break
    dt = time.perf_counter() - t0
    port = {"value": nenv * done_steps / dt, "unit": "env steps/s", "cores": cores, "kind": "port",
            "sample": "replicas 0..%d x the first %d timed action steps of the GPU run, from the GPU's state after "
                      "warm-up (float64 C oracle, OpenMP over envs, %.1f s, %.1f Jacobi sweeps per timestep)"
                      % (nenv - 1, done_steps, dt, tot_sw / max(1, nenv * done_steps * e.cfg.ndt_act))}
    # vectorised NumPy, one env, one core: a fraction of one action step (whole timesteps) bounded by `seconds`
    env = NP.Rayleigh(init_fields=np.swapaxes(state0[0], -1, -2), **cfg_kw)
    t0, ndt, nsw = time.perf_counter(), 0, 0
    a = NP.condition_actions(acts[0, 0].astype(np.float64), env.C)
    while ndt < env.ndt_act:
        nsw += env.timestep(a)
        ndt += 1
        if time.perf_counter() - t0 > seconds:
            break
    dt = time.perf_counter() - t0
    port["numpy"] = {"value": (ndt / env.ndt_act) / dt, "unit": "env steps/s", "cores": 1, "kind": "port",
                     "sample": "replica 0, first %d of %d timesteps of the first timed action step (vectorised "
                               "NumPy restatement, 1 core, %.1f s, %.1f sweeps per timestep)"
                               % (ndt, env.ndt_act, dt, nsw / max(1, ndt))}
    return port


# ------------------------------------------------------------------------------------------------
# stub env: CPU tensors, same surface (launcher / sharding test on machines without a GPU)
# ------------------------------------------------------------------------------------------------
def make_stub_env(batch):
    """CPU stand-in env: beacon_amd.vec.VecEnv's own host logic (packed, double-buffered outputs, masks) with the three
    calls that reach the HIP library replaced by tensor operations on CPU tensors."""
    import numpy as np
    import torch
    from beacon_amd.vec import VecEnv

    class StubEnv(VecEnv):
        action_is_int = False

        def __init__(self, batch):
            self.batch, self.obs_dim, self.n_actions, self.n_sgts = batch, 8, 10, 10
            self.tdtype, self.device, self.h = torch.float32, torch.device("cpu"), None
            self.nx, self.ny, self.ndt_act = 4, 4, 2
            self._alloc_outputs()
            self.sweeps = torch.ones((batch, 2), dtype=torch.int32)

        kernel_name = "stub"

        def _apply_mask(self, mask):
            self._mask = mask

        def _reset(self):
            self.obs.zero_()

        def _step(self, a, noise=None):
            self.obs[:] = a[:, :8] * 2
            self.rwd[:] = a.sum(1)

        def check_status(self):
            return self.status

        def get_counters(self):
            return np.ones((self.batch, 4), dtype=np.uint64)

        def close(self):
            pass

    return StubEnv(batch)


# ------------------------------------------------------------------------------------------------
def secondary_lines(dev, head_acts, head_warm, head_init, head_LH, quick_steps=4):
    """Other configurations of BASELINE.json on this GPU, one short measurement each (HIP events around
    the launch, inputs resident): value = env steps/s, roofline per SURVEY 8d accounting."""
    import numpy as np
    import torch
    from beacon_amd import vec as V
    from beacon_amd.envs import packaged_init
    out = []

    def timed(env, step, n, warm=2):
        for _ in range(warm):
            step()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for s, e in ev:
            s.record()
            step()
            e.record()
        torch.cuda.synchronize()
        env.check_status()
        return float(np.mean([s.elapsed_time(e) for s, e in ev]))

    def timed_loop(env, step, n, warm=5):
        """n eager step() calls back to back between ONE pair of HIP events: what a trainer's loop pays per call.  (An event
        pair around EVERY call, as `timed` records for the millisecond-long 2D steps, puts two extra commands between two
        30 us kernels: burgers 36.9 instead of 32.1 us; the host needs 7.8 us to issue a call -- scripts/host_cost.py.)"""
        for _ in range(warm):
            step()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            step()
        e.record()
        torch.cuda.synchronize()
        env.check_status()
        return s.elapsed_time(e) / n

    def timed_graph(env, actions, noise=None, n=16, reps=5):
        """The same step recorded n times into one HIP graph (VecEnv.capture): ms per step at replay -- what a trainer
        that records its loop sees instead of the per-launch host work of Python."""
        g = env.capture(actions.unsqueeze(0).expand(n, *actions.shape).contiguous(), noise, n_steps=n, keep_steps=False)
        g.replay()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            g.replay()
        e.record()
        torch.cuda.synchronize()
        env.check_status()
        return s.elapsed_time(e) / (reps * n)

    def line(name, env, ms, alg_bytes, extra=None):
        # the state of these kernels is on chip for the whole action step: what binds them is vector-instruction issue.  The
        # instruction count per dispatch is the committed profile's (SQ_INSTS_VALU of profiles/*_summary.json for this kernel);
        # the SURVEY 8d bytes over the time stay beside it as `hbm_effective` (above the HBM peak where nothing goes to HBM)
        dt = "f64" if env.tdtype == torch.float64 else "f32"
        hbm_eff = {"achieved": alg_bytes / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": alg_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "note": "SURVEY 8d algorithmic bytes / time: not the binding resource"}
        prof = committed_profile(env.kernel_name, "SQ_INSTS_VALU", dt)
        if prof is not None:
            ach = prof["value"] / (ms * 1e-3)
            roof = {"bound": "valu_issue", "achieved": ach, "peak": VALU_ISSUE_PEAK, "unit": "wave64 VALU instructions/s",
                    "frac": ach / VALU_ISSUE_PEAK, "instructions": "SQ_INSTS_VALU per dispatch of " + prof["source"],
                    "instructions_measured_in_this_run": False, "hbm_effective": hbm_eff}
        else:
            roof = {"bound": "valu_issue", "achieved": None, "peak": VALU_ISSUE_PEAK, "unit": "wave64 VALU instructions/s", "frac": None,
                    "note": "no committed profile of this kernel", "hbm_effective": hbm_eff}
        d = {"workload": name, "value": env.batch / (ms * 1e-3), "unit": "env steps/s", "ms_per_launch": ms,
             "kernel": env.kernel_name, "dtype": dt, "roofline": roof}
        d.update(extra or {})
        out.append(d)

    rng = np.random.default_rng(7)
    z = {"fields": head_init}
    W = head_warm

    def headline_variant(name, dtype, opts, note, nrep=None, extra=None):
        """The headline workload itself (same initial state, same warm-up and timed action stream) with other solver
        options / in float64 / on its first `nrep` replicas only: ms per step over the same `steps` steps."""
        nrep = head_acts.shape[1] if nrep is None else nrep
        env = V.VecRayleigh(nrep, dev, dtype, z["fields"], L=head_LH[0], H=head_LH[1])
        for k_, v_ in opts.items():
            env.set_option(k_, v_)
        env.reset()
        acts = torch.as_tensor(head_acts[:, :nrep], dtype=env.tdtype, device=dev)
        k = [0]

        sw = []

        def st():
            env.step(acts[k[0]]); k[0] += 1
            if k[0] > W:
                sw.append(env.sweeps.clone())           # every TIMED step's sweep counts, not only the last one's
        n = acts.shape[0] - W
        ms = timed(env, st, n, warm=W)
        c = env.get_counters()
        esz = 4 if dtype == "f32" else 8
        alg = float(np.mean([algorithmic_bytes(env.nx, env.ny, x.cpu().numpy(), esz) for x in sw]))
        d = {"mean_jacobi_sweeps_per_timestep": float(np.mean([x.float().mean().item() for x in sw])), "steps": n,
             "options": opts, "note": note, "replicas": nrep,
             "unverified_landings_last_step": int(c[:, 2].sum()), "repeated_solves_last_step": int(c[:, 3].sum())}
        d.update(extra or {})
        line(name, env, ms, alg, d)
        env.close()

    ai = torch.as_tensor(rng.integers(0, 4, (8, 512)), dtype=torch.int32, device=dev)
    a1 = torch.as_tensor(rng.uniform(-1, 1, (1024,)), dtype=torch.float32, device=dev)

    def leg_stop_rules():
        # what each way of knowing the stop sweep costs on the headline workload (the default, plan 3, is the headline itself)
        headline_variant("headline workload, float32, conv_plan=1 (lower-bound plan)", "f32", {"conv_plan": 1},
                         "residual evaluated on every sweep that the log-convex lower bound of the norm cannot exclude")
        headline_variant("headline workload, float32, spec_start=0", "f32", {"spec_start": 0},
                         "default plan (3: extrapolated, every landing verified) without the speculative opening")
        headline_variant("headline workload, float32, conv_plan=2 spec_start=7 (UNVERIFIED landings: the default of rounds 3-4)", "f32",
                         {"conv_plan": 2, "spec_start": 7},
                         "the extrapolating plan trusting a failing landing -- not proven (the reference's norm can grow by 1.030 "
                         "between sweeps): what the verification of plan 3 costs")

    def leg_strong_scaling_shards():
        # what ONE rank of a 2 / 4 / 8-GPU strong-scaling run of the headline executes: the first 512 / N replicas of the same
        # action stream on this GPU (replicas are independent: a rank's shard does not know the others exist).  Below one
        # replica per CU a step cannot get shorter -- a replica is a serial chain on one CU -- so these lines are what the
        # top-level `scaling_prediction` is computed from
        full = head_acts.shape[1]
        for n_gpus in (2, 4, 8):
            if full % n_gpus == 0 and full // n_gpus >= 1:
                headline_variant("headline workload, the %d replicas of one rank of a %d-GPU strong-scaling run" % (full // n_gpus, n_gpus),
                                 "f32", {}, "replicas 0..%d of the headline's action stream, same steps" % (full // n_gpus - 1),
                                 nrep=full // n_gpus, extra={"strong_scaling_shard_of": n_gpus})

    def leg_headline_f64():
        # rayleigh 128x64 float64 (the reference's arithmetic), B=512, the headline's own steps
        headline_variant("headline workload (rayleigh-v0 128x64 B=512) in float64", "f64", {},
                         "the reference's arithmetic; default plan 3 (proven landings): sweep counts equal to the oracle's")
        headline_variant("headline workload in float64, conv_plan=1", "f64", {"conv_plan": 1},
                         "the lower-bound plan (float64 default until round 5)")

    def leg_mixing_f32():
        # mixing 100x100 B=512 (configs[4])
        env = V.VecMixing(512, dev, "f32")
        env.reset()
        k = [0]

        def st():
            env.step(ai[k[0] % 8]); k[0] += 1
        ms = timed(env, st, max(quick_steps, 8), warm=2)
        line("mixing-v0 100x100 B=512 float32 (configs[4])", env, ms,
             algorithmic_bytes(100, 100, env.sweeps.cpu().numpy(), 4),
             {"mean_jacobi_sweeps_per_timestep": float(env.sweeps.float().mean())})
        env.close()

    def leg_mixing_f64():
        # mixing 100x100 B=512 float64 (the reference's arithmetic)
        env = V.VecMixing(512, dev, "f64")
        env.reset()
        k = [0]

        def st():
            env.step(ai[k[0] % 8]); k[0] += 1
        ms = timed(env, st, 5, warm=1)
        line("mixing-v0 100x100 B=512 float64", env, ms, algorithmic_bytes(100, 100, env.sweeps.cpu().numpy(), 8),
             {"mean_jacobi_sweeps_per_timestep": float(env.sweeps.float().mean())})
        env.close()

    def leg_tall_grid():
        # a grid above ny = 128 (the reference takes any L, H: mixing.py:20-28): mixing(L=1, H=2) = 100x200, B=256 -- ns2d_fast4
        # (Poisson solve in registers, the other phases from HBM/L2; the generic kernel needs 574 ms for this step)
        env = V.VecMixing(256, dev, "f32", L=1.0, H=2.0)
        env.reset()
        k = [0]
        ai2 = ai[:, :256].contiguous()

        def st2():
            env.step(ai2[k[0] % 8]); k[0] += 1
        ms = timed(env, st2, 5, warm=2)
        line("mixing-v0 L=1 H=2 (100x200) B=256 float32", env, ms, algorithmic_bytes(100, 200, env.sweeps.cpu().numpy(), 4),
             {"mean_jacobi_sweeps_per_timestep": float(env.sweeps.float().mean())})
        env.close()

    def leg_burgers():
        # burgers N=512 B=1024 (configs[1]): 12 B per cell per timestep
        env = V.VecBurgers(1024, dev, "f32", nx=512)
        env.reset()
        ms = timed_loop(env, lambda: env.step(a1), 200)
        line("burgers-v0 N=512 B=1024 float32 (configs[1])", env, ms, 12.0 * 512 * env.ndt_act * 1024,
             {"ms_per_step_in_hip_graph": timed_graph(env, a1), "inlet_noise": "drawn inside the step kernel (bcn_set_noise)"})
        env.close()

    def leg_shkadov():
        # shkadov N=4096 10 jets B=1024 (configs[2]): 32 B per cell per timestep
        env = V.VecShkadov(1024, dev, "f32", None, L0=699.2, n_jets=10)
        env.reset()
        # from a DEVELOPED film (shkadov/init.py:13-27: 4000 uncontrolled action steps under inlet noise), not the flat one:
        # the limiter branches and the noise amplification are what a trainer's steps run through
        env.warmup(env.n_warmup_ref, torch.zeros((1024, 10), dtype=env.tdtype, device=dev))
        amp = float((env.get_state()[:, 0] - 1.0).abs().amax(dim=1).mean())
        a10 = torch.as_tensor(rng.uniform(-1, 1, (64, 1024, 10)), dtype=env.tdtype, device=dev)
        k = [0]

        def st_shk():
            env.step(a10[k[0] % 64]); k[0] += 1
        ms = timed_loop(env, st_shk, 50)
        line("shkadov-v0 N=4096 10 jets B=1024 float32 (configs[2])", env, ms, 32.0 * env.nx * env.ndt_act * 1024,
             {"ms_per_step_in_hip_graph": timed_graph(env, a10[0]), "inlet_noise": "drawn inside the step kernel (bcn_set_noise)",
              "initial_state": "developed on the device: %d uncontrolled action steps under inlet noise; mean wave amplitude max|h-1| = %.3f"
                               % (env.n_warmup_ref, amp),
              "actions": "uniform(-1, 1) per replica, jet and step", "blown_up_replicas_after_timing": int((env.status & 2).bool().sum())})
        env.close()

    def leg_sloshing():
        # sloshing (reference default grid) B=1024: 32 B per cell per timestep
        env = V.VecSloshing(1024, dev, "f32", packaged_init("sloshing"))
        env.reset()
        ms = timed_loop(env, lambda: env.step(a1), 200)
        line("sloshing-v0 N=200 B=1024 float32", env, ms, 32.0 * (env.nx + 2) * env.ndt_act * 1024,
             {"ms_per_step_in_hip_graph": timed_graph(env, a1)})
        env.close()

    def leg_lorenz():
        out.append(lorenz_line())

    for leg in (leg_stop_rules, leg_strong_scaling_shards, leg_headline_f64, leg_mixing_f32, leg_mixing_f64, leg_tall_grid, leg_burgers, leg_shkadov,
                leg_sloshing, leg_lorenz):
        try:      # one failing configuration must not cost the others (nor the headline: main() guards this whole function too)
            leg()
        except Exception as e:      # noqa: BLE001 -- reported in the line
            import traceback
            sys.stderr.write("bench.py: secondary leg %s failed:\n%s" % (leg.__name__, traceback.format_exc()))
            out.append({"workload": leg.__name__[4:], "error": "%s: %s" % (type(e).__name__, e)})
    return out


def lorenz_line(episodes=4):
    """BASELINE configs[0]: lorenz-v0, single env, CPU (RK4 ODE -- plumbing, no GPU): whole 500-step episodes of the
    host-only mirror (beacon_amd/lorenz.py, bit-exact against the reference's episodes: tests/test_host.py)."""
    import numpy as np
    from beacon_amd.lorenz import lorenz
    env = lorenz()
    rng = np.random.default_rng(0)
    n, t0 = 0, time.perf_counter()
    for _ in range(episodes):
        env.reset()
        done = False
        while not done:
            _, _, done, _, _ = env.step(np.int64(rng.integers(0, 3)))
            n += 1
    dt = time.perf_counter() - t0
    return {"workload": "lorenz-v0 single env (configs[0]: CPU plumbing, no GPU kernel)", "value": n / dt, "unit": "env steps/s",
            "ms_per_launch": 1e3 * dt / n, "kernel": "host (beacon_amd/lorenz.py)", "dtype": "f64", "steps": n,
            "note": "%d episodes of %d action steps on one host core, random Discrete(3) actions" % (episodes, n // episodes)}


def main():
    args = parse_args()
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        # not launched by torch.distributed.run: start one fresh process per GPU BEFORE anything here touches
        # the GPU (this process never does), and leave rank 0 to print the line
        sys.exit(launch_children(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(world_env or "1")
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    import numpy as np
    import torch
    import torch.distributed as dist
    from beacon_amd.dist import ShardedVecEnv

    if args.stub:
        dev = "cpu"
    else:
        if args.share_gpu:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        dev = "cuda:%d" % local_rank
    fault = os.environ.get("BENCH_FAULT_RANK")          # test hook: this rank dies before the rendezvous
    if fault is not None and int(fault) == rank:
        sys.stderr.write("bench.py: injected fault on rank %d\n" % rank)
        os._exit(3)
    distc = world > 1 or args.force_dist         # the collectives run (always when there is more than one rank)
    if distc:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(free_port()))   # two rehearsals on one host must not collide
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        tmo = datetime.timedelta(seconds=args.dist_timeout)
        if args.stub or args.backend != "nccl":
            dist.init_process_group(args.backend, timeout=tmo)
        else:
            dist.init_process_group(args.backend, timeout=tmo, device_id=torch.device(dev))

    if args.scaling == "strong":
        if args.batch % world:
            raise SystemExit("--scaling strong: global batch %d is not divisible by %d GPUs" % (args.batch, world))
        B = args.batch // world
    else:
        B = args.batch
    Bg = B * world
    K, W = args.steps, args.warmup
    L, H = args.L, args.H
    init, init_src = None, "stub"
    if not args.stub:
        from beacon_amd import vec as V
        if args.gen_init or (L, H) != (2.56, 1.28):
            t_gen = time.perf_counter()
            genv = V.VecRayleigh(1, dev, "f64", None, L=L, H=H, n_sgts=1)
            init = genv.develop()                       # every rank develops the same (seeded) state
            init_src = ("developed on the GPU: %d zero-action steps from the seeded conduction state, float64, %s, %.1f s"
                        % (genv.n_warmup, genv.kernel_name, time.perf_counter() - t_gen))
            genv.close()
        else:
            init = np.load(os.path.join(ROOT, "tests", "golden", "rayleigh_128x64_init.npz"))["fields"]
            init_src = "tests/golden/rayleigh_128x64_init.npz (float64 C oracle warm-up)"

    def make_env(nrep):
        if args.stub:
            return make_stub_env(nrep)
        e = V.VecRayleigh(nrep, dev, args.dtype, init, L=L, H=H)
        if args.variant >= 0:
            e.set_variant(args.variant)
        if args.sched >= 0:
            e.set_sched(args.sched)
        return e

    def sync():
        if not args.stub:
            torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        if not args.stub:
            torch.cuda.synchronize()

    def timed_steps(env, senv, acts, W, K, collective, after_warmup=None):
        """W untimed warm-up steps, then EXACTLY K timed ones between barrier + synchronize on both sides.  Every step is
        ONE HIP launch on torch's current stream, bracketed by HIP events on that stream; with `collective` the
        trainer-facing gather of every step's packed outputs is inside the timed region -- on a side stream behind an
        event (ShardedVecEnv.step_async: the step after next waits for it before it overwrites the buffer), or with
        --no-overlap as a blocking collective between two steps."""
        for k in range(W):
            senv.step(acts[k], scattered=True)
            _ = env.sweeps.clone()                      # (the timed loop's own small allocation, so that the caching allocator has it)
        sync()
        if after_warmup is not None:
            after_warmup()
            sync()
        use_ev = not args.stub
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)] if use_ev else []
        sweeps_all = []
        pend = None
        # rank 0 takes delivery on its own stream, as a trainer's copy stream would: the step stream never waits for a gather
        # younger than three steps (ShardedVecEnv.NBUF output buffers)
        consumer = torch.cuda.Stream(device=dev) if (use_ev and collective and senv.overlap) else None

        def deliver(p_):
            if consumer is None:
                p_.wait()
            else:
                with torch.cuda.stream(consumer):
                    p_.wait()
        import gc
        gc.collect()
        gc.disable()                                    # no collector pause inside the K timed steps (20 ms in one of this round's runs)
        t0 = time.perf_counter()
        for k in range(K):
            if use_ev:
                ev[k][0].record()
            p = None
            if collective and senv.overlap:
                p = senv.step_async(acts[W + k], scattered=True)   # the one HIP launch of this step + its gather (side stream)
            else:
                env.step(acts[W + k])                              # the one HIP launch of this step (torch's current stream)
            if use_ev:
                ev[k][1].record()
            sweeps_all.append(env.sweeps.clone())       # tiny device copy, for the roofline accounting
            if collective and not senv.overlap:
                senv._gather()                          # --no-overlap: blocking collective between two steps
            if pend is not None:
                deliver(pend)                           # rank 0 takes delivery of step k - 1 while step k runs
            pend = p
        if pend is not None:
            deliver(pend)
        sync()
        elapsed = time.perf_counter() - t0
        gc.enable()
        env.check_status()
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        if dist.is_initialized():
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        kern_ms = [s_.elapsed_time(e_) for s_, e_ in ev] if use_ev else [1e3 * elapsed / K] * K
        timed_steps.own_elapsed = elapsed            # this rank's own wall time of the timed region (the return value is the max)
        return float(tt.item()), kern_ms, sweeps_all

    env = make_env(B)
    senv = ShardedVecEnv(env, always_collective=args.force_dist, overlap=not args.no_overlap)
    # one global action stream, every rank takes the slice of its replicas (weak: a longer stream)
    acts_g = np.random.default_rng(1234).uniform(-1.0, 1.0, (W + K, Bg, env.n_sgts))
    if args.zero_actions:
        acts_g[:] = 0.0
    acts_np = acts_g[:, senv.lo:senv.hi]
    acts = torch.as_tensor(acts_np, dtype=env.tdtype, device=dev)
    senv.reset()
    saved = {}

    def keep_state():          # the CPU legs start from the GPU's state after the warm-up steps
        if rank == 0 and not args.stub and not args.no_cpu and world == 1:
            saved["state"] = env.get_state()[:min(os.cpu_count() or 1, B)].cpu().numpy().astype(np.float64)

    elapsed, kern_ms, sweeps_all = timed_steps(env, senv, acts, W, K, distc, keep_state)
    own_elapsed = timed_steps.own_elapsed
    state_after_warmup = saved.get("state")
    cyc = env.get_counters().astype(np.float64)     # of the last step

    # ---- who ran what: one record per rank, so that an N > 1 line proves its own topology ----
    def rank_record():
        rec = {"rank": rank, "local_rank": local_rank, "pid": os.getpid(), "host": socket.gethostname(),
               "replicas": B, "global_replica_range": [senv.lo, senv.hi],
               "ms_per_launch": float(np.mean(kern_ms)), "ms_per_step_own_clock": 1e3 * own_elapsed / K,
               "env_steps_per_s_own_clock": B * K / own_elapsed,
               "jacobi_sweeps_timed": int(sum(int(x.sum().item()) for x in sweeps_all)),
               "kernel": env.kernel_name}
        if args.stub:
            rec["device"] = "cpu (stub env)"
        else:
            pr = torch.cuda.get_device_properties(local_rank)
            rec.update({"device_index": local_rank, "device_name": pr.name,
                        "gcn_arch": getattr(pr, "gcnArchName", None), "compute_units": pr.multi_processor_count,
                        "uuid": str(getattr(pr, "uuid", "")) or None,
                        "pci": "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0),
                                                    getattr(pr, "pci_device_id", 0)) if hasattr(pr, "pci_bus_id") else None,
                        "visible_devices": os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")})
        return rec
    ranks = [rank_record()]
    if dist.is_initialized():
        gathered = [None] * world
        dist.all_gather_object(gathered, ranks[0])
        ranks = gathered
    topo = {"world_size": world, "backend": (args.backend if dist.is_initialized() else None),
            "launcher": "torch.distributed.run" if os.environ.get("TORCHELASTIC_RUN_ID") else
                        ("bench.py self-launch" if world_env is not None else "single process")}
    if dist.is_initialized() and args.backend == "nccl" and not args.stub:
        try:
            topo["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception as e:      # noqa: BLE001
            topo["rccl_version"] = "unavailable: %s" % e
    if not args.stub:
        ids = [(r.get("host"), r.get("uuid") or r.get("pci") or r.get("device_index")) for r in ranks]
        topo["distinct_devices"] = len(set(ids))
        if world > 1 and not args.share_gpu and len(set(ids)) != world:
            raise SystemExit("bench.py: %d ranks but only %d distinct devices (%s): one process per GPU is the contract"
                             % (world, len(set(ids)), ids))

    def dist_path_lines():
        """The headline workload once more through the N > 1 code path on this one GPU: a process group of ONE rank over the
        backend (nccl = RCCL), barriers and the max-reduction around the timed region, and every step's packed outputs
        gathered through the backend inside it -- what the driver's 2 / 4 / 8-GPU runs execute per rank, so that the N = 1
        record shows what that path costs."""
        import datetime
        lines = []
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(args.backend, timeout=datetime.timedelta(seconds=args.dist_timeout),
                                device_id=torch.device(dev))
        try:
            for overlap in (True, False):
                e2 = make_env(B)
                s2 = ShardedVecEnv(e2, always_collective=True, overlap=overlap)
                s2.reset()
                t2, k2, sw2 = timed_steps(e2, s2, acts, W, K, True)
                lines.append({"workload": "headline workload through the N > 1 path: one rank over %s, packed gather %s"
                                          % (args.backend, "on a side stream beside the next step (double-buffered outputs)"
                                             if overlap else "as a blocking collective between two steps (--no-overlap)"),
                              "value": B * K / t2, "unit": "env steps/s", "ms_per_step": 1e3 * t2 / K,
                              "ms_per_launch": float(np.mean(k2)), "kernel": e2.kernel_name, "dtype": args.dtype,
                              "steps": K, "warmup": W, "vs_plain_ms_per_step": 1e3 * t2 / K / (1e3 * elapsed / K),
                              "mean_jacobi_sweeps_per_timestep": float(np.mean([x.float().mean().item() for x in sw2]))})
                e2.close()
        finally:
            # (no barrier on the way out of an exception: on a broken group it would mask the error that broke it)
            try:
                if sys.exc_info()[0] is None:
                    dist.barrier()
                dist.destroy_process_group()
            except Exception as e2:
                sys.stderr.write("bench.py: tearing down the one-rank process group failed: %s\n" % e2)
        return lines

    # N > 1, weak scaling: one more short timed loop with the GLOBAL batch of --batch replicas sharded over the ranks,
    # so that one invocation carries both readings of "batch=512 at 1/2/4/8 GPUs" (same barriers, max over ranks)
    strong = None
    if distc and args.scaling == "weak" and not args.no_strong and args.batch % world == 0:
        Bs, Ks, Ws = args.batch // world, min(K, 5), 1
        env_s = make_env(Bs)
        senv_s = ShardedVecEnv(env_s, always_collective=args.force_dist, overlap=not args.no_overlap)
        a_s = np.random.default_rng(1234).uniform(-1.0, 1.0, (Ws + Ks, args.batch, env_s.n_sgts))[:, senv_s.lo:senv_s.hi]
        a_s = torch.as_tensor(a_s, dtype=env_s.tdtype, device=dev)
        senv_s.reset()
        t_strong, _, _ = timed_steps(env_s, senv_s, a_s, Ws, Ks, True)
        ts = torch.tensor([t_strong], dtype=torch.float64, device=dev)
        strong = {"scaling": "strong", "global_batch": args.batch, "replicas_per_gpu": Bs, "steps": Ks, "warmup": Ws,
                  "ms_per_step": 1e3 * float(ts.item()) / Ks, "value": args.batch * Ks / float(ts.item()),
                  "unit": "env steps/s"}
        env_s.close()

    if rank == 0:
        esz = 4 if args.dtype == "f32" else 8
        sw_np = [s.cpu().numpy() for s in sweeps_all]
        alg = [algorithmic_bytes(env.nx, env.ny, s, esz) for s in sw_np]
        mean_sw = float(np.mean([s.mean() for s in sw_np]))
        launch_s = sum(kern_ms) / len(kern_ms) * 1e-3
        achieved = (sum(alg) / len(alg)) / launch_s / 1e9
        kname = env.kernel_name
        cells = env.nx * env.ny
        sweeps_per_launch = float(np.mean([s.sum() for s in sw_np]))
        hbm_eff = {"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                   "algorithmic_bytes_per_launch": sum(alg) / len(alg),
                   "note": "SURVEY 8d accounting: algorithmic bytes (20 values per cell and timestep + 3 per cell and Jacobi sweep) / "
                           "launch time (HIP events on the launch stream).  A replica's state stays in registers / LDS inside the "
                           "launch, so this EFFECTIVE figure exceeds the HBM peak: HBM is not the binding resource (`traffic`, "
                           "`hbm_measured`); VALU issue is, and the top-level `frac` is that fraction"}
        # binding resource: VALU issue.  Instructions per launch from the committed SQ_INSTS_VALU pass of the same command
        # (PMC counters need rocprofv3), scaled by this run's sweeps per launch over the profile's, over THIS run's launch time
        # (HIP events); without a committed profile of this kernel: the minimum-instruction count (7 VALU per cell and sweep).
        grid_tag = "%d, %d," % (env.nx, env.ny)
        vi = committed_profile(kname, "SQ_INSTS_VALU", args.dtype, grid_tag)
        min_ach = MIN_VALU_PER_CELL_SWEEP * cells / 64.0 * sweeps_per_launch / launch_s
        roof = {"bound": "valu_issue", "achieved": min_ach, "peak": VALU_ISSUE_PEAK, "unit": "wave64 VALU instructions/s",
                "frac": min_ach / VALU_ISSUE_PEAK, "traffic": None, "kernel": kname, "avg_launch_ms": launch_s * 1e3,
                "peak_note": "256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction",
                "instructions": "lower bound: %d VALU instructions per cell and sweep (no committed SQ_INSTS_VALU profile of this kernel)"
                                % MIN_VALU_PER_CELL_SWEEP,
                "min_instr": {"achieved": min_ach, "frac": min_ach / VALU_ISSUE_PEAK,
                              "note": "%d VALU instructions per cell and Jacobi sweep, nothing else counted" % MIN_VALU_PER_CELL_SWEEP}}
        if vi and vi.get("sweeps_per_dispatch"):
            insts = vi["value"] * sweeps_per_launch / vi["sweeps_per_dispatch"]
            roof.update({"achieved": insts / launch_s, "frac": insts / launch_s / VALU_ISSUE_PEAK,
                         "instructions": "SQ_INSTS_VALU of profiles/%s (%.4g per dispatch at %.4g sweeps), scaled to this run's %.4g "
                                         "sweeps per launch" % (vi["source"], vi["value"], vi["sweeps_per_dispatch"], sweeps_per_launch),
                         "instructions_measured_in_this_run": False, "launch_time_measured_in_this_run": True,
                         "source": "profiles/" + vi["source"]})
            sys.stderr.write("bench.py: VALU / HBM counters rescaled from profiles/%s (%.4g sweeps per dispatch there, %.4g per "
                             "launch in this run)\n" % (vi["source"], vi["sweeps_per_dispatch"], sweeps_per_launch))
        # Cross-check of that figure: instructions from a MODEL instead of a rescaled total -- static x sweeps +
        # per_timestep x timesteps, where `static` is the compiler's own VALU count of the plain double sweep (scripts/valu_model.py
        # -> profiles/r06_valu_model.json: 178 per wave and pair = 712 per replica-sweep; the PMC fit between the default and the
        # zero-action profile gives 711.3) and `per_timestep` is what the zero-action profile (one sweep per solve) leaves
        # (profiles/r06_valu_fit.json).  This run's own sweep counts go in; the two figures must agree to a few per cent.
        try:
            fit = json.load(open(os.path.join(ROOT, "profiles", "r06_valu_fit.json")))
            if fit.get("kernel", "").startswith(kname + "<") and args.dtype == "f32" and grid_tag in fit["kernel"]:
                ts_per_launch = float(B * env.ndt_act)
                im = fit["valu_per_replica_sweep_static"] * sweeps_per_launch + fit["valu_per_replica_timestep_outside_plain_sweeps"] * ts_per_launch
                roof["instructions_model"] = {
                    "instructions": im, "achieved": im / launch_s, "frac": im / launch_s / VALU_ISSUE_PEAK,
                    "valu_per_replica_sweep_static": fit["valu_per_replica_sweep_static"],
                    "valu_per_replica_sweep_fitted": fit["valu_per_replica_sweep_fitted"],
                    "valu_per_replica_timestep": fit["valu_per_replica_timestep_outside_plain_sweeps"],
                    "model_over_profile_scaled": (im / launch_s) / roof["achieved"] if roof.get("achieved") else None,
                    "source": "profiles/r06_valu_model.json (hipcc -S of the kernel), profiles/r06_valu_fit.json (two PMC profiles)"}
        except (OSError, ValueError, KeyError) as e:
            roof["instructions_model"] = {"error": "%s: %s" % (type(e).__name__, e)}
        roof["hbm_effective"] = hbm_eff
        tr = committed_profile(kname, "hbm_bytes_per_dispatch", args.dtype, grid_tag)
        if tr:
            roof["traffic"] = tr["value"]
            roof["hbm_measured"] = {"bytes_per_launch": tr["value"], "GB/s": tr["value"] / launch_s / 1e9,
                                    "frac_of_peak": tr["value"] / launch_s / 1e9 / HBM_PEAK_GBS,
                                    "source": "profiles/" + tr["source"],
                                    "measured_in_this_run": False,
                                    "note": "PMC counters need rocprofv3: bytes per launch are those of the committed profile "
                                            "(same command, same workload), divided by THIS run's launch time"}
            roof["traffic_source"] = "profiles/" + tr["source"] + " (committed rocprofv3 --pmc pass; not re-measured by this run)"
        # Poisson phase alone (what north_star's ">= 50 % on the Poisson sweep" refers to): share of the replicas'
        # shader cycles spent inside the Jacobi loop (in-kernel s_memtime, last step) x launch time
        if cyc[:, 1].sum() > 0:
            share = float(cyc[:, 0].sum() / cyc[:, 1].sum())
            pb = 12.0 / 4 * esz * cells * float(sw_np[-1].sum())
            roof["poisson"] = {"time_share": share, "hbm_effective_GBs": pb / (share * kern_ms[-1] * 1e-3) / 1e9,
                               "hbm_effective_frac": pb / (share * kern_ms[-1] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "cycles_per_replica_sweep": float(cyc[:, 0].sum() / max(1.0, float(sw_np[-1].sum())))}
        metric = "aggregate env steps/sec, rayleigh-v0 batch=%d%s %dx%d" % (args.batch, "/GPU" if args.scaling == "weak" else " global", env.nx, env.ny)
        out = {
            "metric": metric,
            "value": Bg * K / elapsed, "unit": "env steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": args.dtype, "data": "stub" if args.stub else "synthetic",
            "config": {"workload": "rayleigh-v0 (BASELINE configs[3]): L=%g H=%g -> %dx%d MAC grid, "
                                   "%d replicas per GPU, 200 timesteps per step, Jacobi to tol=1e-8" % (L, H, env.nx, env.ny, B),
                       "initial_state": init_src,
                       "global_batch": Bg, "grid": [env.nx, env.ny], "ndt_act": env.ndt_act,
                       "mean_jacobi_sweeps_per_timestep": mean_sw, "parallelism": "replica-sharded x%d" % world + (" (rehearsal: all ranks on cuda:0)" if args.share_gpu else ""),
                       "kernel": kname},
            "roofline": roof,
            "ranks": ranks,
            "topology": topo,
        }
        out["solver"] = {"conv_plan": "3 (default): the decay of the reference's residual norm extrapolated, and EVERY landing behind "
                                      "skipped sweeps verified -- the norm can grow by at most 1.030 between sweeps "
                                      "(scripts/weighted_norm_bound.py), so a landing above 1.035 tol proves that no skipped sweep passed; "
                                      "a landing below it is repeated under the lower-bound plan: every stop sweep is the reference's.  "
                                      "Within the slow modes of the Jacobi matrix the growth is far smaller (beacon_amd/stoprule.py), which "
                                      "lowers the landing threshold to min(1.035, C_L (1 + 2 e)^2) tol, e from |d_1|^2 and the last evaluated sweep",
                         "slow_mode_bound": [list(x) for x in env.slow_mode_bound()] if hasattr(env, "lib") else [],   # (the CPU stub has no library)
                         "unverified_landings_last_step": int(cyc[:, 2].sum()), "repeated_solves_last_step": int(cyc[:, 3].sum()),
                         "solves_last_step": int(B * env.ndt_act)}
        if strong is not None:
            out["strong"] = strong
        # (the N > 1 rehearsal BEFORE the CPU legs: the C oracle's OpenMP workers keep spinning on every host core after
        #  their parallel regions, and the blocking gather's host side then measured 2 ms per step late)
        # Nothing below may cost the headline that is already measured: a failure of a secondary leg (RCCL rendezvous of the
        # one-rank rehearsal, a secondary configuration, the CPU legs) becomes an {"error": ...} entry and the line is printed.
        sec = []

        def guarded(what, fn):
            try:
                return fn()
            except Exception as e:      # noqa: BLE001 -- reported in the line
                import traceback
                sys.stderr.write("bench.py: %s failed:\n%s" % (what, traceback.format_exc()))
                sec.append({"workload": what, "error": "%s: %s" % (type(e).__name__, e)})
                return None
        if world == 1 and not args.no_secondary and not args.stub and not distc:
            sec += guarded("headline workload through the N > 1 path (one rank over %s)" % args.backend, dist_path_lines) or []
        if world == 1 and not args.no_cpu and not args.stub:
            cpu = guarded("cpu_baseline", lambda: cpu_legs(state_after_warmup, acts_np[W:], dict(L=L, H=H)))
            if cpu is not None:
                out["cpu_baseline"] = cpu
        if world == 1 and not args.no_secondary and not args.stub:
            env.close()
            sec += guarded("secondary configurations", lambda: secondary_lines(dev, acts_np, W, init, (L, H))) or []
        if sec:
            out["secondary"] = sec
        # Predicted 1 / 2 / 4 / 8-GPU values from what THIS GPU measured (no hardware curve is claimed): replicas are independent,
        # a rank of an N-GPU run executes exactly one of the lines above plus the packed gather, whose cost per step is the
        # difference between the one-rank-over-RCCL line and the plain loop.  A later SCALE run can be checked against this.
        if world == 1 and sec:
            t1 = 1e3 * elapsed / K
            gath = [d for d in sec if "through the N > 1 path" in d.get("workload", "") and "side stream" in d.get("workload", "")
                    and "ms_per_step" in d]
            over = max(0.0, gath[0]["ms_per_step"] - t1) if gath else None
            shards = {d["strong_scaling_shard_of"]: d for d in sec if "strong_scaling_shard_of" in d and "ms_per_launch" in d}
            pred = {"gather_overhead_ms_per_step": over,
                    "note": "weak: N x (512 replicas per GPU, this run's step + the one-rank gather overhead); strong: 512 replicas "
                            "over N GPUs, ms = the 512 / N-replica line of `secondary` + the same overhead; efficiency = value / "
                            "(N x this run's value).  Strong scaling leaves 256 - 512 / N of a GPU's 256 CUs without a replica "
                            "(one workgroup per replica, a replica is a serial chain): its efficiency is bounded by that, not by "
                            "communication", "weak": [], "strong": []}
            o = over or 0.0
            for n_ in (1, 2, 4, 8):
                tw = t1 + (o if n_ > 1 else 0.0)
                pred["weak"].append({"n_gpus": n_, "ms_per_step": tw, "value": n_ * Bg / (tw * 1e-3), "efficiency": t1 / tw})
                if n_ == 1:
                    pred["strong"].append({"n_gpus": 1, "replicas_per_gpu": Bg, "ms_per_step": t1, "value": Bg / (t1 * 1e-3), "efficiency": 1.0})
                elif n_ in shards:
                    ts_ = shards[n_]["ms_per_launch"] + o
                    pred["strong"].append({"n_gpus": n_, "replicas_per_gpu": shards[n_]["replicas"], "ms_per_step": ts_,
                                           "value": Bg / (ts_ * 1e-3), "efficiency": (Bg / (ts_ * 1e-3)) / (n_ * Bg / (t1 * 1e-3))})
            out["scaling_prediction"] = pred
        if world == 1 and not args.no_secondary and not args.stub:
            for d in sec:      # what the proof costs: the unverified rule of rounds 3-4 next to the headline
                if d.get("options") == {"conv_plan": 2, "spec_start": 7} and d.get("dtype") == args.dtype:
                    out["config"]["unverified_stop_rule"] = {"ms_per_step": d["ms_per_launch"], "value": d["value"], "unit": "env steps/s",
                                                             "note": "conv_plan=2, spec_start=7: NOT the default, see `secondary`"}
        print(json.dumps(out), flush=True)
    env.close()
    if distc:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
