#!/usr/bin/env python3
"""Secondary benchmark lines (BASELINE.json configs 2, 3, 5 + sloshing): env steps/s and effective GB/s
(SURVEY 8d algorithmic bytes / launch time) of the other solver kernels on one GPU, with the float64 C
oracle timed on a bounded sample next to each.  Not the headline metric (that is bench.py).
usage: python scripts/bench_envs.py [--steps K] [--no-cpu]"""
import argparse, ctypes as C, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from beacon_amd import vec as V
from beacon_amd.envs import packaged_init

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--no-cpu", action="store_true")
ap.add_argument("--only", default="")
ap.add_argument("--opt", action="append", default=[], help="name=value for bcn_set_option on every env (e.g. cells_per_thread=4)")
args = ap.parse_args()
dev = "cuda:0"


def timed(env, step_fn, K, W):
    for k in range(W):
        step_fn(k)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    t0 = time.perf_counter()
    for k in range(K):
        ev[k][0].record(); step_fn(W + k); ev[k][1].record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    return wall, float(np.mean([a.elapsed_time(b) for a, b in ev]))


def cpu_1d(make, step, seconds=8.0):
    from oracle import oracle as O
    e = make(O)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        step(e, n); n += 1
    return n / (time.perf_counter() - t0)


out = []
K, W = args.steps, args.warmup
rng = np.random.default_rng(7)

if not args.only or "burgers" in args.only:
    B = 1024
    env = V.VecBurgers(B, dev, "f32", nx=512)
    for o in args.opt:
        env.set_option(o.split("=")[0], int(o.split("=")[1]))
    env.reset()
    a = torch.as_tensor(rng.uniform(-1, 1, (K + W, B)), dtype=torch.float32, device=dev)
    nz = torch.as_tensor(rng.uniform(-0.1, 0.1, (K + W, B)), dtype=torch.float32, device=dev)
    wall, ms = timed(env, lambda k: env.step(a[k], nz[k]), K, W)
    bytes_ = 12.0 * env.nx * env.ndt_act * B
    r = {"env": "burgers-v0 B=1024 N=512 (BASELINE configs[1])", "kernel": env.kernel_name, "env_steps_per_s": B * K / wall,
         "launch_ms": ms, "effective_GBps": bytes_ / (ms * 1e-3) / 1e9, "dtype": "f32"}
    if not args.no_cpu:
        r["cpu_oracle_env_steps_per_s_1core"] = cpu_1d(lambda O: (lambda e: (e.reset(), e)[1])(O.burgers(nx=512)),
                                                       lambda e, n: e.step([0.3], 0.05))
    out.append(r); env.close()

if not args.only or "shkadov" in args.only:
    B = 1024
    env = V.VecShkadov(B, dev, "f32", None, L0=699.2, n_jets=10)
    for o in args.opt:
        env.set_option(o.split("=")[0], int(o.split("=")[1]))
    env.reset()
    # from a developed film (shkadov/init.py: 4000 uncontrolled action steps under inlet noise), as bench.py's line
    env.warmup(env.n_warmup_ref, torch.zeros((B, 10), dtype=torch.float32, device=dev))
    a = torch.as_tensor(rng.uniform(-1, 1, (K + W, B, 10)), dtype=torch.float32, device=dev)
    nz = torch.as_tensor(rng.uniform(-5e-4, 5e-4, (K + W, B, 50)), dtype=torch.float32, device=dev)
    wall, ms = timed(env, lambda k: env.step(a[k], nz[k]), K, W)
    bytes_ = 32.0 * env.nx * env.ndt_act * B
    r = {"env": "shkadov-v0 B=1024 10 jets N=4096 (BASELINE configs[2])", "kernel": env.kernel_name,
         "env_steps_per_s": B * K / wall, "launch_ms": ms, "effective_GBps": bytes_ / (ms * 1e-3) / 1e9, "dtype": "f32"}
    if not args.no_cpu:
        def mk(O):
            e = O.shkadov(init=False, L0=699.2, n_jets=10); e.reset_fields(); return e
        r["cpu_oracle_env_steps_per_s_1core"] = cpu_1d(mk, lambda e, n: e.step([0.1] * 10, np.zeros(50)))
    out.append(r); env.close()

if not args.only or "sloshing" in args.only:
    B = 1024
    env = V.VecSloshing(B, dev, "f32", packaged_init("sloshing"))
    for o in args.opt:
        env.set_option(o.split("=")[0], int(o.split("=")[1]))
    env.reset()
    a = torch.as_tensor(rng.uniform(-1, 1, (K + W, B)), dtype=torch.float32, device=dev)
    wall, ms = timed(env, lambda k: env.step(a[k]), K, W)
    bytes_ = 32.0 * env.nx * env.ndt_act * B
    r = {"env": "sloshing-v0 B=1024 N=200", "kernel": env.kernel_name, "env_steps_per_s": B * K / wall,
         "launch_ms": ms, "effective_GBps": bytes_ / (ms * 1e-3) / 1e9, "dtype": "f32"}
    if not args.no_cpu:
        def mk(O):
            e = O.sloshing(init_fields=packaged_init("sloshing")); e.reset(); return e
        r["cpu_oracle_env_steps_per_s_1core"] = cpu_1d(mk, lambda e, n: e.step([0.2]))
    out.append(r); env.close()

if not args.only or "mixing" in args.only:
    B = 512
    env = V.VecMixing(B, dev, "f32")
    env.reset()
    a = torch.as_tensor(rng.integers(0, 4, (K + W, B)), dtype=torch.int32, device=dev)
    Km, Wm = min(K, 4), 1
    wall, ms = timed(env, lambda k: env.step(a[k]), Km, Wm)
    sw = env.sweeps.cpu().numpy()
    bytes_ = env.nx * env.ny * 4.0 * (20.0 * sw.size + 3.0 * float(sw.sum()))
    r = {"env": "mixing-v0 B=512 100x100 (BASELINE configs[4]), first steps from rest", "kernel": env.kernel_name,
         "env_steps_per_s": B * Km / wall, "launch_ms": ms, "effective_GBps": bytes_ / (ms * 1e-3) / 1e9,
         "mean_sweeps_per_timestep": float(sw.mean()), "dtype": "f32"}
    out.append(r); env.close()

if "tall" in args.only:      # a grid above ny = 128 (ns2d_fast4_impl.h): mixing(L=1, H=2) = 100x200, one replica per CU
    B = 256
    env = V.VecMixing(B, dev, "f32", L=1.0, H=2.0)
    env.reset()
    a = torch.as_tensor(rng.integers(0, 4, (K + 3, B)), dtype=torch.int32, device=dev)
    Km, Wm = min(K, 3), 3
    wall, ms = timed(env, lambda k: env.step(a[k]), Km, Wm)
    sw = env.sweeps.cpu().numpy()
    bytes_ = env.nx * env.ny * 4.0 * (20.0 * sw.size + 3.0 * float(sw.sum()))
    r = {"env": "mixing-v0 L=1 H=2 (100x200) B=256", "kernel": env.kernel_name,
         "env_steps_per_s": B * Km / wall, "launch_ms": ms, "effective_GBps": bytes_ / (ms * 1e-3) / 1e9,
         "mean_sweeps_per_timestep": float(sw.mean()), "dtype": "f32"}
    out.append(r); env.close()

for r in out:
    print(json.dumps(r))
