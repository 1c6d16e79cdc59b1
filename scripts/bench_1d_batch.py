"""1D envs at larger batches: throughput scaling of the one-workgroup-per-replica kernels."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
from beacon_amd import vec as V
from beacon_amd.envs import packaged_init
dev = "cuda:0"
def run(env, mk_args, K=20):
    env.reset()
    args = mk_args()
    env.step(*args); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K): env.step(*args)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K
for B in (1024, 4096, 16384, 65536):
    rng = np.random.default_rng(0)
    e = V.VecBurgers(B, dev, "f32", nx=512)
    a = torch.as_tensor(rng.uniform(-1, 1, B), dtype=torch.float32, device=dev); nz = torch.zeros(B, device=dev)
    tb = run(e, lambda: (a, nz)); e.close()
    e = V.VecSloshing(B, dev, "f32", packaged_init("sloshing"))
    ts = run(e, lambda: (a,)); e.close()
    line = "B=%6d  burgers N=512 %.3f ms/step %.1f M env steps/s | sloshing %.3f ms %.1f M" % (B, tb * 1e3, B / tb / 1e6, ts * 1e3, B / ts / 1e6)
    if B <= 16384:
        e = V.VecShkadov(B, dev, "f32", None, L0=699.2, n_jets=10)
        a2 = torch.as_tensor(rng.uniform(-1, 1, (B, 10)), dtype=torch.float32, device=dev); nz2 = torch.zeros((B, 50), device=dev)
        tk = run(e, lambda: (a2, nz2), K=5); e.close()
        line += " | shkadov N=4096 %.2f ms %.2f M" % (tk * 1e3, B / tk / 1e6)
    print(line, flush=True)
