"""Stress: 300 action steps of the bench configuration with per-replica auto-reset (episodes end at different
times because the replicas start at staggered step counters), ticket scheduler, masks; every status word must
stay 0 and the throughput is reported."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
from beacon_amd import vec as V
z = np.load("tests/golden/rayleigh_128x64_init.npz")
B, N = 512, 300
env = V.VecRayleigh(B, "cuda:0", "f32", z["fields"], L=2.56, H=1.28)
env.reset()
env.set_stp(np.arange(B) % env.n_act)                     # staggered episode ends
rng = np.random.default_rng(11)
acts = torch.as_tensor(rng.uniform(-1, 1, (8, B, 10)), dtype=torch.float32, device="cuda:0")
t0 = time.perf_counter(); resets = 0
for k in range(N):
    obs, rwd, done, trunc, _ = env.step(acts[k % 8])
    n = int(done.sum().item())
    if n:
        env.reset_done(); resets += n
    if (k + 1) % 100 == 0:
        st = env.status.cpu().numpy()
        assert (st == 0).all(), st[st != 0]
        print("step %d: %.1f env steps/s, %d resets so far, reward mean %.4f" % (k + 1, B * (k + 1) / (time.perf_counter() - t0), resets, rwd.mean().item()), flush=True)
env.check_status()
print("stress ok: kernel", env.kernel_name)
