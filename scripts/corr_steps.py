"""Per-replica Jacobi work of consecutive action steps on the bench workload: correlation and what a
longest-first order taken from the previous step would be worth (list-scheduling replay on 256 CUs)."""
import os, sys, heapq, numpy as np, torch
sys.path.insert(0, os.getcwd())
from beacon_amd import vec as V
z = np.load("tests/golden/rayleigh_128x64_init.npz")
B = 512
env = V.VecRayleigh(B, "cuda:0", "f32", z["fields"], L=2.56, H=1.28); env.reset()
acts = np.random.default_rng(1234).uniform(-1, 1, (8, B, 10))
W = []
for k in range(8):
    env.step(acts[k]); W.append(env.sweeps.cpu().numpy().astype(np.float64))
W = np.stack(W)                      # [step, replica, timestep]
tot = W.sum(-1)
print("corr(total work step k, step k+1):", [round(float(np.corrcoef(tot[k], tot[k + 1])[0, 1]), 3) for k in range(7)])
def replay(work_chunks, order, ncu=256):
    """chunk-major ticket scheduler: units (c, r) in order; a unit starts when a CU is free AND (c-1, r) is done."""
    nchunk, nrep = work_chunks.shape[1], work_chunks.shape[0]
    cu = [0.0] * ncu; heapq.heapify(cu); done = np.zeros(nrep)
    for c in range(nchunk):
        for r in order:
            t = heapq.heappop(cu); start = max(t, done[r]); end = start + work_chunks[r, c]
            done[r] = end; heapq.heappush(cu, end)
    return max(done)
for k in range(2, 8):
    wc = (W[k].reshape(B, 20, 10).sum(-1) * 1238 + 10 * 29000) / 2.33e9 * 1e3      # ms per chunk
    idx = np.arange(B); lpt_prev = np.argsort(-tot[k - 1]); lpt_true = np.argsort(-tot[k])
    print("step %d: bound %.2f ms (mean) / %.2f (slowest replica); replay index order %.2f, by previous step %.2f, by own work (oracle) %.2f" % (
        k, wc.sum() / 256, wc.sum(1).max(), replay(wc, idx), replay(wc, lpt_prev), replay(wc, lpt_true)))
