# diagnostic: how well does the action change predict a replica's Jacobi work?
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from beacon_amd import vec as V
z = np.load("tests/golden/rayleigh_128x64_init.npz")
B = 512
env = V.VecRayleigh(B, "cuda:0", "f32", z["fields"], L=2.56, H=1.28)
env.reset()
acts = np.random.default_rng(1234).uniform(-1, 1, (6, B, 10))
prev = np.zeros((B, 10))
for k in range(6):
    env.step(acts[k]); torch.cuda.synchronize()
    an = env.actions_norm.cpu().numpy().astype(np.float64)
    sw = env.sweeps.cpu().numpy().sum(1).astype(np.float64)
    d = an - prev
    keys = {"l2(da)": np.sqrt((d ** 2).sum(1)), "linf(da)": np.abs(d).max(1), "l1(da)": np.abs(d).sum(1),
            "l2(a)": np.sqrt((an ** 2).sum(1)), "tv(da)": np.abs(np.diff(d, axis=1)).sum(1)}
    print("step", k, "sweeps mean %.0f max %.0f" % (sw.mean(), sw.max()),
          " ".join("%s r=%.3f" % (n, np.corrcoef(v, sw)[0, 1]) for n, v in keys.items()))
    # LPT quality: makespan on 256 machines with list scheduling in key order vs index order
    def makespan(order):
        import heapq
        h = [0.0] * 256; heapq.heapify(h)
        for i in order: heapq.heappush(h, heapq.heappop(h) + sw[i])
        return max(h)
    print("   makespan index-order %.0f | by l2(da) %.0f | by true work (LPT) %.0f | lower bound max(crit %.0f, mean %.0f)" % (
        makespan(range(B)), makespan(np.argsort(-keys["l2(da)"])), makespan(np.argsort(-sw)), sw.max(), sw.sum() / 256))
    prev = an
