# diagnostic: does the Jacobi work of the first Q timesteps predict a replica's work for the rest of the step?
import os, sys, heapq, numpy as np, torch
sys.path.insert(0, os.getcwd())
from beacon_amd import vec as V
z = np.load("tests/golden/rayleigh_128x64_init.npz")
B = 512
env = V.VecRayleigh(B, "cuda:0", "f32", z["fields"], L=2.56, H=1.28)
env.reset()
acts = np.random.default_rng(1234).uniform(-1, 1, (5, B, 10))
def makespan(order, w, t0=None):
    h = [0.0] * 256 if t0 is None else list(t0); heapq.heapify(h)
    for i in order: heapq.heappush(h, heapq.heappop(h) + w[i])
    return max(h)
for k in range(5):
    env.step(acts[k]); torch.cuda.synchronize()
    sw = env.sweeps.cpu().numpy().astype(np.float64) + 4.0     # + ~4 sweep-equivalents of non-Poisson work per timestep
    tot = sw.sum(1)
    line = "step %d mean %.0f crit %.0f | index-order %.0f | LPT(true) %.0f" % (k, tot.sum() / 256, tot.max(), makespan(range(B), tot), makespan(np.argsort(-tot), tot))
    for Q in (4, 10, 20):
        first, rest = sw[:, :Q].sum(1), sw[:, Q:].sum(1)
        r = np.corrcoef(first, rest)[0, 1]
        m1 = makespan(range(B), first)                     # launch 1 in index order
        m2 = makespan(np.argsort(-first), rest)            # launch 2 in predicted-LPT order
        line += " | Q=%d r=%.3f two-launch %.0f" % (Q, r, m1 + m2)
    print(line)
