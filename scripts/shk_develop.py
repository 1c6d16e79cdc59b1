# diagnostic: does the N=4096 film develop from the flat state under inlet noise?  amplitude max|h-1| every 250 action steps for
# (a) noise drawn in the kernel, (b) an explicit noise tensor of the same law, in float32 and float64
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from beacon_amd import vec as V
B, NJ = 8, 10
for dt in ("f32", "f64"):
    for mode in ("kernel", "explicit"):
        env = V.VecShkadov(B, "cuda:0", dt, None, L0=699.2, n_jets=NJ, seed=11)
        env.reset()
        zero = torch.zeros((B, NJ), dtype=env.tdtype, device="cuda:0")
        g = torch.Generator(device="cuda:0"); g.manual_seed(5)
        out = []
        for k in range(2000):
            nz = None
            if mode == "explicit":
                nz = (2.0 * torch.rand((B, env.ndt_act), generator=g, device="cuda:0", dtype=env.tdtype) - 1.0) * env.sigma
            env.step(zero, nz)
            if k % 250 == 249:
                h = env.get_state()[:, 0]
                out.append("%.2e" % float((h - 1).abs().amax(dim=1).mean()))
        print(dt, mode, out, "h[0]-1 of replicas:", (env.get_state()[:4, 0, 0] - 1).cpu().numpy(), flush=True)
        env.close()
