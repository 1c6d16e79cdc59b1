# diagnostic: bench.py's mixing workload (B=512, 100x100, rng(7) wall choices) for a few steps -- ms per step, in-kernel cycles per
# Jacobi sweep and per timestep outside the solve.  usage: python scripts/mixstat.py [dtype] [steps] [-DFLAG ...] (-D: rebuild first)
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
defs = [a for a in sys.argv[1:] if a.startswith("-D")]
sys.argv = [a for a in sys.argv if not a.startswith("-D")]
if defs:
    from beacon_amd import build
    build.FLAGS.extend(defs); build.build_lib(force=True)
from beacon_amd import vec as V
dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
opts = dict(a.split("=") for a in sys.argv[3:] if "=" in a)
env = V.VecMixing(512, "cuda:0", dtype)
for k_, v_ in opts.items():
    env.set_option(k_, int(v_))
env.reset()
ai = torch.as_tensor(np.random.default_rng(7).integers(0, 4, (8, 512)), dtype=torch.int32, device="cuda:0")
for k in range(2):
    env.step(ai[k % 8])
torch.cuda.synchronize()
ms, cj, ct, sw = [], 0.0, 0.0, 0.0
for k in range(steps):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); env.step(ai[(2 + k) % 8]); e.record(); torch.cuda.synchronize()
    ms.append(s.elapsed_time(e))
    c = env.get_counters().astype(np.float64)
    cj += c[:, 0].sum(); ct += c[:, 1].sum(); sw += float(env.sweeps.sum())
env.check_status()
nts = steps * 512 * env.ndt_act
print(" ".join(defs), "%s %s: %.2f ms/step (min %.2f)  sweeps/dt %.1f  cycles/sweep %.0f  cycles/timestep outside the solve %.0f"
      % (env.kernel_name, dtype, np.mean(ms), np.min(ms), sw / nts, cj / sw, (ct - cj) / nts))
