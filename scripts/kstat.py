# diagnostic: the bench workload for a few steps -- ms per step, in-kernel cycles per Jacobi sweep and per timestep outside
# the solve (bcn_get_counters), late stops / repeats.  usage: python scripts/kstat.py [dtype] [steps] [opt=value ...]
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
defs = [a for a in sys.argv[1:] if a.startswith("-D")]
sys.argv = [a for a in sys.argv if not a.startswith("-D")]
if defs:                                           # experiment build: python scripts/kstat.py f32 6 -DBCN_SOMETHING=1
    from beacon_amd import build
    build.FLAGS.extend(defs); build.build_lib(force=True)
from beacon_amd import vec as V
dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
opts = dict(a.split("=") for a in sys.argv[3:])
z = np.load("tests/golden/rayleigh_128x64_init.npz")
B = int(opts.pop("B", 512))
env = V.VecRayleigh(B, "cuda:0", dtype, z["fields"], L=2.56, H=1.28)
if "sched" in opts:
    env.set_sched(int(opts.pop("sched")))
for k, v in opts.items():
    env.set_option(k, int(v))
env.reset()
acts = torch.as_tensor(np.random.default_rng(1234).uniform(-1, 1, (steps + 2, B, 10)), dtype=env.tdtype, device="cuda:0")
for k in range(2):
    env.step(acts[k])
torch.cuda.synchronize()
ms, cj, ct, sw = [], 0.0, 0.0, 0.0
for k in range(steps):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); env.step(acts[2 + k]); e.record(); torch.cuda.synchronize()
    ms.append(s.elapsed_time(e))
    c = env.get_counters().astype(np.float64)
    cj += c[:, 0].sum(); ct += c[:, 1].sum(); sw += float(env.sweeps.sum())
env.check_status()
nts = steps * B * env.ndt_act
print(" ".join(defs), "%s %s B=%d: %.2f ms/step (min %.2f)  sweeps/dt %.1f  cycles/sweep %.0f  cycles/timestep outside the solve %.0f  jacobi share %.3f  late %d repeats %d"
      % (env.kernel_name, dtype, B, np.mean(ms), np.min(ms), sw / nts, cj / sw, (ct - cj) / nts, cj / ct, int(c[:, 2].sum()), int(c[:, 3].sum())))
