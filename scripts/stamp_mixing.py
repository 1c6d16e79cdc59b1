import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from beacon_amd import build
extra = sys.argv[1:]
for x in [x for x in extra if x.startswith("-ffp-contract=")]:      # per-file flags come last on the command line
    build.FILE_FLAGS = {k: [f for f in v if not f.startswith("-ffp-contract=")] + [x] for k, v in build.FILE_FLAGS.items()}
build.FLAGS.extend(["-DBCN_STAMP"] + extra); build.build_lib(force=True)
from beacon_amd import vec as V
env = V.VecMixing(512, "cuda:0", os.environ.get("BCN_STAMP_DTYPE", "f32")); env.set_ndt_act(40)
if os.environ.get("BCN_TI"): env.set_option("transport_iter", int(os.environ["BCN_TI"]))
env.reset()
for k in range(2):
    env.step(np.full(512, k, dtype=np.int64))
print(extra, "sweeps/dt", env.sweeps.float().mean().item(), "cycles/timestep BC/pred+rhs/jacobi/corr/transp-expl/chain", env.obs.cpu().numpy()[:, :6].mean(0).round(0))
