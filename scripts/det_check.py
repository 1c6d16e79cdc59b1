# diagnostic: identical replicas must give bit-identical results (race detector)
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from beacon_amd import build
extra = [a for a in sys.argv[1:] if a.startswith("-D")]
if extra:
    build.FLAGS.extend(extra); build.build_lib(force=True)
from beacon_amd import vec as V
from beacon_amd.envs import packaged_init
for (cfg, dtype) in (("50", "f32"), ("50", "f64"), ("128", "f32")):
    B = 96
    if cfg == "50":
        env = V.VecRayleigh(B, "cuda:0", dtype, packaged_init("rayleigh"))
    else:
        z = np.load("tests/golden/rayleigh_128x64_init.npz")
        env = V.VecRayleigh(B, "cuda:0", dtype, z["fields"], L=2.56, H=1.28)
    env.reset()
    rng = np.random.default_rng(0)
    bad = 0
    for k in range(4):
        a = np.tile(rng.uniform(-1, 1, (1, 10)), (B, 1))
        obs, *_ = env.step(a)
        st = env.get_state()
        sw = env.sweeps
        nb = int((st != st[0:1]).flatten(1).any(1).sum())
        ns = int((sw != sw[0:1]).any(1).sum())
        bad += nb
        print(cfg, dtype, "step", k, "replicas differing from #0: state", nb, "sweeps", ns, env.kernel_name)
    env.close()
