# diagnostic: ms per action step of rayleigh 110x64 (and 75x50) float64, B = 512: the one-body register-resident kernel (default since
# round 6) against the hybrid kernel of ns2d_fast4_impl.h that such grids took before, and the generic kernel
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from beacon_amd import jit, vec as V
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for L, H in ((2.2, 1.28), (1.5, 1.0)):
    for mode in ("one-body", "hybrid", "generic"):
        if mode == "hybrid":
            orig = jit.choose
            jit.choose = lambda nx, ny, f64, kind: jit._choose4(nx, ny, f64)
            jit._LOADED.clear()
        env = V.VecRayleigh(B, "cuda:0", "f64", None, L=L, H=H)
        if mode == "generic":
            env.set_variant(0)
        st = env.perturbed_conduction_state()
        env.reset()
        env.set_state(np.tile(np.ascontiguousarray(st.transpose(0, 2, 1))[None], (B, 1, 1, 1)))
        acts = torch.as_tensor(np.random.default_rng(1).uniform(-1, 1, (4, B, 10)), dtype=env.tdtype, device="cuda:0")
        env.step(acts[0]); torch.cuda.synchronize()
        ms = []
        for k in range(1, 4 if mode != "generic" else 2):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); env.step(acts[k]); e.record(); torch.cuda.synchronize(); ms.append(s.elapsed_time(e))
        env.check_status()
        print("%dx%d float64 B=%d %-9s %-18s %.2f ms per action step, %.1f sweeps per timestep" %
              (env.nx, env.ny, B, mode, env.kernel_name, np.mean(ms), env.sweeps.float().mean().item()), flush=True)
        env.close()
        if mode == "hybrid":
            jit.choose = orig
            jit._LOADED.clear()
