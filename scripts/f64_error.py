import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from conftest import golden, ref_to_dev
from beacon_amd import vec as V
def dev2ref(s): return np.swapaxes(s.detach().cpu().numpy().astype(np.float64), -1, -2)
g = golden("rayleigh_default")
init = np.stack([g["u_init"], g["v_init"], g["p_init"], g["T_init"]])
for variant in (0, 1):
    env = V.VecRayleigh(1, "cuda:0", "f64", init); env.set_variant(variant); env.reset()
    errs = []
    for k in range(len(g["actions"])):
        obs, rwd, *_ = env.step(g["actions"][k][None])
        st = dev2ref(env.get_state())[0]
        errs.append((max(np.abs(st[i] - g["step%d_%s" % (k, F)]).max() for i, F in enumerate("uvpT")),
                     np.abs(obs.cpu().numpy()[0] - g["step%d_obs" % k]).max(), int(np.abs(env.sweeps.cpu().numpy()[0] - g["itp"][k]).max())))
    print("rayleigh 50x50 f64 variant", variant, env.kernel_name, "per step (fields, obs, sweeps):", [("%.1e" % a, "%.1e" % b, c) for a, b, c in errs])
    env.close()
