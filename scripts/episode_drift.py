# diagnostic: where does the float32 drift over an episode come from?  rayleigh 128x64 B=32, bench action law, 100 steps:
# max |obs32 - obs64| every 5 steps for the default float32 kernel, conv_plan 0 (the literal stop rule), the generic kernel
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from beacon_amd import vec as V
B, N = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 100
init = np.load("tests/golden/rayleigh_128x64_init.npz")["fields"]
acts = np.random.default_rng(1234).uniform(-1.0, 1.0, (N, B, 10))
def run(dt, opts=None, variant=1):
    e = V.VecRayleigh(B, "cuda:0", dt, init, L=2.56, H=1.28)
    e.set_variant(variant)
    for k, v in (opts or {}).items(): e.set_option(k, v)
    e.reset()
    O, R, S, T = [], [], [], []
    for k in range(N):
        obs, rwd, _, _, _ = e.step(acts[k])
        O.append(obs.double().cpu().numpy()); R.append(rwd.double().cpu().numpy()); S.append(e.sweeps.cpu().numpy().copy())
        T.append(e.get_state()[:, 3].double().cpu().numpy())
    e.close()
    return np.array(O), np.array(R), np.array(S), np.array(T)
ref = run("f64")
for name, args in (("f32 default", ("f32",)), ("f32 conv_plan=0", ("f32", {"conv_plan": 0})), ("f32 generic", ("f32", None, 0)),
                   ("f64 conv_plan=3", ("f64", {"conv_plan": 3})), ("f64 generic", ("f64", None, 0))):
    o = run(*args)
    eo = np.abs(o[0] - ref[0]).max(axis=(1, 2)); er = np.abs(o[1] - ref[1]).max(axis=1)
    et = np.abs(o[3] - ref[3]).reshape(N, B, -1).max(axis=2)
    ds = np.abs(o[2].astype(int) - ref[2].astype(int))
    k = int(eo.argmax()); b = int(np.abs(o[0][k] - ref[0][k]).max(axis=1).argmax())
    print(name, "obs err every 5 steps:", " ".join("%.1e" % x for x in eo[::5]), flush=True)
    print("   max at step %d replica %d; T field err there %.2e; rwd err max %.2e; sweeps |diff| max %d, timesteps differing %.3f, total sweeps %d vs %d"
          % (k, b, et[k, b], er.max(), ds.max(), (ds > 0).mean(), o[2].sum(), ref[2].sum()))
    print("   per-replica max obs err:", " ".join("%.0e" % x for x in np.abs(o[0] - ref[0]).max(axis=(0, 2))))
