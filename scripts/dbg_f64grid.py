import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
from beacon_amd import vec as V
from conftest import ref_to_dev
L, H, B = float(os.environ.get("DBG_L", "2.2")), float(os.environ.get("DBG_H", "1.28")), 2
NDT = int(sys.argv[1]) if len(sys.argv) > 1 else 2
env = V.VecRayleigh(B, "cuda:0", "f64", None, L=L, H=H); env.set_ndt_act(NDT)
for a in sys.argv[2:]:
    k_, v_ = a.split("="); env.set_option(k_, int(v_))
x, y = (np.arange(env.nx + 2) - 0.5) / env.nx, (np.arange(env.ny + 2) - 0.5) / env.ny
st0 = np.zeros((4, env.nx + 2, env.ny + 2))
st0[3] = (0.5 - y)[None, :] + 0.08 * np.sin(2 * np.pi * x * L)[:, None] * np.sin(np.pi * y)[None, :]
acts = np.random.default_rng(3).uniform(-1, 1, (B, 10))
out = {}
for variant in (1, 0):
    env.set_variant(variant); env.reset(); env.set_state(np.tile(ref_to_dev(st0)[None], (B, 1, 1, 1)))
    env.step(acts); torch.cuda.synchronize()
    out[variant] = env.get_state()[0].cpu().numpy()      # [4][ny+2][nx+2]
    print(variant, env.kernel_name, "status", env.status.cpu().tolist(), "sweeps:", env.sweeps[0].cpu().tolist())
for f, name in enumerate("uvpT"):
    d = np.abs(out[1][f] - out[0][f])
    j, i = np.unravel_index(d.argmax(), d.shape)
    cols = np.nonzero(d.max(axis=0) > 1e-9)[0]
    print(name, "max diff %.3e at i=%d j=%d; columns with diff > 1e-9: %s" % (d.max(), i, j, (cols.min(), cols.max(), len(cols)) if len(cols) else None))
d = np.abs(out[1][2] - out[0][2])
print("p: columns differing:", np.nonzero(d.max(axis=0) > 1e-9)[0].tolist())
print("p: rows differing:", np.nonzero(d.max(axis=1) > 1e-9)[0].tolist())
print("p diff at row 1:", np.round(d[1, :20], 4).tolist())
