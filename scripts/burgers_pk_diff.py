# diagnostic: float32 burgers, packed kernel against the unpacked one-wave kernel and the float64 oracle (max abs difference per step)
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from beacon_amd import vec as V
from oracle import oracle as O
rng = np.random.default_rng(3)
for nx in (512, 256):
    B = 6
    envs = []
    for ow in (1, 2):
        e = V.VecBurgers(B, "cuda:0", "f32", nx=nx); e.set_option("one_wave", ow); e.reset(); envs.append(e)
    ors = [O.burgers(nx=nx) for _ in range(B)]
    for o in ors: o.reset()
    for k in range(40):
        a, nz = rng.uniform(-1, 1, B), rng.uniform(-0.1, 0.1, B)
        outs = [e.step(a, nz) for e in envs]
        st = [e.get_state().double().cpu().numpy() for e in envs]
        for b, o in enumerate(ors): o.step([a[b]], nz[b])
        ou = np.stack([o.u for o in ors])
        if k % 5 == 0 or k == 39:
            print(nx, k, "pk vs unpacked %.2e  pk vs oracle %.2e  unpacked vs oracle %.2e  rwd %.2e" % (
                np.abs(st[0] - st[1]).max(), np.abs(st[0][:, 0] - ou).max(), np.abs(st[1][:, 0] - ou).max(),
                float((outs[0][1] - outs[1][1]).abs().max())))

# the same for sloshing (sloshing_step_pk_k against sloshing_step_k)
from beacon_amd import envs as E
init = E.packaged_init("sloshing")
B = 6
envs = []
for ow in (1, 2):
    e = V.VecSloshing(B, "cuda:0", "f32", init); e.set_option("one_wave", ow); e.reset(); envs.append(e)
ors = [O.sloshing(init_fields=init) for _ in range(B)]
for o in ors: o.reset()
for k in range(40):
    a = rng.uniform(-1, 1, B)
    outs = [e.step(a) for e in envs]
    st = [e.get_state().double().cpu().numpy() for e in envs]
    rw = []
    for b, o in enumerate(ors): rw.append(o.step([a[b]])[1])
    oh = np.stack([o.h for o in ors]); oq = np.stack([o.q for o in ors])
    if k % 5 == 0 or k == 39:
        print("sloshing", k, "pk vs unpacked %.2e  pk vs oracle h %.2e q %.2e  unpacked vs oracle h %.2e  rwd pk-unpacked %.2e pk-oracle %.2e obs %.2e" % (
            np.abs(st[0] - st[1]).max(), np.abs(st[0][:, 0] - oh).max(), np.abs(st[0][:, 1] - oq).max(), np.abs(st[1][:, 0] - oh).max(),
            float((outs[0][1] - outs[1][1]).abs().max()), float(np.abs(outs[0][1].double().cpu().numpy() - np.array(rw)).max()),
            float((outs[0][0] - outs[1][0]).abs().max())))
