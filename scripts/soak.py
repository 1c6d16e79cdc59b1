"""Soak: one full rayleigh episode (100 action steps) at the bench configuration, twice with the same
actions through the ticket scheduler; both runs must agree bit for bit and end with done = 1."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
from beacon_amd import vec as V
z = np.load("tests/golden/rayleigh_128x64_init.npz")
B = 512
acts = torch.as_tensor(np.random.default_rng(3).uniform(-1, 1, (100, B, 10)), dtype=torch.float32, device="cuda:0")
outs = []
for run in range(2):
    env = V.VecRayleigh(B, "cuda:0", "f32", z["fields"], L=2.56, H=1.28)
    env.reset()
    t0 = time.perf_counter()
    rw = []
    for k in range(env.n_act):
        obs, rwd, done, trunc, _ = env.step(acts[k])
        rw.append(rwd.clone())
    torch.cuda.synchronize()
    env.check_status()
    print("run", run, "kernel", env.kernel_name, "episode s %.2f" % (time.perf_counter() - t0), "done", int(done.sum().item()),
          "mean reward %.4f" % torch.stack(rw).mean().item(), "max sweeps/dt", int(env.sweeps.max().item()))
    outs.append((env.get_state().clone(), torch.stack(rw), obs.clone()))
    assert int(done.sum().item()) == B
    env.close()
assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
print("bitwise identical: OK")
