# diagnostic: measured float32 errors against the float64 C oracle on the two full-step workloads of the parity tests
# (rayleigh bench dispatch, replicas 0..7; mixing B=512 from rest, replicas 0..3): sets the tolerances of tests/test_gpu_parity.py
import os, sys, numpy as np, torch
ROOT = os.getcwd(); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T
from oracle import oracle as O
env, init, acts = T._bench_workload(512, 1, "f32")
obs, rwd, _, _, _ = env.step(acts[0]); env.check_status()
st = T.dev2ref(env.get_state()[:8]); sw = env.sweeps.cpu().numpy()[:8]
ost, oobs, orwd, osw = T._oracle_batch_step(init, acts[0], 8)
for i, F in enumerate("uvpT"):
    print("rayleigh 128x64 f32 %s: max |err| %.2e" % (F, max(T.maxdiff(st[b][i], ost[b][i]) for b in range(8))))
print("   obs %.2e rwd %.2e sweeps rel %.4f" % (max(T.maxdiff(obs.cpu().numpy()[b], oobs[b]) for b in range(8)),
      max(abs(float(rwd[b]) - orwd[b]) for b in range(8)), max(abs(int(sw[b].sum()) - int(osw[b])) / osw[b] for b in range(8))))
env.close()
env = T.V.VecMixing(512, "cuda:0", "f32"); env.reset()
a = (np.arange(512) % 4).astype(np.int64)
obs, rwd, _, _, _ = env.step(a); env.check_status()
st = T.dev2ref(env.get_state()[:4]); sw = env.sweeps.cpu().numpy()
for b in range(4):
    o = O.mixing(); o.reset(); ob, rw, _, _, _ = o.step(int(a[b]))
    print("mixing a=%d f32: " % b + " ".join("%s %.2e" % (F, T.maxdiff(st[b][i], o.st[i])) for i, F in enumerate("uvpC")),
          "obs %.2e rwd %.2e sweeps max rel %.4f abs %d" % (T.maxdiff(obs[b].cpu().numpy(), ob), abs(float(rwd[b]) - rw),
          np.max(np.abs(sw[b] - o.itp) / np.maximum(o.itp, 1)), np.max(np.abs(sw[b] - o.itp))))
