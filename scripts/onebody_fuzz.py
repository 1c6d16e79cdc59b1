# diagnostic: random float64 one-row-per-lane grids (ny <= 64) whose strips are NOT all equally wide -- the one-body form of round 6 --
# with random n_sgts / ra, against the generic kernel (12 timesteps, plain launch and ticket scheduler).  usage: onebody_fuzz.py [n] [seed]
import os, sys, warnings
sys.path.insert(0, os.getcwd())
warnings.simplefilter("ignore")
import numpy as np
from beacon_amd import jit, vec as V
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
done, bad = 0, 0
while done < n:
    L, H = round(float(rng.uniform(1.0, 2.56)), 2), round(float(rng.uniform(1.0, 1.28)), 2)
    nx, ny = int(50 * L), int(50 * H)
    m = jit.choose(nx, ny, True, 0)
    if not m or m["rows"] != 1 or nx - (m["nw"] - 1) * m["R"] == m["R"]:
        continue
    kw = dict(n_sgts=int(rng.integers(1, 13)), ra=float(round(10.0 ** rng.uniform(np.log10(8e3), np.log10(2e5)), -2)))
    mk = lambda B: V.VecRayleigh(B, "cuda:0", "f64", None, L=L, H=H, **kw)
    ok, rep = jit.compare_with_generic(mk, 0, True, ndt=12, batch=3)
    done += 1
    bad += not ok
    print("%3dx%-3d R %2d x %d waves, last strip %2d live  %s  %s  %s" % (nx, ny, m["R"], m["nw"], nx - (m["nw"] - 1) * m["R"], kw,
                                                                     "ok " if ok else "BAD", rep.split(";")[0] if ok else rep), flush=True)
print("cases %d, bad %d" % (done, bad))
sys.exit(1 if bad else 0)
