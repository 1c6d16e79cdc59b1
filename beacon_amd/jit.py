"""Register-resident kernels for grids that are not built into libbeacon_hip.so.

The reference takes any domain size (rayleigh.py:20-27: nx = 50 L, ny = 50 H; mixing.py:20-28: nx = 100 L, ny = 100 H);
the register-resident kernels are templates over the grid (csrc/ns2d_fast_impl.h: one row per lane, ny <= 64;
csrc/ns2d_fast2_impl.h: two rows per lane, 64 < ny <= 128; csrc/ns2d_fast4_impl.h: 128 < ny <= 256, the Poisson solve
in registers with 2..4 rows per lane and the other phases from HBM/L2).  libbeacon_hip.so carries the metric grid and the
reference's defaults; for any other grid `plugin_for()` compiles csrc/jit/ns2d_jit.hip for that ONE grid with hipcc
(about 30 s, once: the shared object is cached in beacon_amd/_jit/, keyed by a hash of its sources and flags, and
travels with the tree like the library itself) and hands its launcher to the library through bcn_set_fast_plugin.
No hipcc, BEACON_JIT=0 or a grid the mapping cannot hold (ny > 256, LDS, registers) -> None: the env keeps the
generic kernel (still on the GPU; only slower).

First-use self-check.  A plugin is a NEW template instantiation compiled on the user's box, at the register limit of the
target (256 VGPRs plus spills), and round 4 met one such instantiation that computed wrong results under a neutral source
change (the 110x64 float64 two-body kernel).  So no plugin is attached on trust: the first env that wants it runs
`verify()` -- six timesteps of a seeded state through the plugin (plain launch and ticket scheduler) and through the generic
kernel, which is ordinary HIP without hand-scheduled code: float64 fields within 1e-9 with equal sweep counts (+-1), float32
within the on-demand grids' tolerance.  Agreement is remembered next to the shared object (`<plugin>.ok`, tagged with the device
and the HIP runtime, so another box checks again); a mismatch leaves `<plugin>.bad`, warns, and the env -- this one and every
later one -- keeps the generic kernel."""
import ctypes as C
import fcntl
import hashlib
import os
import subprocess
import warnings

from . import build as _build

class JitWarning(UserWarning):
    """Every warning of this module: a plugin that could not be built / loaded / checked, or that failed its self-check.
    tests/conftest.py turns the category into an error, so a refused plugin is a red test, not a line in a log."""


JIT_DIR = os.path.join(_build.PKG, "_jit")
JIT_SRC = os.path.join(_build.CSRC, "jit", "ns2d_jit.hip")
LDS_BYTES = 160 * 1024
_LOADED = {}


def _up16(x):
    return (x + 15) // 16 * 16


def _lds_rows1(nx, ny, nw, esz, gf, r=0):
    sy = ny + 2
    rl = nx - (nw - 1) * r if r else 0
    deadpad = (r - rl + 2) * sy if (gf == 2 and r and rl != r) else 0      # FastGeom::DEADPAD: the one-body form's dead columns
    sz = ((nx + 2) * (ny + 2) + 16 + deadpad + 63) // 64 * 64
    misc = 2 * nw * (2 if gf == 2 else 4) * 64 + 224 + 16
    front = _up16(max(misc, 63 * sy + 1))
    back = (ny + 3 * 8 + 2) * sy
    if gf == 1:
        return _up16(misc) * esz
    if gf == 2:
        return (_up16(misc) + 2 * sz) * esz
    return (front + 3 * sz + back) * esz


def _choose4(nx, ny, f64):
    """ns2d_fast4_impl.h: only the Poisson solve lives in registers (rpl = ceil(ny / 64) rows per lane, strips of r columns); 16
    waves of 128 VGPRs where phi and the rhs of a strip fit them, else 8 waves of 256; the fields stay in HBM/L2."""
    esz = 8 if f64 else 4
    if ny > 256:
        return None
    rpl = -(-ny // 64)
    for nwmax, words in ((16, 28), (8, 60)):
        if f64 and nwmax > 8:
            continue
        r = -(-nx // nwmax)
        nw = -(-nx // r)
        if nw < 2 or r * rpl * (2 if f64 else 1) > words:
            continue
        pitch = (nx + 2) | 1
        cap = (LDS_BYTES // esz - 146) // (3 * pitch)
        if cap < 1:
            continue
        nblk = -(-ny // cap)
        lds = (144 + max(2 * nw * 2 * 64 * rpl + pitch * (ny + 2), 3 * pitch * -(-ny // nblk) + 2)) * esz
        if lds <= LDS_BYTES:
            return {"rows": 4, "R": r, "gf": 0, "nw": nw, "rpl": rpl}
    return None


def choose(nx, ny, f64, kind):
    """Mapping of an nx x ny grid onto one workgroup: dict(rows, R, gf) or None.  Waves: 8 (two per SIMD, 256 VGPRs
    each) where the strips fit the register file, else 12 or 16; the last wave may take fewer columns (>= 3).
    Everything in registers / LDS where that fits (rows 1: ny <= 64, rayleigh; rows 2: ny <= 128), else the hybrid of
    ns2d_fast4_impl.h (rows 4: tall grids, wide grids, mixing below ny = 64)."""
    esz = 8 if f64 else 4
    if nx < 6 or ny < 4:
        return None
    if ny <= 64 and kind == 0:      # ns2d_fast_impl.h (one row per lane) implements the rayleigh boundary conditions only
        for nw, rmax in ((8, 16 if f64 else 26), (12, 18), (16, 12), (10, 20), (6, 26), (5, 26), (4, 26), (3, 26), (2, 26)):
            if f64 and nw > 8:
                continue
            r = -(-nx // nw)
            rl = nx - (nw - 1) * r
            if r > rmax or rl < 3 or rl > r or (nw - 1) * r >= nx:
                continue
            # float64: at most 16 columns per lane.  Strips of unequal width: round 4's two-body float64 kernel of 110x64 (strips of
            # 14 and 12 columns, two instantiations of the body in one kernel) computed ONE wrong word per timestep under a neutral
            # source change -- lane 0 of the column where the two bodies meet.  Since round 6 the float64 kernel with
            # T in the global scratch (gf = 2) has ONE body: the last strip is as wide as the others, its surplus columns dead
            # (ns2d_fast_impl.h: DEADC); the all-global variant (gf = 1) keeps equal strips only.  Every plugin, of whatever family,
            # is compared with the generic kernel before its first use (verify()).
            if f64 and r > 16:
                continue
            for gf in ((0,) if not f64 else ((2, 1) if rl == r else (2,))):
                if _lds_rows1(nx, ny, nw, esz, gf, r) <= LDS_BYTES:
                    return {"rows": 1, "R": r, "gf": gf, "nw": nw}
    elif 64 < ny <= 128:
        # float64: the fields live in a global scratch, the Poisson rhs in LDS (gf = 1); rayleigh and mixing alike
        for nw, rmax in (((8, 13), (7, 13), (6, 13), (5, 13), (4, 13)) if f64 else ((8, 16), (12, 10), (16, 7), (7, 16), (6, 20), (5, 24), (4, 26))):
            r = -(-nx // nw)
            rl = nx - (nw - 1) * r
            if r > rmax or rl < 3 or rl > r or (nw - 1) * r >= nx:
                continue
            exch = 2 * nw * 4 * 64 + 160
            lds = (exch + 2 * r * nw * 64) * esz if f64 else (exch + 3 * (nx + 2) * (ny + 2)) * esz
            if lds <= LDS_BYTES:
                return {"rows": 2, "R": r, "gf": 1 if f64 else 0, "nw": nw}
    return _choose4(nx, ny, f64)


CHECKING = False      # inside verify(): the envs it builds attach the plugin without verifying it again


def _runtime_tag():
    import torch
    try:
        name = torch.cuda.get_device_properties(torch.cuda.current_device()).gcnArchName
    except Exception:      # noqa: BLE001
        name = "?"
    return "%s hip %s" % (name, getattr(torch.version, "hip", None))


def _seeded_rayleigh_state(env):
    import numpy as np
    x, y = (np.arange(env.nx + 2) - 0.5) / env.nx, (np.arange(env.ny + 2) - 0.5) / env.ny
    st0 = np.zeros((4, env.nx + 2, env.ny + 2))
    st0[3] = (0.5 - y)[None, :] + 0.08 * np.sin(2 * np.pi * x * env.L)[:, None] * np.sin(np.pi * y)[None, :]
    return np.tile(np.ascontiguousarray(st0.transpose(0, 2, 1))[None], (env.batch, 1, 1, 1))


def compare_with_generic(make_env, kind, f64, ndt=6, batch=3):
    """(ok, report): `ndt` timesteps of a seeded state through the attached register-resident kernel -- plain launch and
    ticket scheduler (chunks of two timesteps on two persistent workgroups: hand-offs through HBM) -- against the generic
    kernel.  float64: fields within 1e-9 (p 5e-8), sweep counts within 1; float32: 2e-4 (p 1e-2), counts within max(3, 2 %)."""
    import numpy as np
    import torch
    runs = {}
    for tag, variant, sched in (("generic", 0, None), ("plain", 1, (0, 0, 0)), ("ticket", 1, (2, 2, 2))):
        env = make_env(batch)
        try:
            env.set_ndt_act(ndt)
            got = env.set_variant(variant)
            if sched is not None:
                env.set_sched(*sched)
            env.reset()
            if kind == 0:
                env.set_state(_seeded_rayleigh_state(env))
                a = np.random.default_rng(3).uniform(-1, 1, (batch, env.n_sgts))
            else:
                a = np.array([0, 2, 3] * batch)[:batch]
            env.step(a)
            torch.cuda.synchronize(env.device)
            runs[tag] = (env.get_state().double().cpu().numpy(), env.sweeps.cpu().numpy().astype(np.int64),
                         env.status.cpu().numpy().copy(), env.kernel_name, got)
        finally:
            env.close()
    ref = runs["generic"]
    tol = 1e-9 if f64 else 2e-4
    rep, ok = [], True
    for tag in ("plain", "ticket"):
        st, sw, status, kname, got = runs[tag]
        d = np.abs(st - ref[0])
        d = np.where(np.isfinite(d), d, np.inf)
        dp, dr = float(d[:, 2].max()), float(np.delete(d, 2, axis=1).max())
        ds = int(np.abs(sw - ref[1]).max())
        good = (got == 1 and not status.any() and not ref[2].any() and dr <= tol and dp <= 50 * tol and
                ds <= (1 if f64 else max(3, int(0.02 * ref[1].max()))))
        ok = ok and good
        rep.append("%s (%s): fields %.2e p %.2e sweeps %d of %d status %s%s" % (tag, kname, dr, dp, ds, int(ref[1].max()),
                                                                                 np.unique(status).tolist(), "" if good else "  <-- MISMATCH"))
    return ok, "; ".join(rep)


def _read(path):
    try:
        with open(path) as fh:
            return fh.read()
    except OSError:
        return None


def _verdict_on_disk(p, tag):
    """True / False from the marker files beside the shared object, None where there is none for this device + runtime."""
    bad = _read(p.path + ".bad")
    if bad is not None:
        return False, bad.strip()[:300]
    ok = _read(p.path + ".ok")
    if ok is not None and ok.split("\n")[0].strip() == tag:
        return True, ok
    return None, ""


def verify(p, make_env, kind, f64):
    """Sets p.verified (True / False) -- from the marker files next to the shared object, or by running compare_with_generic().
    One process per plugin runs the comparison: the N ranks of a fresh node queue on `<plugin>.vlock` and all but the first find
    the first one's verdict on disk when the lock comes to them."""
    global CHECKING
    tag = _runtime_tag()

    def settle(v, text):
        p.verified = v
        if not v:
            warnings.warn("beacon_amd.jit: %s failed its self-check earlier (%s); the generic kernel stays selected"
                          % (os.path.basename(p.path), text), JitWarning)

    v, text = _verdict_on_disk(p, tag)
    if v is not None:
        return settle(v, text)
    lock = None
    try:
        lock = open(p.path + ".vlock", "w")
        fcntl.flock(lock, fcntl.LOCK_EX)
    except OSError:
        lock = None                 # read-only package directory: every process checks for itself
    try:
        v, text = _verdict_on_disk(p, tag)       # written by another rank while this one waited
        if v is not None:
            return settle(v, text)
        CHECKING = True
        try:
            ok, rep = compare_with_generic(make_env, kind, f64)
        except Exception as e:      # noqa: BLE001 -- could not run (out of memory, ...): no verdict, no plugin for this env
            warnings.warn("beacon_amd.jit: the self-check of %s could not run (%s: %s); the generic kernel stays selected for this env"
                          % (os.path.basename(p.path), type(e).__name__, e), JitWarning)
            return
        finally:
            CHECKING = False
        p.verified, p.report = bool(ok), rep
        try:
            with open(p.path + (".ok" if ok else ".bad"), "w") as fh:
                fh.write("%s\n%s\n" % (tag, rep))
        except OSError:
            pass                    # read-only package directory: the verdict holds for this process
    finally:
        if lock is not None:
            fcntl.flock(lock, fcntl.LOCK_UN)
            lock.close()
    if not ok:
        warnings.warn("beacon_amd.jit: the register-resident kernel %s DISAGREES with the generic kernel on its self-check (%s); "
                      "it is not used: the generic kernel stays selected" % (os.path.basename(p.path), rep), JitWarning)


def selftest(device="cuda:0", verbose=False):
    """The register-resident kernels BUILT INTO libbeacon_hip.so through the comparison that guards a plugin's first use
    (compare_with_generic: six timesteps of a seeded state, plain launch and ticket scheduler, against the generic kernel).
    What to run after rebuilding the library with another ROCm / hipcc: these kernels sit at the register limit of the target
    and their code generation is the compiler's.  Returns {name: (ok, report)}."""
    from . import vec as V
    cases = [("rayleigh 128x64", 0, dict(L=2.56, H=1.28), ("f32", "f64")), ("rayleigh 50x50", 0, dict(), ("f32", "f64")),
             ("rayleigh 100x50", 0, dict(L=2.0), ("f32",)), ("rayleigh 150x50", 0, dict(L=3.0), ("f32",)),
             ("rayleigh 200x50", 0, dict(L=4.0), ("f32",)), ("rayleigh 100x100", 0, dict(L=2.0, H=2.0), ("f32",)),   # (float64: the generic kernel)
             ("mixing 100x100", 1, dict(), ("f32", "f64"))]
    out = {}
    for name, kind, kw, dts in cases:
        for dt in dts:
            if kind == 0:
                mk = lambda B, dt=dt, kw=kw: V.VecRayleigh(B, device, dt, None, **kw)
            else:
                mk = lambda B, dt=dt, kw=kw: V.VecMixing(B, device, dt, **kw)
            ok, rep = compare_with_generic(mk, kind, dt == "f64")
            out["%s %s" % (name, dt)] = (ok, rep)
            if verbose:
                print("%-24s %s  %s" % (name + " " + dt, "ok " if ok else "BAD", rep), flush=True)
    return out


def _signature(defs):
    h = hashlib.sha256()
    h.update(repr((_build.ARCH, _build.FLAGS, _build.JIT_FLAGS, sorted(defs.items()))).encode())
    for f in sorted([JIT_SRC] + [os.path.join(_build.CSRC, n) for n in os.listdir(_build.CSRC) if n.endswith(".h")] +
                    [os.path.join(_build.INC, n) for n in os.listdir(_build.INC) if n.endswith(".h")]):
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:12]


def build_plugin(nx, ny, f64, kind, verbose=False, extra_defs=None):
    """Path of the shared object for this grid (compiling it if needed), or None.  extra_defs: more -D flags (part of the
    plugin's name hash), e.g. {"BCN_JIT_BREAK": 1} -- the deliberately wrong kernel of the self-check's own test."""
    m = choose(nx, ny, f64, kind)
    if m is None or os.environ.get("BEACON_JIT", "1") == "0":
        return None
    defs = {"BCN_JIT_ROWS": m["rows"], "BCN_JIT_REAL": "double" if f64 else "float", "BCN_JIT_NX": nx, "BCN_JIT_NY": ny,
            "BCN_JIT_R": m["R"], "BCN_JIT_KIND": kind, "BCN_JIT_GF": m["gf"]}
    if m["rows"] == 4:
        defs["BCN_JIT_RPL"] = m["rpl"]
    for d in os.environ.get("BEACON_JIT_DEFS", "").split():   # experiments: extra -D flags (scripts/tall_base.py)
        defs[d.partition("=")[0]] = d.partition("=")[2] or "1"
    defs.update(extra_defs or {})
    name = "ns2d_%dx%d_%s_k%d_r%d_%s.so" % (nx, ny, "f64" if f64 else "f32", kind, m["R"], _signature(defs))
    path = os.path.join(JIT_DIR, name)
    if os.path.exists(path):
        return path
    cc = _build.hipcc()
    if cc is None or os.environ.get("BEACON_NO_BUILD") == "1":
        return None
    tmp = "%s.tmp%d" % (path, os.getpid())
    try:
        os.makedirs(JIT_DIR, exist_ok=True)
        with open(path + ".lock", "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)          # N ranks of one node asking for the same grid (one lock per plugin)
            try:
                if os.path.exists(path):
                    return path
                cmd = ([cc] + _build.FLAGS + _build.JIT_FLAGS +
                       ["-D%s=%s" % kv for kv in sorted(defs.items())] + ["-I", _build.INC, "-shared", JIT_SRC, "-o", tmp])
                if verbose:
                    print(" ".join(cmd), flush=True)
                subprocess.check_call(cmd)
                os.replace(tmp, path)
            finally:
                fcntl.flock(lock, fcntl.LOCK_UN)
    except (OSError, subprocess.CalledProcessError) as e:
        # a read-only package directory, a full disk or a compiler error: the env keeps the generic kernel
        # (still on the GPU, only slower) instead of failing in its constructor
        warnings.warn("beacon_amd.jit: no register-resident kernel for %dx%d (%s); the generic kernel stays selected"
                      % (nx, ny, e), JitWarning)
        try:
            os.remove(tmp)
        except OSError:
            pass
        return None
    return path


class Plugin(object):
    def __init__(self, path):
        self.path = path
        self.lib = C.CDLL(path)                       # bcn_set_error resolves against libbeacon_hip.so (RTLD_GLOBAL)
        self.lib.bcn_jit_scratch_elems.restype = C.c_size_t
        self.lib.bcn_jit_lds_bytes.restype = C.c_size_t
        self.fn = C.cast(self.lib.bcn_jit_launch, C.c_void_p)
        self.scratch = int(self.lib.bcn_jit_scratch_elems())
        self.lds = int(self.lib.bcn_jit_lds_bytes())
        self.verified = None          # True / False once verify() has compared it with the generic kernel
        self.report = ""


def plugin_for(nx, ny, f64, kind, extra_defs=None):
    """Loaded plugin (kept alive for the life of the process) for this grid, or None."""
    key = (nx, ny, bool(f64), kind, tuple(sorted((extra_defs or {}).items())))
    if key not in _LOADED:
        path = build_plugin(nx, ny, f64, kind, extra_defs=extra_defs)
        try:
            p = Plugin(path) if path else None
        except (OSError, AttributeError) as e:        # a truncated or foreign shared object in the cache
            warnings.warn("beacon_amd.jit: cannot load %s (%s); the generic kernel stays selected" % (path, e), JitWarning)
            p = None
        if p is not None and p.lds > LDS_BYTES:
            p = None
        _LOADED[key] = p
    return _LOADED[key]


# grids of the -m gpu tests (tests/test_gpu_parity.py::test_jit_grids_*): built by __graft_entry__.build() so that
# they ship with the tree; any other grid compiles at its first use
TEST_GRIDS = [(75, 50, False, 0), (75, 50, True, 0), (53, 50, False, 0), (110, 64, False, 0), (110, 64, True, 0),
              (50, 70, False, 0), (60, 120, False, 0), (100, 110, False, 1), (150, 50, True, 0), (110, 65, False, 0),
              (100, 105, False, 1), (50, 70, True, 0), (50, 150, False, 0), (50, 150, True, 0), (64, 200, False, 0),
              (100, 200, False, 1), (50, 145, False, 0), (50, 149, True, 0), (100, 130, False, 1),
              (300, 50, False, 0), (200, 100, False, 1), (53, 150, False, 0), (106, 200, False, 1),
              # float64, two rows per lane, ODD ny, strips of unequal width (9 / 5, 10 / 7): the class of the wrong kernel of round 5
              (50, 75, True, 0), (57, 107, True, 0), (100, 105, True, 1), (50, 75, False, 0),
              # float64, one row per lane, ONE body with dead columns in the last strip (round 6): 7 x 7 + 3 live, 7 x 16 + 13 live
              (52, 50, True, 0), (125, 55, True, 0)]


def fuzz_grids(seed=5):
    """22 random domain sizes with a register-resident mapping, as (L, H, f64, kind): the reference takes any L, H
    (rayleigh.py:20-27, mixing.py:20-28).  Drawn until every kernel family has its quota -- (rows per lane, float64):
    one row 4 + 3, two rows 4 + 3, the hybrid 6 + 2.  Seeded, so that __graft_entry__.build() compiles exactly the plugins
    that tests/test_gpu_parity.py::test_jit_fuzzed_grids_agree_with_the_generic_kernel asks for."""
    import numpy as np
    rng = np.random.default_rng(seed)
    quota = {(1, False): 4, (1, True): 3, (2, False): 4, (2, True): 3, (4, False): 6, (4, True): 2}
    out = []
    while any(quota.values()):
        kind = int(rng.integers(0, 2))
        f64 = bool(rng.integers(0, 2))
        if kind == 0:        # (below L, H = 1 the reference divides by zero: nx_obs_pts = 4 int(L))
            L = round(float(rng.uniform(1.0, 4.5)), 2)
            H = round(float(rng.uniform(1.0, 1.3) if rng.integers(0, 2) else rng.uniform(1.0, 3.2)), 2)
            nx, ny = int(50 * L), int(50 * H)
        else:
            L, H = round(float(rng.uniform(1.0, 1.7)), 2), round(float(rng.uniform(1.0, 2.3)), 2)
            nx, ny = int(100 * L), int(100 * H)
        m = choose(nx, ny, f64, kind)
        if m is None or nx * ny > 30000 or not quota.get((m["rows"], f64)):
            continue
        quota[(m["rows"], f64)] -= 1
        out.append((L, H, f64, kind))
    return out


def _grid(L, H, kind):
    n = 50 if kind == 0 else 100
    return int(n * L), int(n * H)


def fuzz_cases(seed=6):
    """fuzz_grids() (default constructor arguments) plus 15 draws that cover what those 22 never met (the
    wrong 50x75 float64 kernel was the only float64 two-rows-per-lane grid with ODD ny anyone had built, and it ran with
    n_sgts = 5, ra = 5e4): per kernel family and precision -- (rows per lane, float64) -- an odd-ny quota (one row 2 + 2, two rows
    rayleigh 2 + 2 and mixing 2 + 1, the hybrid 1 + 1 each), every case with the reference's OTHER constructor arguments drawn as well: rayleigh n_sgts in 1..12
    and ra in [8e3, 2e5] (rayleigh.py:20-27), mixing re in [50, 400], pe in [1e3, 1e5], side in [0.3, 0.7], C0 in [0.5, 2]
    (mixing.py:20-34).  Returns (L, H, f64, kind, kwargs)."""
    import numpy as np
    out = [(L, H, f64, kind, {}) for L, H, f64, kind in fuzz_grids()]
    rng = np.random.default_rng(seed)
    quota = {(1, False, 0): 2, (1, True, 0): 2, (2, False, 0): 2, (2, True, 0): 2, (2, False, 1): 2, (2, True, 1): 1,
             (4, False, 0): 1, (4, True, 0): 1, (4, False, 1): 1, (4, True, 1): 1}
    while any(quota.values()):
        kind = int(rng.integers(0, 2))
        f64 = bool(rng.integers(0, 2))
        if kind == 0:
            L = round(float(rng.uniform(1.0, 4.5)), 2)
            H = round(float(rng.uniform(1.0, 1.3) if rng.integers(0, 2) else rng.uniform(1.0, 3.2)), 2)
            kw = dict(n_sgts=int(rng.integers(1, 13)), ra=float(round(10.0 ** rng.uniform(np.log10(8e3), np.log10(2e5)), -2)))
        else:
            L, H = round(float(rng.uniform(1.0, 1.7)), 2), round(float(rng.uniform(1.0, 2.3)), 2)
            kw = dict(re=float(round(rng.uniform(50, 400), 0)), pe=float(round(10.0 ** rng.uniform(3, 5), -1)),
                      side=round(float(rng.uniform(0.3, 0.7)), 2), C0=round(float(rng.uniform(0.5, 2.0)), 2))
        nx, ny = _grid(L, H, kind)
        m = choose(nx, ny, f64, kind)
        if m is None or ny % 2 == 0 or nx * ny > 30000 or not quota.get((m["rows"], f64, kind)):
            continue
        quota[(m["rows"], f64, kind)] -= 1
        out.append((L, H, f64, kind, kw))
    return out


def fuzz_grid_keys():
    return [_grid(L, H, kind) + (f64, kind) for L, H, f64, kind, _ in fuzz_cases()]


# plugins built with extra -D flags: the deliberately wrong kernel of the self-check's own test
EXTRA_BUILDS = [((75, 50, True, 0), {"BCN_JIT_BREAK": 1})]


def _bounds_job(cell):
    from . import stoprule
    try:
        from threadpoolctl import threadpool_limits
        with threadpool_limits(limits=2):
            stoprule.bounds(*cell)
    except ImportError:
        stoprule.bounds(*cell)


def prebuild(grids=None, verbose=False):
    """Compile the plugins of a list of (nx, ny, f64, kind) grids (used by __graft_entry__.build() for the grids the
    tests touch, so that they ship with the tree)."""
    from concurrent.futures import ThreadPoolExecutor
    todo = [(g, None) for g in (grids or (TEST_GRIDS + [k for k in fuzz_grid_keys() if k not in TEST_GRIDS]))]
    if grids is None:
        todo += EXTRA_BUILDS
    with ThreadPoolExecutor(max_workers=int(os.environ.get("BEACON_JIT_JOBS", "4"))) as ex:
        paths = list(ex.map(lambda t: build_plugin(t[0][0], t[0][1], t[0][2], t[0][3], verbose, t[1]), todo))
    # the slow-mode constants of the same grids (beacon_amd/stoprule.py; only the one-row-per-lane rayleigh kernels use them),
    # cached in JIT_DIR too: a test on the GPU box then reads them instead of spending a minute of dense algebra
    from . import stoprule
    cells = [(nx, ny, kind, 0.25) for (nx, ny, f64, kind), _ in todo]
    if grids is None:
        for L, H, f64, kind, _ in fuzz_cases():
            nx, ny = _grid(L, H, kind)
            dx, dy = float(L / nx), float(H / ny)
            cells.append((nx, ny, kind, dy * dy / (2.0 * (dx * dx + dy * dy))))
    want = [c for c in sorted(set(cells)) if c[1] <= 128 and min(c[0], c[1]) >= 48 and not os.path.exists(stoprule._cache_file(*c))]
    if want:      # ~10 s of dense algebra per grid: a few grids at a time, two BLAS threads each
        from concurrent.futures import ProcessPoolExecutor
        with ProcessPoolExecutor(max_workers=int(os.environ.get("BEACON_JIT_JOBS", "4"))) as ex:
            list(ex.map(_bounds_job, want))
    if os.path.isdir(JIT_DIR):
        # drop the plugins of older source states of THESE grids (same name up to the hash); plugins that users compiled
        # on demand for other grids stay (a stale one is merely unused: build_plugin() compiles the current hash next to it)
        keep = {os.path.basename(p) for p in paths if p}
        stems = {k.rsplit("_", 1)[0] for k in keep}
        for f in os.listdir(JIT_DIR):
            if f.endswith(".so") and f not in keep and f.rsplit("_", 1)[0] in stems:
                os.remove(os.path.join(JIT_DIR, f))
                for mark in (".ok", ".bad"):
                    if os.path.exists(os.path.join(JIT_DIR, f + mark)):
                        os.remove(os.path.join(JIT_DIR, f + mark))
            elif f.endswith(".lock"):
                # a lock file another process holds (flock) must stay: unlinking it would let a third process lock a NEW
                # file of the same name and compile the same plugin beside the holder
                lp = os.path.join(JIT_DIR, f)
                try:
                    with open(lp, "r+") as lk:
                        fcntl.flock(lk, fcntl.LOCK_EX | fcntl.LOCK_NB)
                        try:
                            os.remove(lp)
                        finally:
                            fcntl.flock(lk, fcntl.LOCK_UN)
                except OSError:
                    pass
    return paths
