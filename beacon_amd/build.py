"""In-tree build of libbeacon_hip.so (hand-written HIP for gfx950) with hipcc.

The shared object is git-ignored but travels to the GPU box with the snapshot.  Staleness is decided by
CONTENT, not by mtime (a snapshot or a fresh checkout may reset file times): the library carries a sidecar
`libbeacon_hip.so.sig` with a hash of every source, header and flag it was built from.  Concurrent builds
(N ranks of one node importing the package at once) are serialised by a file lock and build into a
per-process object directory."""
import fcntl
import glob
import hashlib
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
INC = os.path.join(os.path.dirname(PKG), "include")
LIB = os.path.join(PKG, "libbeacon_hip.so")
OBJ = os.path.join(PKG, "csrc", "_obj")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-fvisibility=hidden",
         "-fgpu-rdc" if False else "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable", "-Wno-unused-but-set-variable"]


# per-file flags.  ns2d_fast: the SLP vectoriser packs the Jacobi arithmetic into v_pk_* ops at the
# price of many register shuffles -- measured slower than the scalar stream on gfx950.
# ns2d_fast2 (256 VGPRs + ~100 spilled): without the scheduler's "unclustered high register pressure" re-scheduling stage the phases
# outside the Jacobi loop lose fewer cycles to spills than the loop gains -- measured twice, A/B/A/B on one box (round 6,
# scripts/variants.py): mixing 100x100 B=512 float32 29.83 -> 29.56 ms, float64 89.7 -> 88.2; the same flag on ns2d_fast: no gain.  Without the SDWA
# peephole on top: 29.92 -> 29.66 (1 301 -> 1 270 cycles per sweep).
# (nine other scheduler switches -- max-ilp / max-memory-clause / iterative-minreg strategies, AMDGPU trackers, no post-RA scheduler, no
# memop clustering, metric bias 0 / 100 -- gained nothing on any of the three kernels: profiles/r06_sched_flags_ab.log)
# ns2d_fast.hip (the float32 instantiations; the float64 ones are ns2d_fast_f64.hip, the same source): -O2 -- measured A/B/A/B on one
# box (round 6): the headline kernel 781 -> 772 cycles per sweep, 18.37 -> 18.18 ms over kstat's 8 steps; the float64 kernel loses
# 0.7 % with it and keeps -O3 (profiles/r06_sched_flags_ab.log)
FILE_FLAGS = {"ns2d_fast.hip": ["-fno-slp-vectorize", "-ffp-contract=on", "-O2"],
              "ns2d_fast_f64.hip": ["-fno-slp-vectorize", "-ffp-contract=on"],
              "ns2d_fast2.hip": ["-fno-slp-vectorize", "-ffp-contract=on", "-mllvm", "-amdgpu-disable-unclustered-high-rp-reschedule",
                                 "-mllvm", "-amdgpu-sdwa-peephole=0"],
              # float64 1D kernels: the reference's operation order without FMA contraction -> bit-identical fields
              "env1d_f64.hip": ["-ffp-contract=off"]}


# the on-demand kernels (beacon_amd/jit.py: csrc/jit/ns2d_jit.hip, any family, any precision): the flags their verification and
# timings were made with
JIT_FLAGS = ["-fno-slp-vectorize", "-ffp-contract=on"]


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    return None


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _deps():
    return (sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inc")) +
            glob.glob(os.path.join(INC, "*.h")))


def signature():
    """Hash of everything the library is built from: sources, headers, flags."""
    h = hashlib.sha256()
    h.update(repr((ARCH, FLAGS, sorted(FILE_FLAGS.items()))).encode())
    for f in sorted(_deps()):
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def stale():
    if not os.path.exists(LIB) or not os.path.exists(LIB + ".sig"):
        return True
    with open(LIB + ".sig") as fh:
        return fh.read().strip() != signature()


def sweep_objdirs():
    """Remove the per-process object directories of builders that are gone (a build that was killed leaves its
    csrc/_obj/p<pid> behind, and the directory ships with every snapshot of the tree)."""
    if not os.path.isdir(OBJ):
        return
    for d in os.listdir(OBJ):
        if not (d.startswith("p") and d[1:].isdigit()):
            continue
        pid = int(d[1:])
        if pid != os.getpid():
            try:
                os.kill(pid, 0)                        # still running: somebody else's build in progress
                continue
            except ProcessLookupError:
                pass
            except OSError:
                continue
        shutil.rmtree(os.path.join(OBJ, d), ignore_errors=True)


def build_lib(force=False, verbose=False):
    """Compile every csrc/*.hip for gfx950 and link libbeacon_hip.so.  Returns its path."""
    if not force and not stale():
        sweep_objdirs()
        return LIB
    cc = hipcc()
    if cc is None:
        raise RuntimeError("hipcc not found: cannot build libbeacon_hip.so")
    os.makedirs(OBJ, exist_ok=True)
    with open(os.path.join(OBJ, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)               # one builder at a time per checkout
        try:
            if not force and not stale():              # another process built it while we waited
                return LIB
            sig = signature()
            objdir = os.path.join(OBJ, "p%d" % os.getpid())
            os.makedirs(objdir, exist_ok=True)

            def one(src):
                obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
                cmd = [cc] + FLAGS + FILE_FLAGS.get(os.path.basename(src), []) + ["-I", INC, "-c", src, "-o", obj]
                if verbose:
                    print(" ".join(cmd), flush=True)
                subprocess.check_call(cmd)
                return obj

            with ThreadPoolExecutor(max_workers=4) as ex:
                objs = list(ex.map(one, sources()))
            tmp = "%s.tmp%d" % (LIB, os.getpid())
            cmd = [cc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", tmp] + objs
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            os.replace(tmp, LIB)
            with open(LIB + ".sig.tmp", "w") as fh:
                fh.write(sig + "\n")
            os.replace(LIB + ".sig.tmp", LIB + ".sig")
            shutil.rmtree(objdir, ignore_errors=True)
            sweep_objdirs()
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
