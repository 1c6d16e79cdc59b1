"""The thin PyTorch-ROCm extension over the C ABI: torch.library ops `torch.ops.beacon.<env>_{step,reset}`.

csrc/torch/beacon_torch.cpp (host code, g++) registers ten ops that take device tensors, read torch's current HIP stream
in C++ and call the bcn_* entry points of libbeacon_hip.so (include/beacon_hip.h).  beacon_amd.vec uses them for reset() /
step() when this library is present -- one dispatcher call per step instead of seven c_void_p conversions and a Python-side
stream query -- and falls back to the ctypes binding of the SAME C ABI when it is not (no g++ / no torch headers): either
way the arithmetic happens in libbeacon_hip.so, and tests/test_gpu_parity.py steps every env through both bindings bit for bit.

Built in-tree like the library itself (the shared object travels with the snapshot; its sidecar .sig holds a hash of the
source, the C header, the flags and the torch version)."""
import fcntl
import hashlib
import os
import shutil
import subprocess

from . import build as _build

SRC = os.path.join(_build.CSRC, "torch", "beacon_torch.cpp")
LIB = os.path.join(_build.PKG, "libbeacon_torch.so")
_LOADED = None


def _flags():
    import torch
    from torch.utils import cpp_extension as ce
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    inc = []
    for p in ce.include_paths("cuda") if "device_type" in ce.include_paths.__code__.co_varnames else ce.include_paths(True):
        inc += ["-isystem", p]
    return (["-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
             # ROCm's HIP headers pick the AMD platform with these (toolchain defines, as hipcc itself passes them)
             "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
            + inc + ["-I", _build.INC],
            ["-L", tlib, "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch", "-L", _build.PKG, "-l:libbeacon_hip.so",
             "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + tlib])


def signature():
    import torch
    h = hashlib.sha256()
    cf, lf = _flags()
    # (path-independent: the tree is built in one place and used in another -- the flags' directories are left out)
    h.update(repr(([f for f in cf if not os.path.isabs(f)], [f for f in lf if not os.path.isabs(f) and "rpath" not in f],
                   torch.__version__)).encode())
    for f in (SRC, os.path.join(_build.INC, "beacon_hip.h")):
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def stale():
    if not os.path.exists(LIB) or not os.path.exists(LIB + ".sig"):
        return True
    with open(LIB + ".sig") as fh:
        return fh.read().strip() != signature()


def build_ext(force=False, verbose=False):
    """Compile csrc/torch/beacon_torch.cpp with g++ against the torch headers; returns the path, or None without a compiler."""
    if not force and not stale():
        return LIB
    cxx = os.environ.get("CXX") or shutil.which("g++")
    if cxx is None or os.environ.get("BEACON_NO_BUILD") == "1":
        return None
    _build.build_lib()                       # links against libbeacon_hip.so
    cf, lf = _flags()
    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not stale():
                return LIB
            tmp = "%s.tmp%d" % (LIB, os.getpid())
            cmd = [cxx] + cf + [SRC, "-o", tmp] + lf
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            os.replace(tmp, LIB)
            with open(LIB + ".sig", "w") as fh:
                fh.write(signature() + "\n")
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


def load():
    """torch.ops.beacon, or None when the extension cannot be had (BEACON_TORCH_EXT=0, no compiler and no prebuilt library)."""
    global _LOADED
    if _LOADED is not None:
        return _LOADED or None
    _LOADED = False
    if os.environ.get("BEACON_TORCH_EXT", "1") == "0":
        return None
    import torch
    from . import _lib
    _lib.load()                              # libbeacon_hip.so first (RTLD_GLOBAL): the extension resolves bcn_* against it
    path = LIB if not stale() else None
    if path is None:
        try:
            path = build_ext()
        except (OSError, subprocess.CalledProcessError) as e:
            import warnings
            warnings.warn("beacon_amd: the torch extension could not be built (%s); the ctypes binding of the same C ABI is used" % e)
            path = None
    if path is None:
        return None
    try:
        torch.ops.load_library(path)
    except (OSError, RuntimeError) as e:     # a truncated library, one built against another torch or another C ABI
        import warnings
        warnings.warn("beacon_amd: the torch extension %s could not be loaded (%s); the ctypes binding of the same C ABI is used"
                      % (os.path.basename(path), str(e).splitlines()[0] if str(e) else type(e).__name__))
        return None
    _LOADED = torch.ops.beacon
    return _LOADED
