"""ctypes binding of libbeacon_hip.so (include/beacon_hip.h).

The product path has NO CPU fallback: if the HIP library cannot be built/loaded, or no
GPU is visible, every env constructor raises."""
import ctypes as C
import os

from . import build as _build

c_i32p = C.POINTER(C.c_int32)
c_u8p = C.POINTER(C.c_uint8)
vp = C.c_void_p

F32, F64 = 0, 1
ST_OK, ST_ITMAX, ST_BLOWUP, ST_PLAN = 0, 1, 2, 4
API_VERSION, COUNTER_WORDS = 4, 4          # include/beacon_hip.h: BCN_API_VERSION, BCN_COUNTER_WORDS


class RayleighCfg(C.Structure):
    _fields_ = [("nx", C.c_int32), ("ny", C.c_int32), ("ndt_act", C.c_int32), ("n_act", C.c_int32),
                ("n_sgts", C.c_int32), ("nx_sgts", C.c_int32), ("nx_obs_pts", C.c_int32),
                ("ny_obs_pts", C.c_int32), ("nx_obs", C.c_int32), ("ny_obs", C.c_int32),
                ("n_obs_steps", C.c_int32), ("itmax", C.c_int32),
                ("dx", C.c_double), ("dy", C.c_double), ("dt", C.c_double),
                ("pr", C.c_double), ("ra", C.c_double), ("Tc", C.c_double), ("Th", C.c_double),
                ("C", C.c_double), ("tol", C.c_double)]


class MixingCfg(C.Structure):
    _fields_ = [("nx", C.c_int32), ("ny", C.c_int32), ("ndt_act", C.c_int32), ("n_act", C.c_int32),
                ("nx_obs_pts", C.c_int32), ("ny_obs_pts", C.c_int32), ("nx_obs", C.c_int32),
                ("ny_obs", C.c_int32), ("n_obs_steps", C.c_int32), ("itmax", C.c_int32),
                ("i_min", C.c_int32), ("i_max", C.c_int32), ("j_min", C.c_int32), ("j_max", C.c_int32),
                ("dx", C.c_double), ("dy", C.c_double), ("dt", C.c_double),
                ("re", C.c_double), ("pe", C.c_double), ("u_max", C.c_double), ("C0", C.c_double),
                ("ref_c", C.c_double), ("tol", C.c_double)]


class BurgersCfg(C.Structure):
    _fields_ = [("nx", C.c_int32), ("ndt_act", C.c_int32), ("n_act", C.c_int32), ("ctrl_pos", C.c_int32),
                ("n_obs_pts", C.c_int32),
                ("dx", C.c_double), ("dt", C.c_double), ("amp", C.c_double), ("u_target", C.c_double)]


class ShkadovCfg(C.Structure):
    _fields_ = [("nx", C.c_int32), ("ndt_act", C.c_int32), ("n_act", C.c_int32), ("n_jets", C.c_int32),
                ("jet_pos", C.c_int32), ("jet_hw", C.c_int32), ("jet_space", C.c_int32),
                ("l_obs", C.c_int32), ("l_rwd", C.c_int32), ("n_obs", C.c_int32), ("obs_stride", C.c_int32),
                ("n_interp", C.c_int32),
                ("dx", C.c_double), ("dt", C.c_double), ("delta", C.c_double), ("jet_amp", C.c_double),
                ("eps", C.c_double), ("h_blow", C.c_double), ("blowup_rwd", C.c_double)]


class SloshingCfg(C.Structure):
    _fields_ = [("nx", C.c_int32), ("ndt_act", C.c_int32), ("n_act", C.c_int32), ("n_interp", C.c_int32),
                ("dx", C.c_double), ("dt", C.c_double), ("g", C.c_double), ("amp", C.c_double),
                ("alpha", C.c_double)]


# every symbol include/beacon_hip.h declares: (restype, argtypes)
SIGNATURES = {
    "bcn_rayleigh_create": (C.c_int, [C.POINTER(RayleighCfg), C.c_int, C.c_int, C.c_int, C.POINTER(vp)]),
    "bcn_rayleigh_reset": (C.c_int, [vp, vp, vp, vp]),
    "bcn_rayleigh_step": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "bcn_mixing_create": (C.c_int, [C.POINTER(MixingCfg), C.c_int, C.c_int, C.c_int, C.POINTER(vp)]),
    "bcn_mixing_reset": (C.c_int, [vp, vp, vp]),
    "bcn_mixing_step": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "bcn_burgers_create": (C.c_int, [C.POINTER(BurgersCfg), C.c_int, C.c_int, C.c_int, C.POINTER(vp)]),
    "bcn_burgers_reset": (C.c_int, [vp, vp, vp]),
    "bcn_burgers_step": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "bcn_shkadov_create": (C.c_int, [C.POINTER(ShkadovCfg), C.c_int, C.c_int, C.c_int, C.POINTER(vp)]),
    "bcn_shkadov_reset": (C.c_int, [vp, vp, vp, vp]),
    "bcn_shkadov_step": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "bcn_sloshing_create": (C.c_int, [C.POINTER(SloshingCfg), C.c_int, C.c_int, C.c_int, C.POINTER(vp)]),
    "bcn_sloshing_reset": (C.c_int, [vp, vp, vp, vp]),
    "bcn_sloshing_step": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp]),
    "bcn_env_kind": (C.c_int, [vp]),
    "bcn_batch": (C.c_int, [vp]),
    "bcn_dtype": (C.c_int, [vp]),
    "bcn_n_obs": (C.c_int, [vp]),
    "bcn_n_act": (C.c_int, [vp]),
    "bcn_ndt_act": (C.c_int, [vp]),
    "bcn_device": (C.c_int, [vp]),
    "bcn_state_elems": (C.c_size_t, [vp]),
    "bcn_get_state": (C.c_int, [vp, vp, C.c_int, vp]),
    "bcn_set_state": (C.c_int, [vp, vp, C.c_int, vp]),
    "bcn_set_mask": (C.c_int, [vp, vp]),
    "bcn_get_stp": (C.c_int, [vp, c_i32p, vp]),
    "bcn_set_stp": (C.c_int, [vp, c_i32p, vp]),
    "bcn_set_variant": (C.c_int, [vp, C.c_int]),
    "bcn_get_counters": (C.c_int, [vp, C.POINTER(C.c_uint64), vp]),
    "bcn_get_counters_n": (C.c_int, [vp, C.POINTER(C.c_uint64), C.c_int, vp]),
    "bcn_api_version": (C.c_int, []),
    "bcn_set_fast_plugin": (C.c_int, [vp, vp, C.c_size_t]),
    "bcn_set_option": (C.c_int, [vp, C.c_char_p, C.c_int]),
    "bcn_set_noise": (C.c_int, [vp, C.c_double, C.c_uint64, C.c_int64]),
    "bcn_set_sched": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    "bcn_set_slow_mode_bound": (C.c_int, [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "bcn_get_slow_mode_bound": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "bcn_kernel_name": (C.c_char_p, [vp]),
    "bcn_destroy": (C.c_int, [vp]),
    "bcn_last_error": (C.c_char_p, []),
    "bcn_version": (C.c_char_p, []),
}

_LIB = None


def lib_path():
    return _build.LIB


def load():
    """dlopen the in-tree library (building it first when stale and hipcc is present)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = _build.LIB
    if _build.stale():
        # BEACON_NO_BUILD=1 (exported by the profiling scripts): never spawn a compiler from this process --
        # under rocprofv3 the children would inherit the profiler's preload (scripts/prof_bench.sh).
        # A stale binary is never dlopen'ed: its cfg structs / signatures may disagree with this source tree.
        if os.environ.get("BEACON_NO_BUILD") == "1":
            raise RuntimeError("libbeacon_hip.so is missing or stale and BEACON_NO_BUILD=1: run "
                               "`python -c 'import __graft_entry__ as g; g.build()'` first")
        if _build.hipcc() is None:
            raise RuntimeError("libbeacon_hip.so is missing or older than its sources and hipcc is not "
                               "available to build it; beacon_amd has no CPU fallback")
        path = _build.build_lib()
    L = C.CDLL(path, mode=C.RTLD_GLOBAL)   # the on-demand kernel plugins resolve bcn_set_error against it
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if L.bcn_api_version() != API_VERSION:
        raise RuntimeError("libbeacon_hip.so has API version %d, this binding expects %d" % (L.bcn_api_version(), API_VERSION))
    _LIB = L
    return L


class BeaconHipError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise BeaconHipError("libbeacon_hip: error %d: %s" % (rc, load().bcn_last_error().decode()))
