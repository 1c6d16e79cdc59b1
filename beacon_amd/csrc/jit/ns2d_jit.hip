// ns2d_jit.hip -- ONE instantiation of the register-resident 2D kernels for a grid that is not built into
// libbeacon_hip.so, compiled on demand by beacon_amd/jit.py into its own small shared object:
//   hipcc ... -DBCN_JIT_ROWS=1|2|4 -DBCN_JIT_REAL=float|double -DBCN_JIT_NX=.. -DBCN_JIT_NY=.. -DBCN_JIT_R=.. -DBCN_JIT_KIND=0|1
//            [-DBCN_JIT_GF=0|1|2] [-DBCN_JIT_RPL=2|3|4]
// (ROWS: 1 = ns2d_fast_impl.h, one row per lane, ny <= 64; 2 = ns2d_fast2_impl.h, two rows per lane, 64 < ny <= 128;
//  4 = ns2d_fast4_impl.h, Poisson solve in registers with BCN_JIT_RPL rows per lane, 128 < ny <= 256.)
// The reference takes any L, H (rayleigh.py:20-27: nx = 50 L, ny = 50 H; mixing.py:20-28: 100 L, 100 H); the library
// hands the argument block of a step to bcn_jit_launch through bcn_set_fast_plugin (include/beacon_hip.h).
#if BCN_JIT_ROWS == 1 && BCN_JIT_KIND != 0
#error "ns2d_fast_impl.h (one row per lane, ny <= 64) implements the rayleigh boundary conditions only: mixing needs BCN_JIT_ROWS=2"
#endif
#if BCN_JIT_ROWS == 1
#include "../ns2d_fast_impl.h"
#elif BCN_JIT_ROWS == 2
#include "../ns2d_fast2_impl.h"
#else
#include "../ns2d_fast4_impl.h"
#endif

#ifndef BCN_JIT_GF
#define BCN_JIT_GF 0
#endif

extern "C" {

__attribute__((visibility("default"))) int bcn_jit_launch(const void* args, int batch, void* stream) {
#ifdef BCN_JIT_BREAK   // TEST HOOK (tests/test_gpu_parity.py): a deliberately wrong kernel -- the first-use self-check of beacon_amd/jit.py must refuse it
  NS2DArgs<BCN_JIT_REAL> a = *static_cast<const NS2DArgs<BCN_JIT_REAL>*>(args);
  a.dt *= BCN_JIT_REAL(1.5);
#else
  const NS2DArgs<BCN_JIT_REAL>& a = *static_cast<const NS2DArgs<BCN_JIT_REAL>*>(args);
#endif
  if (a.nx != BCN_JIT_NX || a.ny != BCN_JIT_NY || a.kind != BCN_JIT_KIND) {
    bcn_set_error("kernel plugin built for %dx%d kind %d, handle is %dx%d kind %d", BCN_JIT_NX, BCN_JIT_NY, BCN_JIT_KIND,
                  a.nx, a.ny, a.kind);
    return BCN_ERR_ARG;
  }
#if BCN_JIT_ROWS == 1
  return launch_fast<BCN_JIT_REAL, BCN_JIT_NX, BCN_JIT_NY, BCN_JIT_R, BCN_JIT_KIND, BCN_JIT_GF>(a, batch, static_cast<hipStream_t>(stream));
#elif BCN_JIT_ROWS == 2
  return launch_fast2<BCN_JIT_REAL, BCN_JIT_NX, BCN_JIT_NY, BCN_JIT_R, BCN_JIT_KIND, BCN_JIT_GF>(a, batch, static_cast<hipStream_t>(stream));
#else
  return launch_fast4<BCN_JIT_REAL, BCN_JIT_NX, BCN_JIT_NY, BCN_JIT_R, BCN_JIT_RPL, BCN_JIT_KIND>(a, batch, static_cast<hipStream_t>(stream));
#endif
}

// elements of per-workgroup field scratch the kernel needs (0: its fields live in LDS)
__attribute__((visibility("default"))) size_t bcn_jit_scratch_elems(void) {
#if BCN_JIT_ROWS == 1
  return FastGeom<BCN_JIT_NX, BCN_JIT_NY, BCN_JIT_R, BCN_JIT_GF>::scratch_elems();
#elif BCN_JIT_ROWS == 2
  return Fast2Geom<BCN_JIT_NX, BCN_JIT_NY, BCN_JIT_R, BCN_JIT_GF>::scratch_elems();
#else
  return 0;   // the fields stay where the generic kernel keeps them
#endif
}

__attribute__((visibility("default"))) size_t bcn_jit_lds_bytes(void) {
#if BCN_JIT_ROWS == 1
  return FastGeom<BCN_JIT_NX, BCN_JIT_NY, BCN_JIT_R, BCN_JIT_GF>::lds_bytes<BCN_JIT_REAL>();
#elif BCN_JIT_ROWS == 2
  return Fast2Geom<BCN_JIT_NX, BCN_JIT_NY, BCN_JIT_R, BCN_JIT_GF>::lds_elems() * sizeof(BCN_JIT_REAL);
#else
  return (size_t)Fast4Geom<BCN_JIT_NX, BCN_JIT_NY, BCN_JIT_R, BCN_JIT_RPL>::lds_elems(sizeof(BCN_JIT_REAL)) * sizeof(BCN_JIT_REAL);
#endif
}

}  // extern "C"
