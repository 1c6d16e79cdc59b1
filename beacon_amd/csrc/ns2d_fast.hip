// ns2d_fast.hip -- the register-resident kernels (ns2d_fast_impl.h) instantiated for the grids built into the library:
// the metric grid 128x64 (float32 / float64), the reference's default 50x50 and its other natural aspect ratios.
// Every other grid gets its own instantiation at run time (ns2d_jit.hip, beacon_amd/jit.py).
#include "ns2d_fast_impl.h"

namespace {

template <typename real>
int fast_config(const NS2DArgs<real>& a) {
  if (a.kind != 0 || a.n_sgts > 64) return 0;
  if (a.nx == 128 && a.ny == 64 && sizeof(real) == 4) return 1;
  if (a.nx == 50 && a.ny == 50) return 2;
  if (a.nx == 128 && a.ny == 64 && sizeof(real) == 8) return 3;   // fields in global scratch
  // the reference's other natural aspect ratios (nx = 50 L, ny = 50 H with H = 1), float32
  if (a.nx == 100 && a.ny == 50 && sizeof(real) == 4) return 4;
  if (a.nx == 150 && a.ny == 50 && sizeof(real) == 4) return 5;
  if (a.nx == 200 && a.ny == 50 && sizeof(real) == 4) return 6;
  return 0;
}

}  // namespace

template <typename real>
bool ns2d_fast_supported(const NS2DArgs<real>& a) { return fast_config<real>(a) != 0 || ns2d_fast2_supported<real>(a); }

template <typename real>
int ns2d_launch_fast(const NS2DArgs<real>& a, int batch, hipStream_t s) {
  if (ns2d_fast2_supported<real>(a)) return ns2d_launch_fast2<real>(a, batch, s);
  switch (fast_config<real>(a)) {
    case 1:
      if constexpr (std::is_same<real, float>::value) return launch_fast<float, 128, 64, BCN_R128, 0>(a, batch, s);
      break;
    case 2: return launch_fast<real, 50, 50, BCN_R50, 0>(a, batch, s);
    case 3:
      if constexpr (std::is_same<real, double>::value) return launch_fast<double, 128, 64, BCN_R128D, 0, BCN_GFD>(a, batch, s);
      break;
    case 4:
      if constexpr (std::is_same<real, float>::value) return launch_fast<float, 100, 50, 10, 0>(a, batch, s);
      break;
    case 5:
      if constexpr (std::is_same<real, float>::value) return launch_fast<float, 150, 50, 15, 0>(a, batch, s);
      break;
    case 6:
      if constexpr (std::is_same<real, float>::value) return launch_fast<float, 200, 50, 25, 0>(a, batch, s);
      break;
    default: break;
  }
  bcn_set_error("no register-resident kernel for this grid");
  return BCN_ERR_UNSUPPORTED;
}

template <typename real>
size_t ns2d_fast_scratch_elems(const NS2DArgs<real>& a) {
  if (ns2d_fast2_supported<real>(a)) return ns2d_fast2_scratch_elems<real>(a);
  if (fast_config<real>(a) == 3) return FastGeom<128, 64, BCN_R128D, BCN_GFD>::scratch_elems();
  return 0;
}
// One translation unit per precision (ns2d_fast_f64.hip includes this file with BCN_FAST_TU_F64): the two are built with different
// optimisation levels -- the float32 kernels gain 1 % from -O2, the float64 ones lose 0.7 % (beacon_amd/build.py, round 6)
#ifndef BCN_FAST_TU_F64
template size_t ns2d_fast_scratch_elems<float>(const NS2DArgs<float>&);
template bool ns2d_fast_supported<float>(const NS2DArgs<float>&);
template int ns2d_launch_fast<float>(const NS2DArgs<float>&, int, hipStream_t);
#else
template size_t ns2d_fast_scratch_elems<double>(const NS2DArgs<double>&);
template bool ns2d_fast_supported<double>(const NS2DArgs<double>&);
template int ns2d_launch_fast<double>(const NS2DArgs<double>&, int, hipStream_t);
#endif
