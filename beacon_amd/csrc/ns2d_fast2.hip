// ns2d_fast2.hip -- the two-rows-per-lane kernels (ns2d_fast2_impl.h) instantiated for the grid built into the library.
#include "ns2d_fast2_impl.h"

template <typename real>
bool ns2d_fast2_supported(const NS2DArgs<real>& a) {
  return a.nx == 100 && a.ny == 100 && (a.kind == 1 || (sizeof(real) == 4 && a.n_sgts <= 64));   // float64: mixing only
}

template <typename real>
int ns2d_launch_fast2(const NS2DArgs<real>& a, int batch, hipStream_t s) {
  if constexpr (std::is_same<real, float>::value) {
    if (a.nx == 100 && a.ny == 100) {
      if (a.kind == 1) return launch_fast2<float, 100, 100, 13, 1>(a, batch, s);
      return launch_fast2<float, 100, 100, 13, 0>(a, batch, s);
    }
  }
  if constexpr (std::is_same<real, double>::value) {   // the reference's arithmetic: fields in global scratch (GF = 1)
    if (a.nx == 100 && a.ny == 100 && a.kind == 1) return launch_fast2<double, 100, 100, 13, 1, 1>(a, batch, s);
  }
  bcn_set_error("no two-rows-per-lane kernel for this grid");
  return BCN_ERR_UNSUPPORTED;
}

template <typename real>
size_t ns2d_fast2_scratch_elems(const NS2DArgs<real>& a) {
  if (sizeof(real) == 8 && a.nx == 100 && a.ny == 100 && a.kind == 1) return Fast2Geom<100, 100, 13, 1>::scratch_elems();
  return 0;
}
template size_t ns2d_fast2_scratch_elems<float>(const NS2DArgs<float>&);
template size_t ns2d_fast2_scratch_elems<double>(const NS2DArgs<double>&);
template bool ns2d_fast2_supported<float>(const NS2DArgs<float>&);
template bool ns2d_fast2_supported<double>(const NS2DArgs<double>&);
template int ns2d_launch_fast2<float>(const NS2DArgs<float>&, int, hipStream_t);
template int ns2d_launch_fast2<double>(const NS2DArgs<double>&, int, hipStream_t);
