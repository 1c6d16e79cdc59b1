// ns2d_fast.hip -- register-resident CDNA4 rayleigh / mixing action step (variant 1).
// Placeholder until the fast path lands: reports "no fast path for this grid".
#include "ns2d.h"

template <typename real> bool ns2d_fast_supported(const NS2DArgs<real>&) { return false; }
template <typename real> int ns2d_launch_fast(const NS2DArgs<real>&, int, hipStream_t) {
  bcn_set_error("no register-resident kernel for this grid");
  return BCN_ERR_UNSUPPORTED;
}
template bool ns2d_fast_supported<float>(const NS2DArgs<float>&);
template bool ns2d_fast_supported<double>(const NS2DArgs<double>&);
template int ns2d_launch_fast<float>(const NS2DArgs<float>&, int, hipStream_t);
template int ns2d_launch_fast<double>(const NS2DArgs<double>&, int, hipStream_t);
