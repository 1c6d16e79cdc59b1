// ns2d_fast_f64.hip -- the float64 instantiations of ns2d_fast.hip as a translation unit of their own (own compiler flags:
// beacon_amd/build.py FILE_FLAGS).
#define BCN_FAST_TU_F64 1
#include "ns2d_fast.hip"
