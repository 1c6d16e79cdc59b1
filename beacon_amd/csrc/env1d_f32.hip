// float instantiations of the 1D kernels (env1d_impl.inc)
#define BCN_ENV1D_FLOAT 1
#include "env1d_impl.inc"
