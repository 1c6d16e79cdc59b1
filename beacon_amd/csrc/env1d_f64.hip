// double instantiations of the 1D kernels (env1d_impl.inc); built with -ffp-contract=off (beacon_amd/build.py)
#define BCN_ENV1D_DOUBLE 1
#include "env1d_impl.inc"
