// dpp_select_repro.hip -- a pitfall of cross-lane reads, met in round 5 (ROCm 7.2 hipcc, gfx950, -O3): a row maximum written
//   s = s > dpp(s) ? s : dpp(s)          (dpp(): a DPP row shift of s, bound_ctrl: lanes without a source read 0)
// comes out 0 in lane 15 of the row.  The second dpp() stands in one arm of a conditional expression; hipcc compiles that arm as
// an exec-masked block (s_and_saveexec ... v_mov_b32_dpp ... s_or exec), and a DPP read of a lane that is switched off is an
// invalid source: it returns 0 (bound_ctrl) or the old value.  Legal code generation for the source as written -- the shift
// must be evaluated IN FRONT of anything lane-dependent (one dpp() into a temporary, or fmaxf).  The same reduction with fmaxf,
// and the same shifts with +, are right.
//   hipcc -O3 --offload-arch=gfx950 -I include scripts/dpp_select_repro.hip -o /tmp/dpp_select_repro && /tmp/dpp_select_repro
// beacon_amd/csrc/bcn_dpp.h spells its DPP max reductions with fmax (row16_max, wave_max_lane63).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../beacon_amd/csrc/bcn_dpp.h"
using namespace bcn_dpp;
__device__ float row_max_select(float s) {
#define STEP(C) s = s > dpp<C, 0xf, 0xf, true>(0.f, s) ? s : dpp<C, 0xf, 0xf, true>(0.f, s);   /* the shift written twice, as first written in bcn_dpp.h */
  STEP(0x111) STEP(0x112) STEP(0x114) STEP(0x118)
#undef STEP
  return s;
}
__global__ void k(float* out) {
  const int lane = threadIdx.x & 63;
  const float v = 0.004f + 0.0001f * ((lane * 7) % 13);
  out[lane] = row_max_select(v);
  out[64 + lane] = row16_max<float>(v);
  out[128 + lane] = row16_sum<float>(v);
}
int main() {
  float *d, h[192];
  (void)hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  float want = 0.f, sum = 0.f;
  for (int l = 0; l < 16; l++) { const float v = 0.004f + 0.0001f * ((l * 7) % 13); want = v > want ? v : want; sum += v; }
  printf("row 0: maximum %g, sum %g\n  lane 15 with the select    : %g  %s\n  lane 15 with fmaxf          : %g  %s\n  lane 15 of the sum          : %g\n",
         want, sum, h[15], h[15] == want ? "ok" : "WRONG", h[64 + 15], h[64 + 15] == want ? "ok" : "WRONG", h[128 + 15]);
  return 0;
}
