// one-off check of the DPP controls the fast kernel relies on (gfx950)
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CTRL, int RM, int BM, bool BC>
__device__ __forceinline__ float dppf(float oldv, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, oldv), __builtin_bit_cast(int, v), CTRL, RM, BM, BC));
}
__global__ void k(float* out) {
  int l = threadIdx.x;
  float v = (float)l;
  out[l] = dppf<0x138, 0xf, 0xf, false>(-1.f, v);        // wave_shr:1 -> lane l gets l-1, lane 0 gets old
  out[64 + l] = dppf<0x130, 0xf, 0xf, false>(-2.f, v);   // wave_shl:1 -> lane l gets l+1, lane 63 gets old
  float s = 1.0f + l;                                    // sum = 2080
  s += dppf<0x111, 0xf, 0xf, true>(0.f, s);
  s += dppf<0x112, 0xf, 0xf, true>(0.f, s);
  s += dppf<0x114, 0xf, 0xf, true>(0.f, s);
  s += dppf<0x118, 0xf, 0xf, true>(0.f, s);
  out[128 + l] = s;                                      // lane 15 of each row: row total
  float t = s;
  t += dppf<0x142, 0xa, 0xf, false>(0.f, t);             // row_bcast:15 rows 1,3
  t += dppf<0x143, 0xc, 0xf, false>(0.f, t);             // row_bcast:31 rows 2,3
  out[192 + l] = t;                                      // lane 63 = wave total
  out[256 + l] = __builtin_amdgcn_readlane(__builtin_bit_cast(int, t), 63) == __builtin_bit_cast(int, t) ? 1.f : 0.f;
}
int main() {
  float* d; hipMalloc(&d, 320 * 4); hipLaunchKernelGGL(k, 1, 64, 0, 0, d);
  float h[320]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("shr: %g %g %g ... %g\n", h[0], h[1], h[2], h[63]);
  printf("shl: %g %g ... %g %g\n", h[64], h[65], h[126], h[127]);
  printf("rowsum lane15,31,47,63: %g %g %g %g (expect 136 392 648 904)\n", h[128+15], h[128+31], h[128+47], h[128+63]);
  printf("total lane63: %g (expect 2080)\n", h[192+63]);
  return 0;
}
