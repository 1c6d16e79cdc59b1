// aggregate VALU throughput per SIMD with 1/2/4 waves per SIMD (gfx950): every wave stamps its own
// start/end; report (max end - min start) / (instructions per wave) = cycles per wave-instruction per wave slot
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CTRL, bool BC>
__device__ __forceinline__ float dppf(float oldv, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, oldv), __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, BC));
}
typedef float float2_ __attribute__((ext_vector_type(2)));
#define N 512
template <int MODE>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, float* sink) {
  float a = threadIdx.x * 0.001f, b = a + 1.f, c = a + 2.f, d = a + 3.f;
  float2_ pa = {a, b}, pb = {c, d}, pc = {a + 1, b + 1}, pd = {c + 1, d + 1};
  const float2_ m = {1.0001f, 1.0001f}, q = {0.1f, 0.1f};
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < N; it++) {
    if (MODE == 0) { a = fmaf(a, 1.0001f, 0.1f); b = fmaf(b, 1.0001f, 0.1f); c = fmaf(c, 1.0001f, 0.1f); d = fmaf(d, 1.0001f, 0.1f);
                     a = fmaf(a, 1.0001f, 0.1f); b = fmaf(b, 1.0001f, 0.1f); c = fmaf(c, 1.0001f, 0.1f); d = fmaf(d, 1.0001f, 0.1f); }
    if (MODE == 1) { pa = __builtin_elementwise_fma(pa, m, q); pb = __builtin_elementwise_fma(pb, m, q); pc = __builtin_elementwise_fma(pc, m, q); pd = __builtin_elementwise_fma(pd, m, q);
                     pa = __builtin_elementwise_fma(pa, m, q); pb = __builtin_elementwise_fma(pb, m, q); pc = __builtin_elementwise_fma(pc, m, q); pd = __builtin_elementwise_fma(pd, m, q); }
    if (MODE == 2) { a += dppf<0x138, true>(0.f, b); b += dppf<0x138, true>(0.f, c); c += dppf<0x138, true>(0.f, d); d += dppf<0x138, true>(0.f, a);
                     a += dppf<0x130, true>(0.f, c); b += dppf<0x130, true>(0.f, d); c += dppf<0x130, true>(0.f, a); d += dppf<0x130, true>(0.f, b); }
    if (MODE == 3) { a = a + b; b = b + c; c = c + d; d = d + a; a = a * c; b = b * d; c = c - a; d = d - b; }
    if (MODE == 4) { a += dppf<0x111, true>(0.f, b); b += dppf<0x111, true>(0.f, c); c += dppf<0x111, true>(0.f, d); d += dppf<0x111, true>(0.f, a);
                     a += dppf<0x112, true>(0.f, c); b += dppf<0x112, true>(0.f, d); c += dppf<0x112, true>(0.f, a); d += dppf<0x112, true>(0.f, b); }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) { out[2 * (threadIdx.x >> 6)] = t0; out[2 * (threadIdx.x >> 6) + 1] = t1; }
  sink[threadIdx.x] = a + b + c + d + pa.x + pa.y + pb.x + pb.y + pc.x + pc.y + pd.x + pd.y;
}
template <int MODE> void run(const char* name, unsigned long long* d, float* s) {
  for (int nt : {256, 512, 1024}) {
    hipLaunchKernelGGL(k<MODE>, 1, nt, 0, 0, d, s);
    unsigned long long h[32]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    unsigned long long mn = ~0ull, mx = 0, w0 = h[1] - h[0];
    for (int w = 0; w < nt / 64; w++) { if (h[2 * w] < mn) mn = h[2 * w]; if (h[2 * w + 1] > mx) mx = h[2 * w + 1]; }
    printf("%-10s waves/SIMD %d: wave0 %.2f cyc/instr, all waves done %.2f cyc per wave-instr slot (=> %.2f cyc per instr per SIMD)\n", name, nt / 256,
           (double)w0 / (8.0 * N), (double)(mx - mn) / (8.0 * N), (double)(mx - mn) / (8.0 * N) / (nt / 256));
  }
}
int main() {
  unsigned long long* d; float* s; hipMalloc(&d, 64 * 8); hipMalloc(&s, 1024 * 4);
  run<0>("fma", d, s); run<1>("pk_fma", d, s); run<2>("dpp_wave", d, s); run<3>("add/mul", d, s); run<4>("dpp_row", d, s);
  return 0;
}
