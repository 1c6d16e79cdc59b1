// micro-benchmarks for the Jacobi sweep building blocks (gfx950): cycles per op via s_memtime
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CTRL, bool BC>
__device__ __forceinline__ float dppf(float oldv, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, oldv), __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, BC));
}
#define TIME(NAME, ...)                                                        \
  {                                                                            \
    __syncthreads();                                                           \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                      \
    for (int it = 0; it < 256; it++) { __VA_ARGS__ }                                  \
    __builtin_amdgcn_s_waitcnt(0);                                             \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                      \
    if (threadIdx.x == 0 && blockIdx.x == 0) out[idx] = (float)(t1 - t0) / 256.0f; \
    idx++;                                                                     \
  }
__global__ __launch_bounds__(1024) void k(float* out, float* sink, int nw) {
  __shared__ float lds[4096];
  int idx = 0;
  float a = threadIdx.x * 0.001f, b = a + 1.f, c = a + 2.f, d = a + 3.f;
  // 8 independent fma
  TIME("fma8", a = fmaf(a, 1.0001f, 0.1f); b = fmaf(b, 1.0001f, 0.1f); c = fmaf(c, 1.0001f, 0.1f); d = fmaf(d, 1.0001f, 0.1f);
       a = fmaf(a, 1.0001f, 0.1f); b = fmaf(b, 1.0001f, 0.1f); c = fmaf(c, 1.0001f, 0.1f); d = fmaf(d, 1.0001f, 0.1f);)
  // 8 wave_shr dpp (independent)
  TIME("wave_shr8", a += dppf<0x138, true>(0.f, b); b += dppf<0x138, true>(0.f, c); c += dppf<0x138, true>(0.f, d); d += dppf<0x138, true>(0.f, a);
       a += dppf<0x138, true>(0.f, c); b += dppf<0x138, true>(0.f, d); c += dppf<0x138, true>(0.f, a); d += dppf<0x138, true>(0.f, b);)
  TIME("row_shr8", a += dppf<0x111, true>(0.f, b); b += dppf<0x111, true>(0.f, c); c += dppf<0x111, true>(0.f, d); d += dppf<0x111, true>(0.f, a);
       a += dppf<0x111, true>(0.f, c); b += dppf<0x111, true>(0.f, d); c += dppf<0x111, true>(0.f, a); d += dppf<0x111, true>(0.f, b);)
  TIME("wave_shl8", a += dppf<0x130, true>(0.f, b); b += dppf<0x130, true>(0.f, c); c += dppf<0x130, true>(0.f, d); d += dppf<0x130, true>(0.f, a);
       a += dppf<0x130, true>(0.f, c); b += dppf<0x130, true>(0.f, d); c += dppf<0x130, true>(0.f, a); d += dppf<0x130, true>(0.f, b);)
  {
    typedef float float2_ __attribute__((ext_vector_type(2)));
    float2_ pa = {a, b}, pb = {c, d}, pc = {a + 1, b + 1}, pd = {c + 1, d + 1};
    const float2_ m = {1.0001f, 1.0001f}, q = {0.1f, 0.1f};
    TIME("pkfma8", pa = __builtin_elementwise_fma(pa, m, q); pb = __builtin_elementwise_fma(pb, m, q); pc = __builtin_elementwise_fma(pc, m, q); pd = __builtin_elementwise_fma(pd, m, q);
         pa = __builtin_elementwise_fma(pa, m, q); pb = __builtin_elementwise_fma(pb, m, q); pc = __builtin_elementwise_fma(pc, m, q); pd = __builtin_elementwise_fma(pd, m, q);)
    a += pa.x + pa.y + pc.x; b += pb.x + pb.y + pd.y;
  }
  // 8 dependent fma (one chain)
  TIME("fmadep8", a = fmaf(a, 1.0001f, 0.1f); a = fmaf(a, 1.0001f, 0.1f); a = fmaf(a, 1.0001f, 0.1f); a = fmaf(a, 1.0001f, 0.1f);
       a = fmaf(a, 1.0001f, 0.1f); a = fmaf(a, 1.0001f, 0.1f); a = fmaf(a, 1.0001f, 0.1f); a = fmaf(a, 1.0001f, 0.1f);)
  // 8 dependent dpp adds
  TIME("dppdep8", a += dppf<0x138, true>(0.f, a); a += dppf<0x130, true>(0.f, a); a += dppf<0x138, true>(0.f, a); a += dppf<0x130, true>(0.f, a);
       a += dppf<0x138, true>(0.f, a); a += dppf<0x130, true>(0.f, a); a += dppf<0x138, true>(0.f, a); a += dppf<0x130, true>(0.f, a);)
  // barrier only
  TIME("barrier", __syncthreads();)
  // lds write + barrier + read
  TIME("lds_xchg", lds[threadIdx.x] = a; __syncthreads(); a += lds[(threadIdx.x + 64) & 1023];)
  // shfl (ds_bpermute) x2
  TIME("shfl2", a += __shfl_up(b, 1, 64); b += __shfl_down(a, 1, 64);)
  sink[threadIdx.x + blockIdx.x * blockDim.x] = a + b + c + d;
}
int main() {
  float *d, *s; hipMalloc(&d, 64 * 4); hipMalloc(&s, 1024 * 512 * 4);
  const char* names[] = {"fma8", "wave_shr8", "row_shr8", "wave_shl8", "pkfma8", "fmadep8", "dppdep8", "barrier", "lds_xchg", "shfl2"};
  for (int nt : {64, 256, 1024}) for (int grid : {1, 512}) {
    hipLaunchKernelGGL(k, grid, nt, 0, 0, d, s, nt / 64);
    float h[16]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("threads %4d grid %3d:", nt, grid);
    for (int i = 0; i < 10; i++) printf(" %s=%.1f", names[i], h[i]);
    printf("  (cycles per loop body, s_memtime)\n");
  }
  return 0;
}
