// ubench_pk.hip -- issue interval of packed-float32 VALU instructions on gfx950 (v_pk_add_f32 / v_pk_fma_f32, with and
// without op_sel swaps), next to the plain ones, with 1 and 2 waves per SIMD.  Question: does one v_pk_* cost one issue
// slot (then two rows per lane would halve the plain-instruction count of the Jacobi sweep) or two?
// Build: hipcc --offload-arch=gfx950 -O3 -w scripts/ubench_pk.hip -o scripts/ubench_pk
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
typedef float v2f __attribute__((ext_vector_type(2)));
#define P_FMA   "v_fma_f32 %0, %0, %4, %1\n v_fma_f32 %1, %1, %4, %2\n v_fma_f32 %2, %2, %4, %3\n v_fma_f32 %3, %3, %4, %0\n v_fma_f32 %0, %0, %4, %1\n v_fma_f32 %1, %1, %4, %2\n v_fma_f32 %2, %2, %4, %3\n v_fma_f32 %3, %3, %4, %0\n"
#define P_PKFMA "v_pk_fma_f32 %0, %0, %4, %1\n v_pk_fma_f32 %1, %1, %4, %2\n v_pk_fma_f32 %2, %2, %4, %3\n v_pk_fma_f32 %3, %3, %4, %0\n v_pk_fma_f32 %0, %0, %4, %1\n v_pk_fma_f32 %1, %1, %4, %2\n v_pk_fma_f32 %2, %2, %4, %3\n v_pk_fma_f32 %3, %3, %4, %0\n"
#define P_PKADD "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
#define SW " op_sel:[0,1] op_sel_hi:[1,0]\n"
#define P_PKSWP "v_pk_add_f32 %0, %0, %1" SW "v_pk_add_f32 %1, %1, %2" SW "v_pk_add_f32 %2, %2, %3" SW "v_pk_add_f32 %3, %3, %0" SW "v_pk_add_f32 %0, %0, %1" SW "v_pk_add_f32 %1, %1, %2" SW "v_pk_add_f32 %2, %2, %3" SW "v_pk_add_f32 %3, %3, %0" SW
#define P_PKMUL "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
// dependent chain of packed fma (latency)
#define P_PKDEP "v_pk_fma_f32 %0, %0, %4, %1\n v_pk_fma_f32 %0, %0, %4, %1\n v_pk_fma_f32 %0, %0, %4, %1\n v_pk_fma_f32 %0, %0, %4, %1\n v_pk_fma_f32 %0, %0, %4, %1\n v_pk_fma_f32 %0, %0, %4, %1\n v_pk_fma_f32 %0, %0, %4, %1\n v_pk_fma_f32 %0, %0, %4, %1\n"
#define P_RCP   "v_rcp_f32 %0, %1\n v_rcp_f32 %1, %2\n v_rcp_f32 %2, %3\n v_rcp_f32 %3, %0\n v_rcp_f32 %0, %1\n v_rcp_f32 %1, %2\n v_rcp_f32 %2, %3\n v_rcp_f32 %3, %0\n"
#define P_CND   "v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %2, vcc\n v_cmp_lt_f32 vcc, %2, %4\n v_cndmask_b32 %3, %3, %0, vcc\n v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %2, vcc\n v_cmp_lt_f32 vcc, %2, %4\n v_cndmask_b32 %3, %3, %0, vcc\n"
#define P_MED   "v_med3_f32 %0, %0, %4, %1\n v_med3_f32 %1, %1, %4, %2\n v_med3_f32 %2, %2, %4, %3\n v_med3_f32 %3, %3, %4, %0\n v_med3_f32 %0, %0, %4, %1\n v_med3_f32 %1, %1, %4, %2\n v_med3_f32 %2, %2, %4, %3\n v_med3_f32 %3, %3, %4, %0\n"
#define P_MOV   "v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0\n v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0\n"
template <int MODE>
__global__ __launch_bounds__(1024) void kb(unsigned long long* out, float* sink) {
  v2f a = {1.f, 2.f}, b = {0.5f, 0.25f}, c = {3.f, 1.f}, d = {0.1f, 0.2f};
  const v2f cf = {0.25f + threadIdx.x * 1e-6f, 0.25f};
  __syncthreads();
  const unsigned long long t_0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < 64; it++) {
#define OPS : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(cf)
    if (MODE == 0) asm volatile(REP16(P_PKFMA) OPS);
    if (MODE == 1) asm volatile(REP16(P_PKADD) OPS);
    if (MODE == 2) asm volatile(REP16(P_PKSWP) OPS);
    if (MODE == 3) asm volatile(REP16(P_PKMUL) OPS);
    if (MODE == 4) asm volatile(REP16(P_PKDEP) OPS);
    if (MODE >= 6) { float a0 = a.x, a1 = b.x, a2 = c.x, a3 = d.x, f = cf.x;
      if (MODE == 6) asm volatile(REP16(P_RCP) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(f));
      if (MODE == 7) asm volatile(REP16(P_CND) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(f) : "vcc");
      if (MODE == 8) asm volatile(REP16(P_MED) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(f));
      if (MODE == 9) asm volatile(REP16(P_MOV) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(f));
      a.x = a0; b.x = a1; c.x = a2; d.x = a3; }
    if (MODE == 5) { float a0 = a.x, a1 = b.x, a2 = c.x, a3 = d.x, f = cf.x;
      asm volatile(REP16(P_FMA) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(f)); a.x = a0; b.x = a1; c.x = a2; d.x = a3; }
  }
  const unsigned long long t_1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) { out[2 * (threadIdx.x >> 6)] = t_0; out[2 * (threadIdx.x >> 6) + 1] = t_1; }
  sink[threadIdx.x] = a.x + a.y + b.x + b.y + c.x + c.y + d.x + d.y;
}
template <int MODE> void run(const char* name, int n_instr) {
  unsigned long long* out; float* sink;
  hipMalloc(&out, 2 * 16 * sizeof(unsigned long long)); hipMalloc(&sink, 1024 * sizeof(float));
  for (int nw : {4, 8, 16}) {   // waves per workgroup = per CU: 1 (alone), 4 (one per SIMD), 8 (two per SIMD), 16
    hipLaunchKernelGGL(kb<MODE>, dim3(1), dim3(64 * nw), 0, 0, out, sink);
    hipLaunchKernelGGL(kb<MODE>, dim3(1), dim3(64 * nw), 0, 0, out, sink);
    hipDeviceSynchronize();
    unsigned long long h[32]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    unsigned long long lo = h[0], hi = h[1];   // the oldest wave of a SIMD issues nearly unimpeded: take the whole group
    for (int q = 0; q < nw; q++) { lo = h[2 * q] < lo ? h[2 * q] : lo; hi = h[2 * q + 1] > hi ? h[2 * q + 1] : hi; }
    const double cyc = (double)(hi - lo) / (64.0 * 16 * n_instr);
    const int per_simd = (nw + 3) / 4;
    printf("%-28s %d waves/SIMD: %.2f cycles per instruction per wave = %.2f per instruction per SIMD\n", name, per_simd, cyc, cyc / per_simd);
  }
  hipFree(out); hipFree(sink);
}
int main() {
  run<5>("v_fma_f32 (independent)", 8);
  run<0>("v_pk_fma_f32 (independent)", 8);
  run<1>("v_pk_add_f32", 8);
  run<2>("v_pk_add_f32 op_sel swap", 8);
  run<3>("v_pk_mul_f32", 8);
  run<4>("v_pk_fma_f32 (dependent)", 8);
  run<6>("v_rcp_f32", 8);
  run<7>("v_cmp_lt_f32 + v_cndmask_b32", 8);
  run<8>("v_med3_f32", 8);
  run<9>("v_mov_b32", 8);
  return 0;
}
