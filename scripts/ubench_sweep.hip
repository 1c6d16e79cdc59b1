// ubench_sweep.hip -- Jacobi sweep mappings for a 128x64 float32 grid on one CU (8 waves), isolated from the
// rest of the kernel: A = lanes along y, 16 columns per lane (ns2d_fast.hip today: 2 DPP adds per cell);
// B = lanes along x with 2 columns per lane, 8 rows per wave (1 DPP add per cell, y-neighbours in registers,
// two halo rows per wave through LDS).  Same arithmetic per cell, same reduction, one barrier per sweep.
// Build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -ffp-contract=on scripts/ubench_sweep.hip -o scripts/ubench_sweep
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__device__ __forceinline__ float add2dpp(float acc, float c) {   // acc + c(lane+1) + c(lane-1), 0 outside
  float t, r;
  asm("s_nop 1\n\t"
      "v_add_f32_dpp %0, %2, %3 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %1, %2, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
      : "=&v"(t), "=v"(r) : "v"(c), "v"(acc));
  return r;
}
__device__ __forceinline__ float add_shl(float acc, float c) {   // acc + c(lane+1)
  float r;
  asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(c), "v"(acc));
  return r;
}
__device__ __forceinline__ float add_shr(float acc, float c) {   // acc + c(lane-1)
  float r;
  asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(c), "v"(acc));
  return r;
}
template <int CTRL, int RM>
__device__ __forceinline__ float dppf(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, RM, 0xf, CTRL < 0x140));
}
__device__ __forceinline__ float wave_sum63(float s) {
  s += dppf<0x111, 0xf>(s); s += dppf<0x112, 0xf>(s); s += dppf<0x114, 0xf>(s); s += dppf<0x118, 0xf>(s);
  s += dppf<0x142, 0xa>(s); s += dppf<0x143, 0xc>(s);
  return s;
}

// cell variants of mapping A: 0 = fused DPP pair behind one s_nop (the kernel), 1 = compiler-scheduled DPP builtins,
// 2 = both neighbours by v_mov_dpp, then plain adds, 3 = (n + e) + (s + w) with two independent fused DPP adds
template <int VAR>
__device__ __forceinline__ float nsum(float e, float w, float c) {
  if (VAR == 0 || VAR >= 4) return add2dpp(e + w, c);
  if (VAR == 1) return (e + w) + dppf<0x130, 0xf>(c) + dppf<0x138, 0xf>(c);
  if (VAR == 2) { const float n = dppf<0x130, 0xf>(c), s = dppf<0x138, 0xf>(c); return (e + w) + (n + s); }
  return add_shl(e, c) + add_shr(w, c);
}

// ---- A: lanes along y ------------------------------------------------------------------------
template <int VAR, int R = 16, int NW = 8>
__global__ __launch_bounds__(NW * 64) void sweepA(float* out, const float* in, int nsweep, float cx) {
  __shared__ float ex[2][NW][2][64];
  __shared__ __attribute__((aligned(16))) float errp[2][16];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  float A[R], B[R], nb[R];
  for (int k = 0; k < R; k++) { A[k] = 0; nb[k] = in[((blockIdx.x * 1024 + tid) * 16 + k) % (256 * 512 * 16)]; }
  const float cB = (lane == 0 || lane == 63) ? cx : 0.f, wl = 1.f + (lane == 0) + (lane == 63);
  const int wm = w > 0 ? w - 1 : 0, wp = w < NW - 1 ? w + 1 : NW - 1;
  float hW = 0, hE = 0, hWr = 0, hEr = 0, e8[NW], errsum = 0;
  for (int q = 0; q < NW; q++) e8[q] = 0;
  int xb = 0;
#define CELLA(c, e, wv, nbk) (cx * nsum<VAR>(e, wv, c) + (cB * (c) + (nbk)))
#define SWEEPA(S, D)                                                                     \
  {                                                                                      \
    float acc = 0, accb = 0;                                                             \
    _Pragma("unroll") for (int k = 1; k < R - 1; k++) { float ph = CELLA(S[k], S[k + 1], S[k - 1], nb[k]); float d = ph - S[k]; \
      if (VAR == 6) {} else if (VAR == 4 && (k & 1)) accb += d * d; else acc += d * d; D[k] = ph; } \
    const float pI = wl * (acc + accb);                                                  \
    if (VAR != 5) __builtin_amdgcn_sched_barrier(0);                                     \
    { _Pragma("unroll") for (int st = 1; st < NW; st *= 2) _Pragma("unroll") for (int q = 0; q + st < NW; q += 2 * st) e8[q] += e8[q + st]; \
      errsum += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, e8[0]), 0)); } \
    hW = (w > 0) ? hWr : S[0]; hE = (w < NW - 1) ? hEr : S[R - 1];                       \
    float p0 = CELLA(S[0], S[1], hW, nb[0]), pl = CELLA(S[R - 1], hE, S[R - 2], nb[R - 1]); \
    float d0 = p0 - S[0], dl = pl - S[R - 1]; D[0] = p0; D[R - 1] = pl;                  \
    float tot = wave_sum63(pI + wl * (d0 * d0) + wl * (dl * dl));                        \
    ex[xb][w][0][lane] = p0; ex[xb][w][1][lane] = pl; if (lane == 63) errp[xb][w] = tot; \
    __syncthreads();                                                                     \
    _Pragma("unroll") for (int q = 0; q < NW; q++) e8[q] = errp[xb][q];                  \
    hWr = ex[xb][wm][1][lane]; hEr = ex[xb][wp][0][lane]; xb ^= 1;                       \
  }
  for (int it = 0; it < nsweep; it += 2) { SWEEPA(A, B) SWEEPA(B, A) }
  float r = errsum;
  for (int k = 0; k < R; k++) r += A[k];
  out[(blockIdx.x * 1024 + tid) % (256 * 512)] = r;
}

// ---- A0 with s_memtime stamps (diagnostic): where does a sweep's time go?  segments: after barrier -> interior cells
// done and LDS reads arrived | -> edge cells, reduction, LDS stores issued and drained | -> through the barrier
__global__ __launch_bounds__(512) void sweepA_stamp(float* out, const float* in, int nsweep, float cx, unsigned long long* seg_out) {
  constexpr int R = 16, NW = 8, VAR = 0;
  __shared__ float ex[2][NW][2][64];
  __shared__ __attribute__((aligned(16))) float errp[2][16];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  float A[R], B[R], nb[R];
  for (int k = 0; k < R; k++) { A[k] = 0; nb[k] = in[((blockIdx.x * 1024 + tid) * 16 + k) % (256 * 512 * 16)]; }
  const float cB = (lane == 0 || lane == 63) ? cx : 0.f, wl = 1.f + (lane == 0) + (lane == 63);
  const int wm = w > 0 ? w - 1 : 0, wp = w < NW - 1 ? w + 1 : NW - 1;
  float hW = 0, hE = 0, hWr = 0, hEr = 0, e8[NW], errsum = 0;
  for (int q = 0; q < NW; q++) e8[q] = 0;
  int xb = 0;
  unsigned long long seg[3] = {0, 0, 0}, tl = __builtin_amdgcn_s_memtime();
#define STAMP(i) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)"); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); seg[i] += t_ - tl; tl = t_; __builtin_amdgcn_sched_barrier(0); }
#define SWEEPS(S, D)                                                                     \
  {                                                                                      \
    float acc = 0;                                                                       \
    _Pragma("unroll") for (int k = 1; k < R - 1; k++) { float ph = CELLA(S[k], S[k + 1], S[k - 1], nb[k]); float d = ph - S[k]; acc += d * d; D[k] = ph; } \
    const float pI = wl * acc;                                                           \
    STAMP(0)                                                                             \
    { _Pragma("unroll") for (int st = 1; st < NW; st *= 2) _Pragma("unroll") for (int q = 0; q + st < NW; q += 2 * st) e8[q] += e8[q + st]; \
      errsum += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, e8[0]), 0)); } \
    hW = (w > 0) ? hWr : S[0]; hE = (w < NW - 1) ? hEr : S[R - 1];                       \
    float p0 = CELLA(S[0], S[1], hW, nb[0]), pl = CELLA(S[R - 1], hE, S[R - 2], nb[R - 1]); \
    float d0 = p0 - S[0], dl = pl - S[R - 1]; D[0] = p0; D[R - 1] = pl;                  \
    float tot = wave_sum63(pI + wl * (d0 * d0) + wl * (dl * dl));                        \
    ex[xb][w][0][lane] = p0; ex[xb][w][1][lane] = pl; if (lane == 63) errp[xb][w] = tot; \
    STAMP(1)                                                                             \
    __syncthreads();                                                                     \
    _Pragma("unroll") for (int q = 0; q < NW; q++) e8[q] = errp[xb][q];                  \
    hWr = ex[xb][wm][1][lane]; hEr = ex[xb][wp][0][lane]; xb ^= 1;                       \
    STAMP(2)                                                                             \
  }
  for (int it = 0; it < nsweep; it += 2) { SWEEPS(A, B) SWEEPS(B, A) }
  float r = errsum;
  for (int k = 0; k < R; k++) r += A[k];
  out[(blockIdx.x * 1024 + tid) % (256 * 512)] = r;
  if (lane == 0 && blockIdx.x == 0) for (int q = 0; q < 3; q++) seg_out[w * 3 + q] = seg[q] / nsweep;
}

// ---- B: lanes along x, two columns per lane, 8 rows per wave -----------------------------------
__global__ __launch_bounds__(512) void sweepB(float* out, const float* in, int nsweep, float cx) {
  constexpr int NR = 8, NW = 8;   // rows per wave; cell (r, c): row w*8 + r, column 2*lane + c
  __shared__ float ex[2][NW][2][2][64];   // [buf][wave][bottom row | top row][column 0/1][lane]
  __shared__ __attribute__((aligned(16))) float errp[2][16];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  float A[NR][2], B[NR][2], nb[NR][2];
  for (int r = 0; r < NR; r++) for (int c = 0; c < 2; c++) { A[r][c] = 0; nb[r][c] = in[(blockIdx.x * 512 + tid) * 16 + r * 2 + c]; }
  // x walls: lane 0 column 0 / lane 63 column 1 mirror themselves (Neumann); y walls: rows 0 / 63 mirror themselves
  const float cW = (lane == 0) ? cx : 0.f, cE = (lane == 63) ? cx : 0.f;
  const float w0 = 1.f + (lane == 0), w1 = 1.f + (lane == 63);
  const int wm = w > 0 ? w - 1 : 0, wp = w < NW - 1 ? w + 1 : NW - 1;
  float hS[2] = {0, 0}, hN[2] = {0, 0}, hSr[2] = {0, 0}, hNr[2] = {0, 0}, e8[8], errsum = 0;
  for (int q = 0; q < 8; q++) e8[q] = 0;
  int xb = 0;
  // cell (r, 0): west = lane-1's column 1 (DPP), east = own column 1; cell (r, 1): west = own column 0, east = lane+1's column 0
#define CELLB0(S, r, s_, n_) (cx * add_shr(((n_) + (s_)) + S[r][1], S[r][1]) + (cW * S[r][0] + nb[r][0]))
#define CELLB1(S, r, s_, n_) (cx * add_shl(((n_) + (s_)) + S[r][0], S[r][0]) + (cE * S[r][1] + nb[r][1]))
#define ROWB(S, D, r, s0, s1, n0, n1, rw)                                                \
  { float p0 = CELLB0(S, r, s0, n0), p1 = CELLB1(S, r, s1, n1);                          \
    float d0 = p0 - S[r][0], d1 = p1 - S[r][1]; acc += (rw) * (w0 * (d0 * d0) + w1 * (d1 * d1)); D[r][0] = p0; D[r][1] = p1; }
#define SWEEPB(S, D)                                                                     \
  {                                                                                      \
    float acc = 0;                                                                       \
    _Pragma("unroll") for (int r = 1; r < NR - 1; r++) ROWB(S, D, r, S[r - 1][0], S[r - 1][1], S[r + 1][0], S[r + 1][1], 1.f) \
    __builtin_amdgcn_sched_barrier(0);                                                   \
    { float s = ((e8[0] + e8[1]) + (e8[2] + e8[3])) + ((e8[4] + e8[5]) + (e8[6] + e8[7])); errsum += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s), 0)); } \
    _Pragma("unroll") for (int c = 0; c < 2; c++) { hS[c] = (w > 0) ? hSr[c] : S[0][c]; hN[c] = (w < NW - 1) ? hNr[c] : S[NR - 1][c]; } \
    const float rwb = (w == 0) ? 2.f : 1.f, rwt = (w == NW - 1) ? 2.f : 1.f;             \
    ROWB(S, D, 0, hS[0], hS[1], S[1][0], S[1][1], rwb)                                   \
    ROWB(S, D, NR - 1, S[NR - 2][0], S[NR - 2][1], hN[0], hN[1], rwt)                    \
    float tot = wave_sum63(acc);                                                         \
    _Pragma("unroll") for (int c = 0; c < 2; c++) { ex[xb][w][0][c][lane] = D[0][c]; ex[xb][w][1][c][lane] = D[NR - 1][c]; } \
    if (lane == 63) errp[xb][w] = tot;                                                   \
    __syncthreads();                                                                     \
    _Pragma("unroll") for (int q = 0; q < 8; q++) e8[q] = errp[xb][q];                   \
    _Pragma("unroll") for (int c = 0; c < 2; c++) { hSr[c] = ex[xb][wm][1][c][lane]; hNr[c] = ex[xb][wp][0][c][lane]; } \
    xb ^= 1;                                                                             \
  }
  for (int it = 0; it < nsweep; it += 2) { SWEEPB(A, B) SWEEPB(B, A) }
  float r = errsum;
  for (int q = 0; q < NR; q++) r += A[q][0] + A[q][1];
  out[blockIdx.x * 512 + tid] = r;
}

// ---- C: mapping A with TWO sweeps per barrier: depth-2 halos, the strip-edge neighbours' columns are computed
// redundantly in the first sub-sweep (R+2 cells), the second sub-sweep needs nothing from other waves (R cells);
// both residuals are reduced and published together.  Same arithmetic per cell and per residual as A0.
__global__ __launch_bounds__(512) void sweepC(float* out, const float* in, int nsweep, float cx) {
  constexpr int R = 16, NW = 8;
  __shared__ float ex[2][NW][4][64];   // columns 0, 1, R-2, R-1 of the newest phi
  __shared__ __attribute__((aligned(16))) float errp[2][2][8];
  __shared__ float nbx[NW][2][64];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  float A[R], B[R], nb[R];
  for (int k = 0; k < R; k++) { A[k] = 0; nb[k] = in[(blockIdx.x * 512 + tid) * R + k]; }
  const float cB = (lane == 0 || lane == 63) ? cx : 0.f, wl = 1.f + (lane == 0) + (lane == 63);
  const int wm = w > 0 ? w - 1 : 0, wp = w < NW - 1 ? w + 1 : NW - 1;
  nbx[w][0][lane] = nb[0]; nbx[w][1][lane] = nb[R - 1];
  __syncthreads();
  const float nbW = nbx[wm][1][lane], nbE = nbx[wp][0][lane];
  float hW1 = 0, hW2 = 0, hE1 = 0, hE2 = 0, e1[8], e2[8], errsum = 0;
  for (int q = 0; q < 8; q++) { e1[q] = 0; e2[q] = 0; }
  int xb = 0;
#define CELLC(c, e, wv, nbk) (cx * add2dpp((e) + (wv), c) + (cB * (c) + (nbk)))
  for (int it = 0; it < nsweep; it += 2) {
    // sub-sweep 1: A -> B on columns -1 .. R; interior first
    float acc1 = 0;
#pragma unroll
    for (int k = 1; k < R - 1; k++) { float ph = CELLC(A[k], A[k + 1], A[k - 1], nb[k]); float d = ph - A[k]; acc1 += d * d; B[k] = ph; }
    __builtin_amdgcn_sched_barrier(0);
    { float s1 = ((e1[0] + e1[1]) + (e1[2] + e1[3])) + ((e1[4] + e1[5]) + (e1[6] + e1[7]));
      float s2 = ((e2[0] + e2[1]) + (e2[2] + e2[3])) + ((e2[4] + e2[5]) + (e2[6] + e2[7]));
      errsum += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s1), 0)) +
                __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s2), 0)); }
    const float a_w1 = (w > 0) ? hW1 : A[0], a_e1 = (w < NW - 1) ? hE1 : A[R - 1];   // phi_{n-1} at columns -1, R
    float p0 = CELLC(A[0], A[1], a_w1, nb[0]), pl = CELLC(A[R - 1], a_e1, A[R - 2], nb[R - 1]);
    float pW = CELLC(hW1, A[0], hW2, nbW), pE = CELLC(hE1, hE2, A[R - 1], nbE);   // neighbours' edge columns, redundantly
    { float d0 = p0 - A[0], dl = pl - A[R - 1]; acc1 += d0 * d0; acc1 += dl * dl; }
    B[0] = p0; B[R - 1] = pl;
    pW = (w > 0) ? pW : p0; pE = (w < NW - 1) ? pE : pl;                           // walls: Neumann ghost = new edge value
    // sub-sweep 2: B -> A on the own columns, nothing needed from other waves
    float acc2 = 0;
#pragma unroll
    for (int k = 0; k < R; k++) {
      const float e = (k < R - 1) ? B[k < R - 1 ? k + 1 : 0] : pE, wv = (k > 0) ? B[k > 0 ? k - 1 : 0] : pW;
      float ph = CELLC(B[k], e, wv, nb[k]); float d = ph - B[k]; acc2 += d * d; A[k] = ph;
    }
    const float t1 = wave_sum63(wl * acc1), t2 = wave_sum63(wl * acc2);
    ex[xb][w][0][lane] = A[0]; ex[xb][w][1][lane] = A[1]; ex[xb][w][2][lane] = A[R - 2]; ex[xb][w][3][lane] = A[R - 1];
    if (lane == 63) { errp[xb][0][w] = t1; errp[xb][1][w] = t2; }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; q++) { e1[q] = errp[xb][0][q]; e2[q] = errp[xb][1][q]; }
    hW1 = ex[xb][wm][3][lane]; hW2 = ex[xb][wm][2][lane]; hE1 = ex[xb][wp][0][lane]; hE2 = ex[xb][wp][1][lane];
    xb ^= 1;
  }
  float r = errsum;
  for (int k = 0; k < R; k++) r += A[k] + B[k];
  out[blockIdx.x * 512 + tid] = r;
}


// ---- D: mapping A re-ordered so that nothing dependent sits in front of the barrier --------------------------
// (1) the two strip-edge cells are computed in the MIDDLE of the sweep (after H interior cells have covered the
//     latency of the halo reads) and their LDS stores are covered by the remaining interior cells;
// (2) LAG = 1: the wave reduction of a sweep's residual partial is done DURING the next sweep, one DPP step after
//     each of its first interior cells, and published before that sweep's barrier: the convergence decision about
//     sweep k is taken early in sweep k+2 (phi rotates through three arrays so that phi_k is still intact then);
//     LAG = 0: reduction at the end of the sweep as in A (two arrays).
#define RED_STEP(r, CTRL) asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r))
__device__ __forceinline__ float red_bcast(float r, int which) {   // row_bcast:15 (rows 1,3) / row_bcast:31 (rows 2,3), behind s_nop 1
  float t = 0;
  if (which == 0) asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(t) : "v"(r));
  else asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(t) : "v"(r));
  return r + t;
}
template <int LAG, int H>
__global__ __launch_bounds__(512) void sweepD(float* out, const float* in, int nsweep, float cx) {
  constexpr int R = 16, NW = 8;
  __shared__ float ex[2][NW][2][64];
  __shared__ __attribute__((aligned(16))) float errp[2][16];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  float A[R], B[R], C[R], nb[R];
  for (int k = 0; k < R; k++) { A[k] = 0; B[k] = 0; C[k] = 0; nb[k] = in[((blockIdx.x * 1024 + tid) * 16 + k) % (256 * 512 * 16)]; }
  const float cB = (lane == 0 || lane == 63) ? cx : 0.f, wl = 1.f + (lane == 0) + (lane == 63);
  const int wm = w > 0 ? w - 1 : 0, wp = w < NW - 1 ? w + 1 : NW - 1;
  float hWr = 0, hEr = 0, e8[NW], errsum = 0, ppart = 0;
  for (int q = 0; q < NW; q++) e8[q] = 0;
  int xb = 0;
#define CELLD(c, e, wv, nbk) (cx * add2dpp((e) + (wv), c) + (cB * (c) + (nbk)))
#define INTERIOR(S, D, k) { float ph = CELLD(S[k], S[k + 1], S[k - 1], nb[k]); float d = ph - S[k]; acc += d * d; D[k] = ph; }
#define SWEEPD(S, D)                                                                     \
  {                                                                                      \
    float acc = 0, red = ppart;                                                          \
    _Pragma("unroll") for (int k = 1; k <= H; k++) {                                     \
      INTERIOR(S, D, k)                                                                  \
      if (LAG == 1) {                                                                    \
        if (k == 1) red += dppf<0x111, 0xf>(red);                                        \
        if (k == 2) red += dppf<0x112, 0xf>(red);                                        \
        if (k == 3) red += dppf<0x114, 0xf>(red);                                        \
        if (k == 4) red += dppf<0x118, 0xf>(red);                                        \
        if (k == 5) red += dppf<0x142, 0xa>(red);                                        \
        if (k == 6) red += dppf<0x143, 0xc>(red);                                        \
      }                                                                                  \
      if (LAG == 2) {   /* the same steps as inline asm behind s_nop 1 */                \
        if (k == 1) RED_STEP(red, "row_shr:1");                                          \
        if (k == 2) RED_STEP(red, "row_shr:2");                                          \
        if (k == 3) RED_STEP(red, "row_shr:4");                                          \
        if (k == 4) RED_STEP(red, "row_shr:8");                                          \
        if (k == 5) red = red_bcast(red, 0);                                             \
        if (k == 6) red = red_bcast(red, 1);                                             \
      }                                                                                  \
    }                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                   \
    const float hW = (w > 0) ? hWr : S[0], hE = (w < NW - 1) ? hEr : S[R - 1];           \
    const float p0 = CELLD(S[0], S[1], hW, nb[0]), pl = CELLD(S[R - 1], hE, S[R - 2], nb[R - 1]); \
    const float d0 = p0 - S[0], dl = pl - S[R - 1]; D[0] = p0; D[R - 1] = pl;            \
    ex[xb][w][0][lane] = p0; ex[xb][w][1][lane] = pl;                                    \
    if (LAG) { if (lane == 63) errp[xb][w] = red; }                                      \
    __builtin_amdgcn_sched_barrier(0);                                                   \
    _Pragma("unroll") for (int k = H + 1; k < R - 1; k++) INTERIOR(S, D, k)              \
    { _Pragma("unroll") for (int st = 1; st < NW; st *= 2) _Pragma("unroll") for (int q = 0; q + st < NW; q += 2 * st) e8[q] += e8[q + st]; \
      errsum += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, e8[0]), 0)); } \
    ppart = wl * acc + wl * (d0 * d0) + wl * (dl * dl);                                  \
    if (!LAG) { const float tot = wave_sum63(ppart); if (lane == 63) errp[xb][w] = tot; } \
    __syncthreads();                                                                     \
    _Pragma("unroll") for (int q = 0; q < NW; q++) e8[q] = errp[xb][q];                  \
    hWr = ex[xb][wm][1][lane]; hEr = ex[xb][wp][0][lane]; xb ^= 1;                       \
  }
  if (LAG) { for (int it = 0; it < nsweep; it += 3) { SWEEPD(A, B) SWEEPD(B, C) SWEEPD(C, A) } }
  else { for (int it = 0; it < nsweep; it += 2) { SWEEPD(A, B) SWEEPD(B, A) } }
  float r = errsum;
  for (int k = 0; k < R; k++) r += A[k] + B[k] + C[k];
  out[(blockIdx.x * 1024 + tid) % (256 * 512)] = r;
}

// ---- E: mapping A with the cell arithmetic in hand-ordered inline-asm blocks of 4 / 3 / 2 cells: the same seven
// instructions per cell as A0 (add, 2 fused DPP adds, ghost fma, cx fmac, difference, residual fmac -- bit-identical
// results), but interleaved across the cells of a block so that no DPP add directly follows its producer: no s_nop,
// no dependent-DPP stall; the two edge cells sit in the middle of the sweep (halo reads covered by the first 7
// interior cells, their stores by the last 7); the exchange buffers are addressed with compile-time parity.
#define DPP_UP " wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define DPP_DN " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
__device__ __forceinline__ void cells4(float& d0, float& d1, float& d2, float& d3, float sm, float s0, float s1, float s2,
                                       float s3, float sp, float n0, float n1, float n2, float n3, float cB, float cx, float& acc) {
  float t0, t1, t2, t3;
  asm volatile(
      "v_add_f32 %[t0], %[s1], %[sm]\n v_add_f32 %[t1], %[s2], %[s0]\n v_add_f32 %[t2], %[s3], %[s1]\n v_add_f32 %[t3], %[sp], %[s2]\n"
      "v_fma_f32 %[d0], %[cB], %[s0], %[n0]\n v_fma_f32 %[d1], %[cB], %[s1], %[n1]\n v_fma_f32 %[d2], %[cB], %[s2], %[n2]\n v_fma_f32 %[d3], %[cB], %[s3], %[n3]\n"
      "v_add_f32_dpp %[t0], %[s0], %[t0]" DPP_UP "v_add_f32_dpp %[t1], %[s1], %[t1]" DPP_UP "v_add_f32_dpp %[t2], %[s2], %[t2]" DPP_UP "v_add_f32_dpp %[t3], %[s3], %[t3]" DPP_UP
      "v_add_f32_dpp %[t0], %[s0], %[t0]" DPP_DN "v_add_f32_dpp %[t1], %[s1], %[t1]" DPP_DN "v_add_f32_dpp %[t2], %[s2], %[t2]" DPP_DN "v_add_f32_dpp %[t3], %[s3], %[t3]" DPP_DN
      "v_fmac_f32 %[d0], %[cx], %[t0]\n v_fmac_f32 %[d1], %[cx], %[t1]\n v_fmac_f32 %[d2], %[cx], %[t2]\n v_fmac_f32 %[d3], %[cx], %[t3]\n"
      "v_sub_f32 %[t0], %[d0], %[s0]\n v_sub_f32 %[t1], %[d1], %[s1]\n v_sub_f32 %[t2], %[d2], %[s2]\n v_sub_f32 %[t3], %[d3], %[s3]\n"
      "v_fmac_f32 %[acc], %[t0], %[t0]\n v_fmac_f32 %[acc], %[t1], %[t1]\n v_fmac_f32 %[acc], %[t2], %[t2]\n v_fmac_f32 %[acc], %[t3], %[t3]\n"
      : [d0] "=&v"(d0), [d1] "=&v"(d1), [d2] "=&v"(d2), [d3] "=&v"(d3), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2),
        [t3] "=&v"(t3), [acc] "+v"(acc)
      : [sm] "v"(sm), [s0] "v"(s0), [s1] "v"(s1), [s2] "v"(s2), [s3] "v"(s3), [sp] "v"(sp), [n0] "v"(n0), [n1] "v"(n1),
        [n2] "v"(n2), [n3] "v"(n3), [cB] "v"(cB), [cx] "s"(cx));
}
__device__ __forceinline__ void cells3(float& d0, float& d1, float& d2, float sm, float s0, float s1, float s2, float sp,
                                       float n0, float n1, float n2, float cB, float cx, float& acc) {
  float t0, t1, t2;
  asm volatile(
      "v_add_f32 %[t0], %[s1], %[sm]\n v_add_f32 %[t1], %[s2], %[s0]\n v_add_f32 %[t2], %[sp], %[s1]\n"
      "v_fma_f32 %[d0], %[cB], %[s0], %[n0]\n v_fma_f32 %[d1], %[cB], %[s1], %[n1]\n v_fma_f32 %[d2], %[cB], %[s2], %[n2]\n"
      "v_add_f32_dpp %[t0], %[s0], %[t0]" DPP_UP "v_add_f32_dpp %[t1], %[s1], %[t1]" DPP_UP "v_add_f32_dpp %[t2], %[s2], %[t2]" DPP_UP
      "v_add_f32_dpp %[t0], %[s0], %[t0]" DPP_DN "v_add_f32_dpp %[t1], %[s1], %[t1]" DPP_DN "v_add_f32_dpp %[t2], %[s2], %[t2]" DPP_DN
      "v_fmac_f32 %[d0], %[cx], %[t0]\n v_fmac_f32 %[d1], %[cx], %[t1]\n v_fmac_f32 %[d2], %[cx], %[t2]\n"
      "v_sub_f32 %[t0], %[d0], %[s0]\n v_sub_f32 %[t1], %[d1], %[s1]\n v_sub_f32 %[t2], %[d2], %[s2]\n"
      "v_fmac_f32 %[acc], %[t0], %[t0]\n v_fmac_f32 %[acc], %[t1], %[t1]\n v_fmac_f32 %[acc], %[t2], %[t2]\n"
      : [d0] "=&v"(d0), [d1] "=&v"(d1), [d2] "=&v"(d2), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [acc] "+v"(acc)
      : [sm] "v"(sm), [s0] "v"(s0), [s1] "v"(s1), [s2] "v"(s2), [sp] "v"(sp), [n0] "v"(n0), [n1] "v"(n1), [n2] "v"(n2),
        [cB] "v"(cB), [cx] "s"(cx));
}
// the two strip-edge cells: (s0; east s1, west hW) and (sl; east hE, west sl1); returns the new values and differences
__device__ __forceinline__ void cells_edge(float& p0, float& pl, float& df0, float& dfl, float s0, float s1, float hW, float n0,
                                           float sl, float hE, float sl1, float nl, float cB, float cx) {
  float t0, t1;
  asm volatile(
      "v_add_f32 %[t0], %[s1], %[hW]\n v_add_f32 %[t1], %[hE], %[sl1]\n"
      "v_fma_f32 %[p0], %[cB], %[s0], %[n0]\n v_fma_f32 %[pl], %[cB], %[sl], %[nl]\n"
      "v_add_f32_dpp %[t0], %[s0], %[t0]" DPP_UP "v_add_f32_dpp %[t1], %[sl], %[t1]" DPP_UP
      "v_add_f32_dpp %[t0], %[s0], %[t0]" DPP_DN "v_add_f32_dpp %[t1], %[sl], %[t1]" DPP_DN
      "v_fmac_f32 %[p0], %[cx], %[t0]\n v_fmac_f32 %[pl], %[cx], %[t1]\n"
      "v_sub_f32 %[df0], %[p0], %[s0]\n v_sub_f32 %[dfl], %[pl], %[sl]\n"
      : [p0] "=&v"(p0), [pl] "=&v"(pl), [df0] "=&v"(df0), [dfl] "=&v"(dfl), [t0] "=&v"(t0), [t1] "=&v"(t1)
      : [s0] "v"(s0), [s1] "v"(s1), [hW] "v"(hW), [n0] "v"(n0), [sl] "v"(sl), [hE] "v"(hE), [sl1] "v"(sl1), [nl] "v"(nl),
        [cB] "v"(cB), [cx] "s"(cx));
}

template <int EDGE_MID>
__global__ __launch_bounds__(512) void sweepE(float* out, const float* in, int nsweep, float cx) {
  constexpr int R = 16, NW = 8;
  __shared__ float ex[2][NW][2][64];
  __shared__ __attribute__((aligned(16))) float errp[2][16];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  float A[R], B[R], nb[R];
  for (int k = 0; k < R; k++) { A[k] = 0; B[k] = 0; nb[k] = in[((blockIdx.x * 1024 + tid) * 16 + k) % (256 * 512 * 16)]; }
  const float cB = (lane == 0 || lane == 63) ? cx : 0.f, wl = 1.f + (lane == 0) + (lane == 63);
  const int wm = w > 0 ? w - 1 : 0, wp = w < NW - 1 ? w + 1 : NW - 1;
  float hWr = 0, hEr = 0, e8[NW], errsum = 0;
  for (int q = 0; q < NW; q++) e8[q] = 0;
#define SWEEPE(S, D, XB)                                                                 \
  {                                                                                      \
    float acc = 0, p0, pl, d0, dl;                                                       \
    cells4(D[1], D[2], D[3], D[4], S[0], S[1], S[2], S[3], S[4], S[5], nb[1], nb[2], nb[3], nb[4], cB, cx, acc); \
    cells3(D[5], D[6], D[7], S[4], S[5], S[6], S[7], S[8], nb[5], nb[6], nb[7], cB, cx, acc);                 \
    if (EDGE_MID) {                                                                      \
      const float hW = (w > 0) ? hWr : S[0], hE = (w < NW - 1) ? hEr : S[R - 1];         \
      cells_edge(p0, pl, d0, dl, S[0], S[1], hW, nb[0], S[R - 1], hE, S[R - 2], nb[R - 1], cB, cx); \
      ex[XB][w][0][lane] = p0; ex[XB][w][1][lane] = pl;                                  \
    }                                                                                    \
    cells4(D[8], D[9], D[10], D[11], S[7], S[8], S[9], S[10], S[11], S[12], nb[8], nb[9], nb[10], nb[11], cB, cx, acc); \
    cells3(D[12], D[13], D[14], S[11], S[12], S[13], S[14], S[15], nb[12], nb[13], nb[14], cB, cx, acc);      \
    { _Pragma("unroll") for (int st = 1; st < NW; st *= 2) _Pragma("unroll") for (int q = 0; q + st < NW; q += 2 * st) e8[q] += e8[q + st]; \
      errsum += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, e8[0]), 0)); } \
    if (!EDGE_MID) {                                                                     \
      const float hW = (w > 0) ? hWr : S[0], hE = (w < NW - 1) ? hEr : S[R - 1];         \
      cells_edge(p0, pl, d0, dl, S[0], S[1], hW, nb[0], S[R - 1], hE, S[R - 2], nb[R - 1], cB, cx); \
      ex[XB][w][0][lane] = p0; ex[XB][w][1][lane] = pl;                                  \
    }                                                                                    \
    D[0] = p0; D[R - 1] = pl;                                                            \
    const float tot = wave_sum63(wl * acc + wl * (d0 * d0) + wl * (dl * dl));            \
    if (lane == 63) errp[XB][w] = tot;                                                   \
    __syncthreads();                                                                     \
    _Pragma("unroll") for (int q = 0; q < NW; q++) e8[q] = errp[XB][q];                  \
    hWr = ex[XB][wm][1][lane]; hEr = ex[XB][wp][0][lane];                                \
  }
  for (int it = 0; it < nsweep; it += 2) { SWEEPE(A, B, 0) SWEEPE(B, A, 1) }
  float r = errsum;
  for (int k = 0; k < R; k++) r += A[k] + B[k];
  out[(blockIdx.x * 1024 + tid) % (256 * 512)] = r;
}

int main() {
  const int nwg = 256, nsweep = 4000;
  float *in, *out;
  hipMalloc(&in, nwg * 512 * 16 * sizeof(float));
  hipMalloc(&out, nwg * 1024 * sizeof(float));
  std::vector<float> h(nwg * 512 * 16);
  for (size_t i = 0; i < h.size(); i++) h[i] = 1e-3f * (float)((i * 2654435761u) % 1000) / 1000.f;
  hipMemcpy(in, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  int clk = 0;
  hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
  const char* names[24] = {"A0 lanes along y, fused DPP pair (kernel)", "A1 compiler-scheduled DPP builtins", "A2 two v_mov_dpp + plain adds",
                          "A3 two independent fused DPP adds", "B  lanes along x, 1 DPP per cell", "C  A0 with two sweeps per barrier (depth-2 halos)",
                          "A0 with 16 waves x 8 columns", "A0 with 4 waves x 32 columns", "A0 with 12 waves x 11 columns (132 columns: x 0.97)",
                          "A0 with 13 waves x 10 columns (130 columns)", "A0 with 10 waves x 13 columns (130 columns)",
                          "A4 two residual accumulators", "A5 no sched_barrier", "A6 no interior residual (floor, wrong)",
                          "D  edges mid-sweep (H=6), reduction at the end", "D  edges mid-sweep (H=6), LAGGED reduction, 3 arrays", "D  edges mid-sweep (H=8), LAGGED reduction", "D  edges mid-sweep (H=4), reduction at the end", "D edges mid-sweep (H=10), LAGGED", "D edges at end (H=14), LAGGED", "E  asm cell blocks, edges mid-sweep", "E  asm cell blocks, edges at the end", "D  edges mid (H=6), LAGGED, reduction steps behind s_nop 1", "D  edges mid (H=8), LAGGED, reduction steps behind s_nop 1"};
  for (int rep = 0; rep < 2; rep++)
    for (int v = 0; v < 24; v++) {
      hipEventRecord(e0);
      if (v == 0) hipLaunchKernelGGL(sweepA<0>, dim3(nwg), dim3(512), 0, 0, out, in, nsweep, 0.25f);
      if (v == 1) hipLaunchKernelGGL(sweepA<1>, dim3(nwg), dim3(512), 0, 0, out, in, nsweep, 0.25f);
      if (v == 2) hipLaunchKernelGGL(sweepA<2>, dim3(nwg), dim3(512), 0, 0, out, in, nsweep, 0.25f);
      if (v == 3) hipLaunchKernelGGL(sweepA<3>, dim3(nwg), dim3(512), 0, 0, out, in, nsweep, 0.25f);
      if (v == 4) hipLaunchKernelGGL(sweepB, dim3(nwg), dim3(512), 0, 0, out, in, nsweep, 0.25f);
      if (v == 5) hipLaunchKernelGGL(sweepC, dim3(nwg), dim3(512), 0, 0, out, in, nsweep, 0.25f);
      if (v == 11) hipLaunchKernelGGL(sweepA<4>, dim3(nwg), dim3(512), 0, 0, out, in, nsweep, 0.25f);
      if (v == 12) hipLaunchKernelGGL(sweepA<5>, dim3(nwg), dim3(512), 0, 0, out, in, nsweep, 0.25f);
      if (v == 13) hipLaunchKernelGGL(sweepA<6>, dim3(nwg), dim3(512), 0, 0, out, in, nsweep, 0.25f);
      if (v == 14) hipLaunchKernelGGL((sweepD<0, 6>), dim3(nwg), dim3(512), 0, 0, out, in, nsweep, 0.25f);
      if (v == 15) hipLaunchKernelGGL((sweepD<1, 6>), dim3(nwg), dim3(512), 0, 0, out, in, 3999, 0.25f);
      if (v == 16) hipLaunchKernelGGL((sweepD<1, 8>), dim3(nwg), dim3(512), 0, 0, out, in, 3999, 0.25f);
      if (v == 17) hipLaunchKernelGGL((sweepD<0, 4>), dim3(nwg), dim3(512), 0, 0, out, in, nsweep, 0.25f);
      if (v == 18) hipLaunchKernelGGL((sweepD<1, 10>), dim3(nwg), dim3(512), 0, 0, out, in, 3999, 0.25f);
      if (v == 19) hipLaunchKernelGGL((sweepD<1, 14>), dim3(nwg), dim3(512), 0, 0, out, in, 3999, 0.25f);
      if (v == 20) hipLaunchKernelGGL(sweepE<1>, dim3(nwg), dim3(512), 0, 0, out, in, nsweep, 0.25f);
      if (v == 21) hipLaunchKernelGGL(sweepE<0>, dim3(nwg), dim3(512), 0, 0, out, in, nsweep, 0.25f);
      if (v == 22) hipLaunchKernelGGL((sweepD<2, 6>), dim3(nwg), dim3(512), 0, 0, out, in, 3999, 0.25f);
      if (v == 23) hipLaunchKernelGGL((sweepD<2, 8>), dim3(nwg), dim3(512), 0, 0, out, in, 3999, 0.25f);
      if (v == 6) hipLaunchKernelGGL((sweepA<0, 8, 16>), dim3(nwg), dim3(1024), 0, 0, out, in, nsweep, 0.25f);
      if (v == 7) hipLaunchKernelGGL((sweepA<0, 32, 4>), dim3(nwg), dim3(256), 0, 0, out, in, nsweep, 0.25f);
      if (v == 8) hipLaunchKernelGGL((sweepA<0, 11, 12>), dim3(nwg), dim3(768), 0, 0, out, in, nsweep, 0.25f);
      if (v == 9) hipLaunchKernelGGL((sweepA<0, 10, 13>), dim3(nwg), dim3(832), 0, 0, out, in, nsweep, 0.25f);
      if (v == 10) hipLaunchKernelGGL((sweepA<0, 13, 10>), dim3(nwg), dim3(640), 0, 0, out, in, nsweep, 0.25f);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("%-45s %.0f ns per sweep = ~%.0f cycles at %.2f GHz (max clock)\n", names[v], ms * 1e6 / nsweep, ms * 1e6 / nsweep * (clk * 1e-6), clk * 1e-6);
    }
  {
    unsigned long long* sg; hipMalloc(&sg, 24 * 8);
    hipLaunchKernelGGL(sweepA_stamp, dim3(nwg), dim3(512), 0, 0, out, in, nsweep, 0.25f, sg);
    unsigned long long h[24]; hipMemcpy(h, sg, sizeof(h), hipMemcpyDeviceToHost);
    for (int w = 0; w < 8; w++) printf("stamped A0, wave %d: cells+reads %llu | edges+reduce+stores %llu | barrier+read issue %llu cycles per sweep\n", w, h[3 * w], h[3 * w + 1], h[3 * w + 2]);
  }
  return 0;
}
