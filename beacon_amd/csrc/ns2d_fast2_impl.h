// ns2d_fast2_impl.h -- register-resident action step for grids with 64 < ny <= 128, two rows per lane: kernel templates.
// Instantiated by ns2d_fast2.hip (100x100: mixing's default grid, rayleigh at L = H = 2) and by ns2d_jit.hip (any other grid).
//
// Same construction as ns2d_fast.hip, with TWO rows per lane: lane l holds rows j = 2l+1 and 2l+2
// of the columns i = w*R+1 .. owned by wave w (R columns; the last wave takes the remainder RL, its
// code is a second instantiation of the same body selected by a wave-uniform branch, so 100 columns
// run as 7 x 13 + 9 on 8 waves = 512 threads = 2 waves per SIMD with 256 VGPRs each).  Consequences:
//   * a cell has ONE cross-lane y-neighbour (row 2l+1 looks down to lane l-1's upper row, row 2l+2
//     looks up to lane l+1's lower row), the other one is the thread's own register: one DPP
//     shift per cell instead of two;
//   * the lanes above the last row pair are kept at phi = 0, which IS mixing's top Dirichlet
//     ghost (mixing.py:450-451); a Neumann top (rayleigh) is a per-lane coefficient of the
//     centre value, as is the Neumann bottom;
//   * u*, v* are stored in place of u, v in LDS after the predictor (u, v are dead until the
//     corrector rewrites them), so the Jacobi loop keeps p, rhs and the phi ping-pong in VGPRs:
//     4 x 2R registers;
//   * transport: ONE wave walks the skewed wavefront "lane l works on column t - l" with both of
//     its rows per step; the west values stay in registers, the south value of the lower row comes
//     from the lane below by DPP (transport_chain2).
// Semantics and citations: ns2d_generic.hip.  Plain launch, one workgroup per replica.
#pragma once
#include <stdlib.h>

#include <type_traits>

#ifndef BCN_PD2F
#define BCN_PD2F 6    // steps per block of the float32 transport wave (two register sets: it runs PD .. 2 PD steps ahead); measured on the
                      // same box, mixing bench workload, cycles per timestep outside the solve / per sweep: 4: 58.6 k / 1 262, 6: 57.2 k / 1 263,
                      // 8: 56.6 k / 1 292 (the out-of-line function's register use changes the caller's allocation in the Jacobi loop)
#endif
#ifndef BCN_PDG2
#define BCN_PDG2 8    // prefetch depth of the transport wave when the fields are in global memory
#endif
#include "bcn_dpp.h"
#include "ns2d.h"
#include "ns2d_device.h"
#include "ns2d_sched.h"

namespace {

using namespace bcn_dpp;

// GF = 1 (float64: three float64 fields exceed LDS): u, v, S live in a per-workgroup global scratch (L2-resident), the
// Poisson rhs in LDS ([cell of the thread][thread]: conflict-free), p and the phi ping-pong in registers.
template <int NX, int NY, int R, int GF = 0>
struct Fast2Geom {
  static_assert(NY <= 128, "two rows per lane");   // odd ny: the last lane's upper row is the ghost row (inactive)
  static constexpr int NW = (NX + R - 1) / R;
  static constexpr int RL = NX - (NW - 1) * R;   // columns of the last wave
  static_assert(NW <= 16 && RL >= 3 && RL <= R, "at most 16 waves, at least 3 columns each");
  static constexpr int NT = NW * 64;
  static constexpr int LH = (NY + 1) / 2;  // active lanes
  static constexpr int SY = NY + 2;
  static constexpr int SX = NX + 2;
  static constexpr int SZ = SX * SY;
  // LDS map (elements): exchange [2][NW][2 sides][2 rows][64] | errp 64 | sact 64 | red 32 | U V S
  static constexpr int EXCH = 2 * NW * 4 * 64;
  static constexpr size_t lds_elems() { return GF ? (size_t)EXCH + 160 + 2 * (size_t)R * NT : (size_t)EXCH + 160 + 3 * (size_t)SZ; }
  static constexpr size_t scratch_elems() { return GF ? 3 * (size_t)SZ + 16 : 0; }   // + 16: the transport wave's sink
};

// Ordered part of the transport step by ONE wave (out of line, see ns2d_fast.hip).  At step t lane l
// works on column i = t - l + 1, rows 2l+1 and 2l+2: S' = A + aW S'(i-1,j) + aS S'(i,j-1) with the
// west values in registers, the south value of the lower row from the lane below (its upper row of
// the previous step) and the explicit part A, u, v prefetched PD steps ahead from LDS.  Lanes outside
// the domain compute on clamped addresses and write to `dummy`.
// GLB: the fields (and `dummy`) are in global memory -- as GLOBAL pointers: as generic pointers of this out-of-line function
// every access was a flat_load / flat_store, which count on both wait counters and cost 64-bit address arithmetic
template <typename real, int NX, int NY, int PD, bool GLB>
__device__ __attribute__((noinline)) void transport_chain2(real* Tl_, const real* Ul_, const real* Vl_, real* dummy_,
                                                           real c0x, real c1x, real c0y, real c1y) {
  typedef typename std::conditional<GLB, __attribute__((address_space(1))) real, real>::type freal;
  freal* const Tl = (freal*)Tl_;
  const freal* const Ul = (const freal*)Ul_;
  const freal* const Vl = (const freal*)Vl_;
  freal* const dummy = (freal*)dummy_;
  constexpr int SY = NY + 2, SZ = (NX + 2) * SY, LH = (NY + 1) / 2;   // PD: steps of prefetch (4 from LDS, more from global memory)
  constexpr bool ODD = (NY & 1) != 0;
  constexpr int NSTEP = NX + LH - 1;
  const int lane = threadIdx.x & 63;
  const bool active = lane < LH;
  const int la = active ? lane : LH - 1;
  const int cb = (1 - la) * SY + 2 * la + 1;     // index of (i, 2l+1) at step t: cb + t*SY
  auto at = [&](int t) { int x = cb + t * SY; x = x < 1 ? 1 : x; return x > SZ - 2 ? SZ - 2 : x; };
  real a0[PD], a1[PD], u0[PD], u1[PD], v0[PD], v1[PD], g[PD];
#pragma unroll
  for (int q = 0; q < PD; q++) {
    const int x = at(q);
    a0[q] = Tl[x]; a1[q] = Tl[x + 1]; g[q] = Tl[x - 1];
    u0[q] = Ul[x]; u1[q] = Ul[x + 1];
    v0[q] = Vl[x]; v1[q] = Vl[x + 1];
  }
  real tp0 = Tl[2 * la + 1], tp1 = Tl[2 * la + 2];   // west ghosts (column 0)
  for (int t0 = 0; t0 < NSTEP; t0 += PD) {
#pragma unroll
    for (int q = 0; q < PD; q++) {
      const int t = t0 + q;
      const real s = from_below(g[q], tp1);          // lane 0: the south ghost T[i][0]
      const real aw0 = c0x + c1x * u0[q], as0 = c0y + c1y * v0[q];
      const real aw1 = c0x + c1x * u1[q], as1 = c0y + c1y * v1[q];
      const real tn0 = a0[q] + aw0 * tp0 + as0 * s;
      const real tn1 = a1[q] + aw1 * tp1 + as1 * tn0;
      const bool ok = active && lane <= t && lane > t - NX;
      tp0 = ok ? tn0 : tp0;
      tp1 = ok ? tn1 : tp1;
      freal* dst = ok ? Tl + (cb + t * SY) : dummy;
      dst[0] = tn0;
      if (ODD) { freal* d1 = (ok && lane < LH - 1) ? dst : dummy; d1[1] = tn1; }   // odd ny: the last lane has no upper row
      else dst[1] = tn1;
      const int x = at(t + PD);
      a0[q] = Tl[x]; a1[q] = Tl[x + 1]; g[q] = Tl[x - 1];
      u0[q] = Ul[x]; u1[q] = Ul[x + 1];
      v0[q] = Vl[x]; v1[q] = Vl[x + 1];
    }
  }
}

// float32 with the fields in LDS (GF == 0): the same recurrence with the instruction count of the lone wave cut (it
// issues one instruction per ~4.6 cycles whatever its kind).  The south ghost row's term of row 1 is folded into A
// beforehand, so that lane 0 needs nothing from below and the south term of the lower row is ONE v_fmac_f32_dpp (a lane
// without a valid source lane keeps its first sum); the two rows of a column are adjacent in LDS (one paired access for
// A, u, v and for the result); blocks of four steps with two register sets, a block first requesting the next block's
// operands.  The control flow stays uniform (EXEC-masked steps made hipcc wait for every LDS operation in flight at each
// step): lanes in front of / behind the domain read a clamped address, keep their west values and write to `dummy`.
typedef float v2f __attribute__((ext_vector_type(2)));
template <int NX, int NY>
__device__ __attribute__((noinline)) void transport_chain2_f32(float* Tl, const float* Ul, const float* Vl, float* dummy,
                                                               float c0x, float c1x, float c0y, float c1y) {
  constexpr int SY = NY + 2, SZ = (NX + 2) * SY, LH = (NY + 1) / 2, PD = BCN_PD2F, NSTEP = NX + LH - 1;
  constexpr bool ODD = (NY & 1) != 0;
  const int lane = threadIdx.x & 63;
  // row 1: A += aS * S[i][0] (LDS accesses of one wave are in program order)
  for (int i = 1 + lane; i <= NX; i += 64) Tl[i * SY + 1] += (c0y + c1y * Vl[i * SY + 1]) * Tl[i * SY];
  if (lane >= LH) return;
  const int cb = (1 - lane) * SY + 2 * lane + 1;     // index of (i, 2l+1) at step t: cb + t*SY
  float tp0 = Tl[2 * lane + 1], tp1 = Tl[2 * lane + 2];   // west ghosts (column 0)
  float aA[2 * PD], uA[2 * PD], vA[2 * PD], aB[2 * PD], uB[2 * PD], vB[2 * PD];
  // MASK 0: every lane inside (steady state); 1: lanes <= t; 2: lanes > t - NX; 3: both tests
#define BCN_LOAD2(RA, RU, RV, T0, MASK)                                                       \
  _Pragma("unroll") for (int q = 0; q < PD; q++) {                                            \
    int x = cb + ((T0) + q) * SY;                                                             \
    if (MASK & 1) x = x < 1 ? 1 : x;                                                          \
    if (MASK & 2) x = x > SZ - 2 ? SZ - 2 : x;                                                \
    RA[2 * q] = Tl[x]; RA[2 * q + 1] = Tl[x + 1];                                             \
    RU[2 * q] = Ul[x]; RU[2 * q + 1] = Ul[x + 1];                                             \
    RV[2 * q] = Vl[x]; RV[2 * q + 1] = Vl[x + 1];                                             \
  }
#define BCN_BLOCK2(RA, RU, RV, NA, NU, NV, T0, MASK, NMASK)                                   \
  {                                                                                           \
    BCN_LOAD2(NA, NU, NV, (T0) + PD, NMASK)                                                   \
    _Pragma("unroll") for (int q = 0; q < PD; q++) {                                          \
      const int t = (T0) + q;                                                                 \
      /* both rows per packed fma: a lone wave issues a v_pk_fma_f32 in the time of a v_fma_f32 */ \
      const v2f ru = {RU[2 * q], RU[2 * q + 1]}, rv = {RV[2 * q], RV[2 * q + 1]}, ra = {RA[2 * q], RA[2 * q + 1]}; \
      const v2f tpp = {tp0, tp1};                                                             \
      const v2f awp = c0x + c1x * ru, asp = c0y + c1y * rv;                                   \
      const v2f tw = ra + awp * tpp;                                                          \
      const float as0 = asp.x, as1 = asp.y;                                                   \
      float t0 = tw.x;                                                                        \
      asm("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf"        \
          : "+v"(t0) : "v"(tp1), "v"(as0));                                                   \
      float t1 = tw.y;                                                                        \
      t1 += as1 * t0;                                                                         \
      float* Tq = Tl + (cb + t * SY);                                                         \
      if (MASK != 0) {                                                                        \
        const bool ok = (!(MASK & 1) || lane <= t) && (!(MASK & 2) || lane > t - NX);         \
        tp0 = ok ? t0 : tp0;                                                                  \
        tp1 = ok ? t1 : tp1;                                                                  \
        Tq = ok ? Tq : dummy;                                                                 \
      } else {                                                                                \
        tp0 = t0; tp1 = t1;                                                                   \
      }                                                                                       \
      if (ODD) { Tq[0] = t0; float* Th = (lane < LH - 1) ? Tq : dummy; Th[1] = t1; }   /* odd ny: the last lane has no upper row */ \
      else { Tq[0] = t0; Tq[1] = t1; }                                                        \
    }                                                                                         \
  }
#define BCN_CHAIN2X(T0, T1, MASK, LASTMASK)                                                   \
  for (int t0 = (T0); t0 < (T1); t0 += 2 * PD) {                                              \
    BCN_BLOCK2(aA, uA, vA, aB, uB, vB, t0, MASK, MASK)                                        \
    if (t0 + 2 * PD < (T1)) BCN_BLOCK2(aB, uB, vB, aA, uA, vA, t0 + PD, MASK, MASK)           \
    else BCN_BLOCK2(aB, uB, vB, aA, uA, vA, t0 + PD, MASK, LASTMASK)                          \
  }
  // lanes 0..LH-1 are all inside the domain for t in [LH-1, NX); phase bounds are multiples of 2 PD.  The last block of
  // a phase prefetches the first block of the next one: with that phase's clamp.
  constexpr int P2 = 2 * PD;
  constexpr int TA2 = ((LH - 1 + P2 - 1) / P2) * P2, TB2 = (NX / P2) * P2, TE2 = ((NSTEP + P2 - 1) / P2) * P2;
  BCN_LOAD2(aA, uA, vA, 0, 1)
  if constexpr (TA2 <= TB2) {
    BCN_CHAIN2X(0, TA2, 1, 0)
    BCN_CHAIN2X(TA2, TB2, 0, 2)
    BCN_CHAIN2X(TB2, TE2, 2, 2)
  } else {
    BCN_CHAIN2X(0, TE2, 3, 3)
  }
#undef BCN_CHAIN2X
#undef BCN_BLOCK2
#undef BCN_LOAD2
}

template <typename real, int NX, int NY, int R, int RW, int KIND, bool EQ, int GF>
__device__ __forceinline__ void fast2_body(const NS2DArgs<real>& A, char* smem, const int w, const int b,
                                           const int it_begin, const int it_end, const bool first_chunk,
                                           const bool last_chunk) {
  using G = Fast2Geom<NX, NY, R, GF>;
  constexpr int NW = G::NW, NT = G::NT, SY = G::SY, SX = G::SX, SZ = G::SZ, LH = G::LH;
  real* exch = reinterpret_cast<real*>(smem);
  real* errp = exch + G::EXCH;   // [2][2][16]: reference norm / unweighted norm partials
  real* sact = errp + 64;        // [64]
  real* red = sact + 64;         // [32]
  real* gscr = GF ? A.fscr + (size_t)blockIdx.x * A.fscr_stride : nullptr;
  real* Ul = GF ? gscr : red + 32;
  real* Vl = Ul + SZ;
  real* Tl = Vl + SZ;
  real* nbl = red + 32;          // GF: Poisson rhs [2 * R][NT]
  real* sink = GF ? gscr + 3 * SZ : red + 16;   // where the transport wave's masked lanes write

  const int tid = threadIdx.x, lane = tid & 63;
  const bool active = lane < LH;
  constexpr bool ODD = (NY & 1) != 0;
  const bool act1 = active && !(ODD && lane == LH - 1);   // the upper row of the pair exists (odd ny: not in the last lane)
  const int la = active ? lane : 0;          // lanes past the top row pair shadow lane 0 (never write)
  int j0 = 2 * la + 1;                       // rows j0 (a = 0) and j0 + 1 (a = 1); GF: laundered once per timestep (below)
  const int i0 = w * R + 1;
  const size_t off = (size_t)b * A.ncell;
  real* __restrict__ gu = A.u + off;
  real* __restrict__ gv = A.v + off;
  real* __restrict__ gp = A.p + off;
  real* __restrict__ gS = A.S + off;
  auto ex = [&](int buf, int wave, int side, int a) -> real* {
    return exch + (((buf * NW + wave) * 2 + side) * 2 + a) * 64;
  };

  // ---- load: HBM [j][i] -> LDS [i][j]; p -> registers --------------------------------------
  for (int c = tid; c < SX * SY; c += NT) {
    const int jj = c / SX, ii = c - jj * SX;
    Ul[ii * SY + jj] = gu[c];
    Vl[ii * SY + jj] = gv[c];
    Tl[ii * SY + jj] = gS[c];
  }
  if (tid < 64) errp[tid] = 0;
  real p[2][RW];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int k = 0; k < RW; k++) p[a][k] = (a == 0 ? active : act1) ? gp[(j0 + a) * SX + i0 + k] : real(0);

  // ---- action conditioning (rayleigh.py:162-171) / wall speeds (mixing.py:212-234) ----------
  real u_t = 0, u_b = 0, v_l = 0, v_r = 0;
  if (KIND == 0 && !first_chunk) {   // later chunks of a scheduled step reuse the conditioned vector
    if (tid < A.n_sgts) sact[tid] = A.a_last[(size_t)b * A.n_sgts + tid];
  } else if (KIND == 0) {
    const int n = A.n_sgts;
    const real* src = A.actions ? A.actions + (size_t)b * n : A.a_last + (size_t)b * n;
    real mean = 0;
    for (int k = 0; k < n; k++) mean += src[k];
    mean /= (real)n;
    real m = 1;
    for (int k = 0; k < n; k++) {
      real t = bcn_abs(src[k] - mean) / A.C;
      m = t > m ? t : m;
    }
    real mine = (tid < n) ? (src[tid] - mean) / m : real(0);
    __syncthreads();
    if (tid < n) {
      sact[tid] = mine;
      A.a_last[(size_t)b * n + tid] = mine;
      if (A.actions_norm) A.actions_norm[(size_t)b * n + tid] = mine;
    }
  } else {
    const int act = (A.iactions && first_chunk) ? A.iactions[b] : A.ia_last[b];
    __syncthreads();
    if (tid == 0 && first_chunk) A.ia_last[b] = act;
    if (act == 0) { u_b = A.u_max; u_t = -A.u_max; }
    if (act == 1) { u_b = -A.u_max; u_t = A.u_max; }
    if (act == 2) { v_r = A.u_max; v_l = -A.u_max; }
    if (act == 3) { v_r = -A.u_max; v_l = A.u_max; }
  }
  __syncthreads();

  const real dt = A.dt, rdx = A.rdx, rdy = A.rdy, rdx2 = A.rdx2, rdy2 = A.rdy2;
  const real cx = A.cx, cy = A.cy;
  // lanes past the top row pair keep phi = 0: zero coefficients and zero rhs
  const real cxl = active ? cx : real(0), cyl = active ? cy : real(0);
  const real cxl1 = act1 ? cx : real(0), cyl1 = act1 ? cy : real(0);   // upper row
  // y-ghost coefficients of the centre value: bottom row (lane 0, a = 0) always Neumann; top row
  // (last active lane, a = 1) Neumann for rayleigh, Dirichlet 0 for mixing
  const bool top0 = ODD && lane == LH - 1 && KIND == 0, top1 = !ODD && lane == LH - 1 && KIND == 0;   // the top row's Neumann ghost
  const real cB0 = ((lane == 0) ? cy : real(0)) + (top0 ? cy : real(0));
  const real cB1 = top1 ? cy : real(0);
  // error weights of this lane's two rows (ghosts copy their interior neighbour)
  const real wl0 = active ? real(1) + (lane == 0 ? 1 : 0) + (top0 ? 1 : 0) : real(0);
  const real wl1 = act1 ? real(1) + (top1 ? 1 : 0) : real(0);
  // weights of the strip's first / last column: a wall column's cells also stand for their ghost copies in the ghost
  // column (additive: the corner ghosts are never set, rayleigh.py:432-446)
  const real cW0 = wl0 + ((active && w == 0) ? real(1) : real(0)), cW1 = wl1 + ((act1 && w == 0) ? real(1) : real(0));
  const real cE0 = wl0 + ((active && w == NW - 1) ? real(1) : real(0)), cE1 = wl1 + ((act1 && w == NW - 1) ? real(1) : real(0));
  const int wm = (w > 0) ? w - 1 : 0, wp = (w < NW - 1) ? w + 1 : NW - 1;

  int status = 0;
  int xb = 0;
#ifdef BCN_STAMP
  unsigned long long seg[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long tl = __builtin_amdgcn_s_memtime();
#define BCN_PH(x) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t__ = __builtin_amdgcn_s_memtime(); seg[x] += t__ - tl; tl = t__; __builtin_amdgcn_sched_barrier(0); }
#else
#define BCN_PH(x)
#endif
  if (!first_chunk) status = A.status[b];   // a replica that overflowed stays stopped (status is never NULL: capi.hip)
  const unsigned long long cyc_u0 = __builtin_amdgcn_s_memtime();
  unsigned long long cyc_j = 0;
  // red[21]: solves whose stop sweep the extrapolating plan could not verify ("late stops"); red[22]: solves repeated
  // under the proven plan (conv_plan 3): see the end of the Jacobi loop.  Written and read by thread 0 only.
  real* const guard = red + 21;
  if (tid == 0) { guard[0] = 0; guard[1] = 0; }
  // slow_k (red[26..30]; red[24..25] are the scheduler's words): the slow-mode landing guard of conv_plan 3, as in ns2d_fast_impl.h --
  // [0] per solve, log2 sqrt(3 |d_1|^2 / tol); [1], [3]: log2 of the two mode cutoffs; [2], [4]: their growth bounds
  real* const slow_k = red + 26;
  if (tid == 0) {
    slow_k[0] = 0;
    slow_k[1] = (real)A.slow_l2lc[0]; slow_k[2] = (real)A.slow_cl[0];
    slow_k[3] = (real)A.slow_l2lc[1]; slow_k[4] = (real)A.slow_cl[1];
  }
  for (int it = it_begin; it < it_end && status == 0; it++) {
    // fields in the global scratch: keep hipcc from hoisting the (64-bit) addresses of a whole timestep out of the loop
    if (GF) asm volatile("" : "+v"(j0));
    // ---- boundary conditions on the LDS fields (rayleigh.py:180-202 / mixing.py:153-171) ------
    for (int jj = 1 + tid; jj <= NY; jj += NT) {
      Ul[1 * SY + jj] = 0;
      Ul[(NX + 1) * SY + jj] = 0;
      if (jj >= 2) {
        Vl[0 * SY + jj] = 2 * v_l - Vl[1 * SY + jj];
        Vl[(NX + 1) * SY + jj] = 2 * v_r - Vl[NX * SY + jj];
      }
      Tl[0 * SY + jj] = Tl[1 * SY + jj];
      Tl[(NX + 1) * SY + jj] = Tl[NX * SY + jj];
    }
    for (int ii = 1 + tid; ii <= NX + 1; ii += NT) {
      const bool wall = (ii == 1) || (ii == NX + 1);
      const real utop = wall ? real(0) : Ul[ii * SY + NY];
      const real ubot = wall ? real(0) : Ul[ii * SY + 1];
      Ul[ii * SY + NY + 1] = 2 * u_t - utop;
      Ul[ii * SY + 0] = 2 * u_b - ubot;
      if (ii <= NX) {
        Vl[ii * SY + NY + 1] = 0;
        Vl[ii * SY + 1] = 0;
        if (KIND == 0) {
          Tl[ii * SY + NY + 1] = 2 * A.Tc - Tl[ii * SY + NY];
          const int k = (ii - 1) / A.nx_sgts;
          if (k < A.n_sgts) Tl[ii * SY + 0] = 2 * (A.Th + sact[k]) - Tl[ii * SY + 1];
        } else {
          Tl[ii * SY + NY + 1] = Tl[ii * SY + NY];
          Tl[ii * SY + 0] = Tl[ii * SY + 1];
        }
      }
    }
    ex(xb, w, 1, 0)[lane] = p[0][RW - 1];
    ex(xb, w, 1, 1)[lane] = p[1][RW - 1];
    __syncthreads();
    if (GF) asm volatile("" : "+v"(j0));
    BCN_PH(0)

    // ---- predictor (rayleigh.py:370-407 / mixing.py:381-416) -> u*, v* in place of u, v --------------------
    if constexpr (GF != 0) {
      // Fields in the global scratch (float64): column by column, west to east, over a sliding window of three columns x
      // four rows (j0-1 .. j0+2: contiguous, one wide load per field and column) -- a fifth of the loads of the row-wise
      // form below, whose 12 loads per cell are what this phase waits for when the fields are not in LDS.  u*, v* of column
      // k-1 are written once column k is computed: by then every reader of the old column k-1 -- columns k-2, k-1, k of
      // this wave, all rows in the same instructions -- has its values in registers.  Only the strip's first and last
      // column are read by OTHER waves: they wait for the barrier.
      const real pWh0 = (w > 0) ? ex(xb, w - 1, 1, 0)[lane] : real(0);
      const real pWh1 = (w > 0) ? ex(xb, w - 1, 1, 1)[lane] : real(0);
      xb ^= 1;
      // ring of columns i0-1 .. (loaded PF columns ahead of their use: the loads are global); column i0-1+q sits in slot q % NC
      constexpr int PF = 3, NC = 3 + PF;
      real uw[NC][4], vw[NC][4];      // [slot][row j0-1, j0, j0+1, j0+2]
      auto load_col = [&](real (&uc)[4], real (&vc)[4], const int i) {
        const int c = i * SY + j0 - 1;
#pragma unroll
        for (int r = 0; r < 4; r++) { uc[r] = Ul[c + r]; vc[r] = Vl[c + r]; }
      };
#pragma unroll
      for (int q = 0; q < 2 + PF; q++)
        if (q <= RW + 1) load_col(uw[q % NC], vw[q % NC], i0 - 1 + q);
      real usF[2], vsF[2], usP[2] = {0, 0}, vsP[2] = {0, 0};   // first column (deferred) / previous column (pending)
#pragma unroll
      for (int k = 0; k < RW; k++) {
        const int i = i0 + k;
        if (k + 2 + PF <= RW + 1) load_col(uw[(k + 2 + PF) % NC], vw[(k + 2 + PF) % NC], i0 - 1 + k + 2 + PF);
        const int sW = k % NC, sC = (k + 1) % NC, sE = (k + 2) % NC;   // slots of columns k-1, k, k+1 (compile-time: k is unrolled)
        real usk[2], vsk[2];
#pragma unroll
        for (int a = 0; a < 2; a++) {
          const int j = j0 + a, r = a + 1;
          const real uc = uw[sC][r], uE_ = uw[sE][r], uW_ = uw[sW][r], uN_ = uw[sC][r + 1], uS_ = uw[sC][r - 1];
          const real vc = vw[sC][r], vE_ = vw[sE][r], vW_ = vw[sW][r], vN_ = vw[sC][r + 1], vS_ = vw[sC][r - 1];
          const real pc = p[a][k];
          const real pW = (k > 0) ? p[a][k > 0 ? k - 1 : 0] : (a == 0 ? pWh0 : pWh1);
          const real pS = (a == 0) ? from_below(p[1][k], p[1][k]) : p[0][k];
          {
            real uE = real(0.5) * (uE_ + uc), uW = real(0.5) * (uc + uW_);
            real uN2 = real(0.5) * (uN_ + uc), uS2 = real(0.5) * (uc + uS_);
            real vN2 = real(0.5) * (vN_ + vw[sW][r + 1]), vS2 = real(0.5) * (vc + vW_);
            real conv = (uE * uE - uW * uW) * rdx + (uN2 * vN2 - uS2 * vS2) * rdy;
            real diff = ((uE_ - 2 * uc + uW_) * rdx2 + (uN_ - 2 * uc + uS_) * rdy2) * A.kmom;
            real pres = (pc - pW) * rdx;
            usk[a] = (i >= 2) ? uc + dt * (diff - conv - pres) : real(0);
          }
          {
            real vE = real(0.5) * (vE_ + vc), vW = real(0.5) * (vc + vW_);
            real uE = real(0.5) * (uE_ + uw[sE][r - 1]), uW = real(0.5) * (uc + uS_);
            real vN2 = real(0.5) * (vN_ + vc), vS2 = real(0.5) * (vc + vS_);
            real conv = (uE * vE - uW * vW) * rdx + (vN2 * vN2 - vS2 * vS2) * rdy;
            real diff = ((vE_ - 2 * vc + vW_) * rdx2 + (vN_ - 2 * vc + vS_) * rdy2) * A.kmom;
            real pres = (pc - pS) * rdy;
            const real buoy = (KIND == 0) ? Tl[i * SY + j] : real(0);
            vsk[a] = (j >= 2) ? vc + dt * (diff - conv - pres + buoy) : real(0);
          }
        }
        // the pending column k-1 (not the strip's first one) goes to memory now: the wave-wide loads of that column have
        // returned (their data went into column k-2 .. k), and the loads in flight are of columns >= k+2.  The fence keeps
        // hipcc from moving a store in front of a load of rows j0 - 1 / j0 + 2 of that column (other lanes' cells: per thread the
        // addresses differ, see the transport's explicit part below)
        asm volatile("" ::: "memory");
        if (k >= 2) {
          if (active) {
            const int c = (i - 1) * SY + j0;
            Ul[c] = usP[0]; Vl[c] = vsP[0];
            if (act1) { Ul[c + 1] = usP[1]; Vl[c + 1] = vsP[1]; }
          }
        }
        if (k == 0) { usF[0] = usk[0]; usF[1] = usk[1]; vsF[0] = vsk[0]; vsF[1] = vsk[1]; }
        usP[0] = usk[0]; usP[1] = usk[1]; vsP[0] = vsk[0]; vsP[1] = vsk[1];
      }
      __syncthreads();   // every read of the old first / last column by the neighbouring strips is done
      if (active) {
        const int cF = i0 * SY + j0, cL = (i0 + RW - 1) * SY + j0;
        Ul[cF] = usF[0]; Vl[cF] = vsF[0];
        Ul[cL] = usP[0]; Vl[cL] = vsP[0];
        if (act1) {
          Ul[cF + 1] = usF[1]; Vl[cF + 1] = vsF[1];
          Ul[cL + 1] = usP[1]; Vl[cL + 1] = vsP[1];
        }
      }
      __syncthreads();
    } else {
    // ---- predictor (rayleigh.py:370-407 / mixing.py:381-416) -> u*, v* (registers, then LDS) ---
    real us[2][RW], vs[2][RW];
    {
      const real pWh0 = (w > 0) ? ex(xb, w - 1, 1, 0)[lane] : real(0);
      const real pWh1 = (w > 0) ? ex(xb, w - 1, 1, 1)[lane] : real(0);
      xb ^= 1;
#pragma unroll
      for (int a = 0; a < 2; a++) {
        __builtin_amdgcn_sched_barrier(0);   // finish one row before loading the next: register pressure
        const int j = j0 + a;
        real ur[RW + 2], uS[RW + 1], uN[RW], vr[RW + 2], vN[RW + 1], vS[RW];
#pragma unroll
        for (int k = 0; k < RW + 2; k++) { ur[k] = Ul[(i0 - 1 + k) * SY + j]; vr[k] = Vl[(i0 - 1 + k) * SY + j]; }
#pragma unroll
        for (int k = 0; k < RW + 1; k++) { uS[k] = Ul[(i0 + k) * SY + j - 1]; vN[k] = Vl[(i0 - 1 + k) * SY + j + 1]; }
#pragma unroll
        for (int k = 0; k < RW; k++) { uN[k] = Ul[(i0 + k) * SY + j + 1]; vS[k] = Vl[(i0 + k) * SY + j - 1]; }
#pragma unroll
        for (int k = 0; k < RW; k++) {
          const int i = i0 + k;
          const real uc = ur[k + 1], uE_ = ur[k + 2], uW_ = ur[k], uN_ = uN[k], uS_ = uS[k];
          const real vc = vr[k + 1], vE_ = vr[k + 2], vW_ = vr[k], vN_ = vN[k + 1], vS_ = vS[k];
          const real pc = p[a][k];
          const real pW = (k > 0) ? p[a][k > 0 ? k - 1 : 0] : (a == 0 ? pWh0 : pWh1);
          // south neighbour of the lower row lives in the lane below (its upper row)
          const real pS = (a == 0) ? from_below(p[1][k], p[1][k]) : p[0][k];
          {
            real uE = real(0.5) * (uE_ + uc), uW = real(0.5) * (uc + uW_);
            real uN2 = real(0.5) * (uN_ + uc), uS2 = real(0.5) * (uc + uS_);
            real vN2 = real(0.5) * (vN_ + vN[k]), vS2 = real(0.5) * (vc + vW_);
            real conv = (uE * uE - uW * uW) * rdx + (uN2 * vN2 - uS2 * vS2) * rdy;
            real diff = ((uE_ - 2 * uc + uW_) * rdx2 + (uN_ - 2 * uc + uS_) * rdy2) * A.kmom;
            real pres = (pc - pW) * rdx;
            us[a][k] = (i >= 2) ? uc + dt * (diff - conv - pres) : real(0);
          }
          {
            real vE = real(0.5) * (vE_ + vc), vW = real(0.5) * (vc + vW_);
            real uE = real(0.5) * (uE_ + uS[k + 1]), uW = real(0.5) * (uc + uS_);
            real vN2 = real(0.5) * (vN_ + vc), vS2 = real(0.5) * (vc + vS_);
            real conv = (uE * vE - uW * vW) * rdx + (vN2 * vN2 - vS2 * vS2) * rdy;
            real diff = ((vE_ - 2 * vc + vW_) * rdx2 + (vN_ - 2 * vc + vS_) * rdy2) * A.kmom;
            real pres = (pc - pS) * rdy;
            const real buoy = (KIND == 0) ? Tl[i * SY + j] : real(0);
            vs[a][k] = (j >= 2) ? vc + dt * (diff - conv - pres + buoy) : real(0);
          }
        }
      }
    }
    __syncthreads();   // every read of the old u, v is done: u*, v* take their place
    if (active) {
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int k = 0; k < RW; k++) {
          if (a == 1 && !act1) continue;
          Ul[(i0 + k) * SY + j0 + a] = us[a][k];
          Vl[(i0 + k) * SY + j0 + a] = vs[a][k];
        }
    }
    __syncthreads();
    }

    // ---- Poisson rhs from u*, v* in LDS (u*[1,.] = u*[nx+1,.] = v*[.,1] = v*[.,ny+1] = 0 are the
    //      wall values the BC pass left there) ------------------------------------------------
    real nb[2][GF ? 1 : RW];   // GF: in LDS (NB below)
#define NB(a, k) (GF ? nbl[((a) * R + (k)) * NT + tid] : nb[a][GF ? 0 : (k)])
#pragma unroll
    for (int a = 0; a < 2; a++) {
      const int j = j0 + a;
#pragma unroll
      for (int k = 0; k < RW; k++) {
        const int c = (i0 + k) * SY + j;
        const real div = (Ul[c + SY] - Ul[c]) * rdx + (Vl[c + 1] - Vl[c]) * rdy;
        NB(a, k) = (a == 0 ? active : act1) ? -A.cb * div : real(0);
      }
    }

    BCN_PH(1)
    const unsigned long long cyc_j0 = __builtin_amdgcn_s_memtime();
    if (A.conv_plan == 3) {
      // |d_1|^2 = |phi_1 - 0|^2 = sum nb^2 (plain, interior): what the slow-mode landing guard scales with (ns2d_fast_impl.h).  One
      // barrier per solve; the partials go to the errp half the first evaluated sweep does not write.
      real a1p = 0;
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int k = 0; k < RW; k++) a1p += NB(a, k) * NB(a, k);
      const real a1w = wave_sum_lane63<real>(a1p);
      if (lane == 63) errp[(xb ^ 1) * 32 + w] = a1w;
      __syncthreads();
      const real a1 = read_lane(row16_sum<real>(errp[(xb ^ 1) * 32 + (lane & 15)]), 15);
      if (tid == 0) slow_k[0] = (real)(0.5f * __log2f(3.f * (float)a1 / (float)A.tol));
    }
    // ---- Jacobi sweeps: one barrier per sweep, phi ping-pong in registers ---------------------
    // The residual is evaluated only on the sweeps that can pass the test (A.conv_plan, see ns2d_fast.hip); a sweep
    // that evaluates it does so behind its own barrier, with the arithmetic the fused form had.
    real phA[2][RW], phB[2][RW];
    real hW0 = 0, hW1 = 0, hE0 = 0, hE1 = 0;
    real hW0r, hW1r, hE0r, hE1r;
    int itp;
    bool finalB;
    // the evaluation plan of this solve: conv_plan 3 is plan 2 whose unverified stops are repeated under plan 1 (below)
    int plan = (A.conv_plan == 3) ? 2 : A.conv_plan;
    for (;;) {   // the solve (once; twice when conv_plan 3 repeats it: u*, v* and the rhs are untouched by the sweeps)
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int k = 0; k < RW; k++) phA[a][k] = 0;
    hW0r = 0; hW1r = 0; hE0r = 0; hE1r = 0;
    itp = 0;
    finalB = false;
    int k_prev = -1;
    float l2u_prev = 0, l2w_prev = 0;
    int skip_left = 0;
    // tolL: what a landing evaluation -- the first one behind skipped sweeps -- must exceed for the skip to be verified: under
    // plan 3 BCN_CONV_GUARD * tol, which proves that no skipped sweep passed (bcn_common.h); under plan 2 tol itself
    // (plan 3: until the first evaluation, then the smaller of that and the slow-mode guard behind the last evaluated sweep)
    real tolL = (A.conv_plan == 3) ? A.tol * real(BCN_CONV_GUARD) : A.tol;
    const float l2tol_u = __log2f((float)A.tol * 1.02f), l2tol_w = __log2f((float)tolL * 1.003f);
    constexpr int JMAX = 256;
    // lower row (a = 0): south = lane below's upper row (DPP), north = own upper row;
    // upper row (a = 1): south = own lower row, north = lane above's lower row (DPP)
#define BCN_CELL(DST, SRC, K, EV, WV)                                                                \
    {                                                                                                \
      const real c0 = SRC[0][K], c1 = SRC[1][K];                                                     \
      real sn0, sn1;   /* south + north of the lower / upper row */                                  \
      add_pair_neighbours(c0, c1, sn0, sn1);                                                         \
      /* mixing: cB1 is 0 at compile time (Dirichlet top) and hipcc cannot fold 0 * x + b itself: 1 237 -> 1 193 cycles per sweep */ \
      const real b1 = (KIND != 0) ? NB(1, K) : cB1 * c1 + NB(1, K);                                  \
      if (EQ) {                                                                                      \
        DST[0][K] = cxl * ((EV##0 + WV##0) + sn0) + (cB0 * c0 + NB(0, K));                           \
        DST[1][K] = cxl1 * ((EV##1 + WV##1) + sn1) + b1;                                             \
      } else {                                                                                       \
        DST[0][K] = cxl * (EV##0 + WV##0) + (cyl * sn0 + (cB0 * c0 + NB(0, K)));                     \
        DST[1][K] = cxl1 * (EV##1 + WV##1) + (cyl1 * sn1 + b1);                                      \
      }                                                                                              \
    }
#define BCN_CELLS(SRC, DST)                                                                          \
      _Pragma("unroll") for (int k = 1; k < RW - 1; k++) {                                            \
        const real e0 = SRC[0][k + 1], e1 = SRC[1][k + 1], w0 = SRC[0][k - 1], w1 = SRC[1][k - 1];  \
        BCN_CELL(DST, SRC, k, e, w)                                                                  \
      }                                                                                              \
      __builtin_amdgcn_sched_barrier(0);   /* halo-dependent part stays behind the interior cells */ \
      hW0 = (w > 0) ? hW0r : SRC[0][0];                                                              \
      hW1 = (w > 0) ? hW1r : SRC[1][0];                                                              \
      hE0 = (w < NW - 1) ? hE0r : SRC[0][RW - 1];                                                     \
      hE1 = (w < NW - 1) ? hE1r : SRC[1][RW - 1];                                                     \
      {                                                                                              \
        const real e0 = SRC[0][1], e1 = SRC[1][1], w0 = hW0, w1 = hW1;                               \
        BCN_CELL(DST, SRC, 0, e, w)                                                                  \
      }                                                                                              \
      {                                                                                              \
        const real e0 = hE0, e1 = hE1, w0 = SRC[0][RW - 2], w1 = SRC[1][RW - 2];                       \
        BCN_CELL(DST, SRC, RW - 1, e, w)                                                              \
      }                                                                                              \
      ex(xb, w, 0, 0)[lane] = DST[0][0];                                                             \
      ex(xb, w, 0, 1)[lane] = DST[1][0];                                                             \
      ex(xb, w, 1, 0)[lane] = DST[0][RW - 1];                                                         \
      ex(xb, w, 1, 1)[lane] = DST[1][RW - 1];
#define BCN_HALO_READS                                                                               \
      hW0r = ex(xb, wm, 1, 0)[lane];                                                                 \
      hW1r = ex(xb, wm, 1, 1)[lane];                                                                 \
      hE0r = ex(xb, wp, 0, 0)[lane];                                                                 \
      hE1r = ex(xb, wp, 0, 1)[lane];                                                                 \
      xb ^= 1;
#define BCN_FAST(SRC, DST) { BCN_CELLS(SRC, DST) __syncthreads(); itp++; BCN_HALO_READS }
#define BCN_CHECK(SRC, DST, DST_IS_B)                                                                \
    {                                                                                                \
      BCN_CELLS(SRC, DST)                                                                            \
      real acc0 = 0, acc1 = 0;                                                                       \
      _Pragma("unroll") for (int k = 1; k < RW - 1; k++) {                                            \
        const real d0 = DST[0][k] - SRC[0][k], d1 = DST[1][k] - SRC[1][k];                           \
        acc0 += d0 * d0; acc1 += d1 * d1;                                                            \
      }                                                                                              \
      const real pI = wl0 * acc0 + wl1 * acc1;                                                       \
      real dW0 = DST[0][0] - SRC[0][0], dW1 = DST[1][0] - SRC[1][0];                                 \
      real dE0 = DST[0][RW - 1] - SRC[0][RW - 1], dE1 = DST[1][RW - 1] - SRC[1][RW - 1];                 \
      dW0 *= dW0; dW1 *= dW1; dE0 *= dE0; dE1 *= dE1;                                                \
      const real part = pI + (cW0 * dW0 + cW1 * dW1) + (cE0 * dE0 + cE1 * dE1);                      \
      const real tot63 = wave_sum_lane63<real>(part);                                                \
      if (lane == 63) errp[xb * 32 + w] = tot63;                                                     \
      if (plan == 1) {   /* unweighted interior norm (lanes past the top row pair hold zeros) */ \
        const real totu63 = wave_sum_lane63<real>((acc0 + acc1) + (dW0 + dW1) + (dE0 + dE1));        \
        if (lane == 63) errp[xb * 32 + 16 + w] = totu63;                                             \
      }                                                                                              \
      __syncthreads();                                                                               \
      itp++;                                                                                         \
      const real eW = errp[xb * 32 + (lane & 15)];                                                   \
      const real eU = (plan == 1) ? errp[xb * 32 + 16 + (lane & 15)] : real(0);               \
      BCN_HALO_READS                                                                                 \
      const real err = read_lane(row16_sum<real>(eW), 15);                                           \
      /* the reference tests the sweep count FIRST (mixing.py:460-463): sweep itmax + 1 overflows even if it passes */ \
      if (itp > A.itmax) { status |= BCN_ST_ITMAX; finalB = DST_IS_B; break; }                       \
      /* a landing (the sweep before this one was not evaluated) that does not clear tolL leaves the skipped sweeps unverified */ \
      const bool unv = plan >= 2 && !A.verify_conv && itp >= 2 && k_prev != itp - 2 && !(err > tolL); \
      if (!(err > A.tol) || unv) {                                                                   \
        if (skip_left > 0) status |= BCN_ST_PLAN;                                                    \
        finalB = DST_IS_B; break;                                                                    \
      }                                                                                              \
      n = 0;                                                                                         \
      if (skip_left > 0) {                                                                           \
        skip_left--;                                                                                 \
      } else if (plan > 0) {                                                                  \
        const float l2w = __log2f((float)err);                                                       \
        float l2u = 0;                                                                               \
        if (plan == 1) l2u = __log2f((float)read_lane(row16_sum<real>(eU), 15));              \
        int j = 0;                                                                                   \
        if (k_prev >= 0) {                                                                           \
          const float rg = 1.f / (float)(itp - 1 - k_prev);                                          \
          if (plan == 1) {                                                                    \
            const float room_u = l2u - l2tol_u, rho_u = (l2u - l2u_prev) * rg;                       \
            if (room_u > 0.f) j = (rho_u < 0.f) ? (int)fminf(room_u / -rho_u, (float)JMAX) : JMAX;   \
          } else {                                                                                   \
            float l2tl = l2tol_w;                                                                    \
            if (A.conv_plan == 3) {   /* the slow-mode guard of the landing this skip ends in; this sweep, itp, is the last evaluated one */ \
              const float e0 = (float)slow_k[0], fi = (float)itp;                                    \
              const float t0 = 1.f + 2.f * exp2f(e0 + fi * (float)slow_k[1]), t1 = 1.f + 2.f * exp2f(e0 + fi * (float)slow_k[3]); \
              const float g = fminf(fminf((float)slow_k[2] * t0 * t0, (float)slow_k[4] * t1 * t1) * 1.001f, (float)BCN_CONV_GUARD); \
              tolL = A.tol * (real)g;                                                                \
              l2tl = __log2f((float)tolL * 1.003f);                                                  \
            }                                                                                        \
            const float room_w = l2w - l2tl, rho_w = (l2w - l2w_prev) * rg;                          \
            int jw = 0;                                                                              \
            if (room_w > 0.f) jw = (rho_w < 0.f) ? (int)fminf(room_w / -rho_w, (float)JMAX) : JMAX;  \
            j = jw - 1 - (jw >> 4) + A.plan_overshoot;                                                                  \
            j = j > 0 ? j : 0;                                                                       \
          }                                                                                          \
        }                                                                                            \
        j = __builtin_amdgcn_readfirstlane(j);                                                       \
        l2u_prev = l2u; l2w_prev = l2w; k_prev = itp - 1;                                            \
        if (A.verify_conv) skip_left = j; else n = j;                                                \
      }                                                                                              \
    }
    for (;;) {
      int n;
      BCN_CHECK(phA, phB, true)
      if (n == 0) {
        BCN_CHECK(phB, phA, false)
        n &= ~1;
      } else {
        BCN_FAST(phB, phA)
        n = (n - 1) & ~1;
      }
      if (n > A.itmax - itp) n = (A.itmax - itp > 0 ? A.itmax - itp : 0) & ~1;   // the overflow test sits in the check sweeps
      for (; n > 0; n -= 2) {
        BCN_FAST(phA, phB)
        BCN_FAST(phB, phA)
      }
    }
#undef BCN_CHECK
#undef BCN_FAST
#undef BCN_HALO_READS
#undef BCN_CELLS
#undef BCN_CELL
    // guard of the extrapolating plan, as in ns2d_fast_impl.h: a passing evaluation that directly follows skipped sweeps
    // is a stop the plan did not foresee ("late stop": counted; conv_plan 3 repeats the solve under the proven plan)
    const bool late = plan >= 2 && itp >= 2 && k_prev != itp - 2 && !(status & BCN_ST_ITMAX);
    if (late && tid == 0) guard[0] += 1;
    if (!(late && A.conv_plan == 3)) break;
    plan = 1;
    if (tid == 0) guard[1] += 1;
    }
#undef NB
    if (finalB) {
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int k = 0; k < RW; k++) phA[a][k] = phB[a][k];
    }
    hW0 = hW0r; hW1 = hW1r;   // west halo of the final phi (unused by wave 0)
    if (A.sweeps && tid == 0) A.sweeps[(size_t)b * A.ndt_act + it] = itp;

    cyc_j += __builtin_amdgcn_s_memtime() - cyc_j0;
    if (GF) asm volatile("" : "+v"(j0));   // (the addresses of the phases below are not kept across the solve)
    BCN_PH(2)
    // ---- p += phi, corrector: u = u* - dt dphi/dx, v = v* - dt dphi/dy (in place in LDS) -------
#pragma unroll
    for (int a = 0; a < 2; a++) {
      const int j = j0 + a;
#pragma unroll
      for (int k = 0; k < RW; k++) {
        const int i = i0 + k;
        const real ph = phA[a][k];
        const real pw = (k > 0) ? phA[a][k > 0 ? k - 1 : 0] : (a == 0 ? hW0 : hW1);
        const real ps = (a == 0) ? from_below(phA[1][k], phA[1][k]) : phA[0][k];
        p[a][k] += ph;
        if (a == 0 ? active : act1) {
          const int c = i * SY + j;
          if (i >= 2) Ul[c] = Ul[c] - dt * (ph - pw) * rdx;
          if (j >= 2) Vl[c] = Vl[c] - dt * (ph - ps) * rdy;
        }
      }
    }
    __syncthreads();

    if (GF) asm volatile("" : "+v"(j0));
    BCN_PH(3)
    // ---- transport: explicit part of every cell, then the ordered part by one wave ------------
    constexpr bool PARX = KIND == 1 && GF == 0 && std::is_same<real, float>::value;
    bool par_done = false;
    if constexpr (GF != 0) {
      // Fields in the global scratch: column by column, west to east.  A cell reads the OLD value of its east and north
      // neighbours, so a column can be written as soon as it is computed: its readers are the column to its west (this lane,
      // earlier in program order) and the row below (this wave, the same load instruction, whose data the store depends
      // on).  Only the strip's first column is read by ANOTHER wave: it waits for the barrier (2 live values instead of
      // 2 * RW float64 pairs).
      real A0[2];
#pragma unroll
      for (int k = 0; k < RW; k++) {
        real Ak[2];
#pragma unroll
        for (int a = 0; a < 2; a++) {
          const int c = (i0 + k) * SY + j0 + a;
          const real uE = Ul[c + SY], uW = Ul[c], vN = Vl[c + 1], vS = Vl[c];
          const real T0 = Tl[c], TE = Tl[c + SY], TN = Tl[c + 1];
          const real expl = A.ksc * ((TE - 2 * T0) * rdx2 + (TN - 2 * T0) * rdy2) -
                            (uE * real(0.5) * (TE + T0) - uW * real(0.5) * T0) * rdx -
                            (vN * real(0.5) * (TN + T0) - vS * real(0.5) * T0) * rdy;
          Ak[a] = T0 + dt * expl;
        }
        // The stores below must stay BEHIND both rows' loads above: the upper row's TN is row j0 + 2 -- the lower row of the
        // lane above, which that lane overwrites with its Ak[0] here.  Per thread the two addresses differ, so nothing but
        // this fence tells hipcc so.  With even ny the two rows leave as ONE 16-byte store that needs Ak[1] and so follows the
        // loads by data dependence; odd ny splits it (the last lane has no upper row) and hipcc then moved the Ak[0] store in
        // front of the upper row's loads: every upper row behind a strip's first column read the NEW value of its north
        // neighbour (50x75 float64, round 5: T wrong by 2e-2 after one timestep).
        asm volatile("" ::: "memory");
        if (k == 0) { A0[0] = Ak[0]; A0[1] = Ak[1]; }
        else if (active) {
          Tl[(i0 + k) * SY + j0] = Ak[0];
          if (act1) Tl[(i0 + k) * SY + j0 + 1] = Ak[1];
        }
      }
      __syncthreads();
      if (active) {
        Tl[i0 * SY + j0] = A0[0];
        if (act1) Tl[i0 * SY + j0 + 1] = A0[1];
      }
    } else {
      real Ac[2][RW];
      real rho_l = 0;   // PARX: max |aW| + |aS| over the thread's cells (the coefficients of the new west / south values, below)
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int k = 0; k < RW; k++) {
          const int c = (i0 + k) * SY + j0 + a;
          const real uE = Ul[c + SY], uW = Ul[c], vN = Vl[c + 1], vS = Vl[c];
          const real T0 = Tl[c], TE = Tl[c + SY], TN = Tl[c + 1];
          const real expl = A.ksc * ((TE - 2 * T0) * rdx2 + (TN - 2 * T0) * rdy2) -
                            (uE * real(0.5) * (TE + T0) - uW * real(0.5) * T0) * rdx -
                            (vN * real(0.5) * (TN + T0) - vS * real(0.5) * T0) * rdy;
          Ac[a][k] = T0 + dt * expl;
          if constexpr (PARX) {
            if (active && (a == 0 || act1))
              rho_l = __builtin_fmaxf((float)rho_l, (float)(bcn_abs(dt * A.ksc * rdx2 + (real(0.5) * dt * rdx) * uW) +
                                                            bcn_abs(dt * A.ksc * rdy2 + (real(0.5) * dt * rdy) * vS)));
          }
        }
      if constexpr (PARX) {
        if (A.transport_iter > 0) {
          const real rw63 = wave_max_lane63<real>(rho_l);
          if (lane == 63) errp[w] = rw63;
        }
      }
      __syncthreads();
      if constexpr (PARX) {
        // The ordered part S' = A + aW S'(i-1,j) + aS S'(i,j-1) (mixing.py:478-497: the in-place sweep reads the NEW west and south
        // values) is a lower-triangular system (I - L) S' = A, and the reference's sweep is its forward substitution: nx + ny/2
        // dependent steps that one wave walks while seven wait (24 k of a timestep's 143 k cycles).  mixing's scalar is passive and
        // barely diffusive (Pe = 1e4): |aW| + |aS| <= 2 dt (1 / (Pe dx^2) + u_max / (2 dx)) = 0.204, so the Neumann series
        // S' = sum_m L^m A converges by a factor rho = max(|aW| + |aS|) per term, and one term is ONE parallel pass of all eight
        // waves over their cells -- two fmas per cell, the strip's last column exchanged as in the Jacobi sweeps.  rho is measured
        // in every timestep; M = the number of terms that leaves rho^(M+1) <= 2^-27 (7e-9 of a scalar in [0, 1]: below float32's
        // rounding of the sweep itself) is 12 at rho = 0.2; where M would exceed A.transport_iter (velocities far above u_max, or
        // the option set to 0) the ordered sweep below runs instead.  float32 only: float64 keeps the reference's order.
        if (A.transport_iter > 0) {
          real aw[2][RW], as[2][RW], X[2][RW];
          const real rho = read_lane(row16_max<real>(errp[lane & 15]), 15);
          // rho^(M+1) <= 2^-27  <=>  M + 1 >= 27 / -log2(rho)
          const float need = (rho > real(0)) ? 27.f / -__log2f((float)rho) : 0.f;
#ifdef BCN_DBG_PASSES   // diagnostic build (scripts/mixstat.py -DBCN_DBG_PASSES=n): a fixed number of passes, to time one (wrong results)
          const int M = BCN_DBG_PASSES + 0 * (int)need;
#else
          const int M = (rho < real(0.9)) ? __builtin_amdgcn_readfirstlane((int)need) : 1 << 20;   // (ceil(need) - 1 <= (int)need)
#endif
          par_done = M <= A.transport_iter;
          if (par_done) {
            // the coefficients, formed here and not in the explicit part (2 x 2 RW more live values through it tripled the spills)
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
              for (int k = 0; k < RW; k++) {
                const int c = (i0 + k) * SY + j0 + a;
                const bool on = active && (a == 0 || act1);
                aw[a][k] = on ? dt * A.ksc * rdx2 + (real(0.5) * dt * rdx) * Ul[c] : real(0);
                as[a][k] = on ? dt * A.ksc * rdy2 + (real(0.5) * dt * rdy) * Vl[c] : real(0);
              }
            // the ghost column i = 0 and the ghost row j = 0 are boundary values, not unknowns: their terms belong to A
            if (w == 0) { Ac[0][0] += aw[0][0] * Tl[0 * SY + j0]; Ac[1][0] += aw[1][0] * Tl[0 * SY + j0 + 1]; }
            if (lane == 0) {
#pragma unroll
              for (int k = 0; k < RW; k++) Ac[0][k] += as[0][k] * Tl[(i0 + k) * SY + 0];
            }
            real (&Ak)[2][RW] = Ac;   // A stays in the registers of the explicit part
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
              for (int k = 0; k < RW; k++) X[a][k] = Ak[a][k];
            // one pass SRC -> DST, every cell from the previous pass's values (two register sets: no value is copied, every fma of a
            // pass is independent of the others -- an in-place sweep in the reference's order converges in as many passes by the
            // same bound and needs no second set, but its 3 x RW dependent instructions ran 2 x longer than these 4 x RW independent ones)
            real Y[2][RW];
            auto pass = [&](const real (&SRC)[2][RW], real (&DST)[2][RW]) {
              ex(xb, w, 1, 0)[lane] = SRC[0][RW - 1];
              ex(xb, w, 1, 1)[lane] = SRC[1][RW - 1];
              __syncthreads();
              const real hw0 = (w > 0) ? ex(xb, w - 1, 1, 0)[lane] : real(0), hw1 = (w > 0) ? ex(xb, w - 1, 1, 1)[lane] : real(0);
              xb ^= 1;
              // the strip's first column -- the only one that needs the halo -- last: the LDS read is in flight behind the others
#pragma unroll
              for (int kk = 1; kk <= RW; kk++) {
                const int k = kk < RW ? kk : 0;
                // the south term of the lower row as ONE v_fmac_f32_dpp: t += aS * (upper row of the lane below); lane 0 has no lane
                // below and keeps t -- its ghost row is folded into A above.  (SRC was written a pass ago: no DPP hazard to wait for.)
                float t = aw[0][k] * (k ? SRC[0][k ? k - 1 : 0] : hw0) + Ak[0][k];
                asm("v_fmac_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(t) : "v"(SRC[1][k]), "v"(as[0][k]));
                DST[0][k] = t;
                DST[1][k] = as[1][k] * SRC[0][k] + (aw[1][k] * (k ? SRC[1][k ? k - 1 : 0] : hw1) + Ak[1][k]);
                if (kk == RW - 1) __builtin_amdgcn_sched_barrier(0);
              }
            };
            int m = 0;
            for (; m + 2 <= M; m += 2) { pass(X, Y); pass(Y, X); }
            if (m < M) {
              pass(X, Y);
#pragma unroll
              for (int a = 0; a < 2; a++)
#pragma unroll
                for (int k = 0; k < RW; k++) X[a][k] = Y[a][k];
            }
            if (active) {
#pragma unroll
              for (int a = 0; a < 2; a++)
#pragma unroll
                for (int k = 0; k < RW; k++) if (a == 0 || act1) Tl[(i0 + k) * SY + j0 + a] = X[a][k];
            }
          }
        }
      }
      if (!par_done && active) {   // the ordered sweep below reads A from T
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
          for (int k = 0; k < RW; k++) if (a == 0 || act1) Tl[(i0 + k) * SY + j0 + a] = Ac[a][k];
      }
    }
    __syncthreads();
    if (GF) asm volatile("" : "+v"(j0));
    BCN_PH(4)
    if (w == 0 && !par_done) {
      if constexpr (std::is_same<real, float>::value && GF == 0)
        transport_chain2_f32<NX, NY>(Tl, Ul, Vl, sink, dt * A.ksc * rdx2, real(0.5) * dt * rdx, dt * A.ksc * rdy2, real(0.5) * dt * rdy);
      else
        transport_chain2<real, NX, NY, (GF ? BCN_PDG2 : 4), (GF != 0)>(Tl, Ul, Vl, sink, dt * A.ksc * rdx2, real(0.5) * dt * rdx,
                                       dt * A.ksc * rdy2, real(0.5) * dt * rdy);
    }
    __syncthreads();
    BCN_PH(5)
  }

  // ---- store: LDS [i][j] -> HBM [j][i]; p and its ghosts -------------------------------------
  for (int c = tid; c < SX * SY; c += NT) {
    const int jj = c / SX, ii = c - jj * SX;
    gu[c] = Ul[ii * SY + jj];
    gv[c] = Vl[ii * SY + jj];
    gS[c] = Tl[ii * SY + jj];
  }
  if (active) {
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int k = 0; k < RW; k++) {
        if (a == 1 && !act1) continue;
        const int i = i0 + k, j = j0 + a, c = j * SX + i;
        const real dp = p[a][k] - gp[c];
        if (i == 1) gp[c - 1] += dp;
        if (i == NX) gp[c + 1] += dp;
        if (j == 1) gp[c - SX] += dp;
        if (j == NY && KIND == 0) gp[c + SX] += dp;
        gp[c] = p[a][k];
      }
  }
  __syncthreads();
  if (last_chunk) {
    ns2d_finish<real, NT>(A, b, gu, gv, gS, status, red);
  } else if (tid == 0) {
    A.status[b] = status;
  }
  if (tid == 0 && A.cyc) {   // this replica's units run one after the other (chunk hand-off): plain read-modify-write
    A.cyc[4 * (size_t)b] += cyc_j;
    A.cyc[4 * (size_t)b + 1] += __builtin_amdgcn_s_memtime() - cyc_u0;
    A.cyc[4 * (size_t)b + 2] += (unsigned long long)guard[0];
    A.cyc[4 * (size_t)b + 3] += (unsigned long long)guard[1];
  }
#ifdef BCN_STAMP   // diagnostic build only: cycles per timestep of each phase over the first obs entries
  __syncthreads();
#ifndef BCN_STAMP_WAVE
#define BCN_STAMP_WAVE 0     // the wave whose phase times are reported
#endif
  if (tid == 64 * BCN_STAMP_WAVE && A.obs_out)
    for (int q = 0; q < 6; q++) A.obs_out[(size_t)b * A.n_obs + q] = (real)seg[q] / (real)(it_end - it_begin);
#endif
}

template <typename real, int NX, int NY, int R, int KIND, bool EQ, int GF>
__device__ __forceinline__ void fast2_unit(const NS2DArgs<real>& A, const int b, const int it_begin, const int it_end,
                                           const bool first_chunk, const bool last_chunk, char* smem) {
  using G = Fast2Geom<NX, NY, R, GF>;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (G::RL != R && w == G::NW - 1)
    fast2_body<real, NX, NY, R, G::RL, KIND, EQ, GF>(A, smem, w, b, it_begin, it_end, first_chunk, last_chunk);
  else
    fast2_body<real, NX, NY, R, R, KIND, EQ, GF>(A, smem, w, b, it_begin, it_end, first_chunk, last_chunk);
}

// plain launch: one workgroup per replica, the whole action step
template <typename real, int NX, int NY, int R, int KIND, bool EQ, int GF>
__global__ __launch_bounds__(((NX + R - 1) / R) * 64) void ns2d_fast2_step(NS2DArgs<real> A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = blockIdx.x;
  if (A.mask && !A.mask[b]) return;
  fast2_unit<real, NX, NY, R, KIND, EQ, GF>(A, b, 0, A.ndt_act, true, true, smem);
}

// ticketed chunk scheduler (ns2d_sched.h): persistent workgroups draw (chunk, replica) units
template <typename real, int NX, int NY, int R, int KIND, bool EQ, int GF>
__global__ __launch_bounds__(((NX + R - 1) / R) * 64) void ns2d_fast2_sched(NS2DArgs<real> A, SchedCtl* ctl, int batch,
                                                                          int nchunk) {
  using G = Fast2Geom<NX, NY, R, GF>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // two words at the end of the `red` scratch row (block_sum uses red[0..NW), the transport sink red[16..17])
  unsigned int* s_words = reinterpret_cast<unsigned int*>(reinterpret_cast<real*>(smem) + G::EXCH + 128 + 24);
  ns2d_sched_loop<real>(A, ctl, batch, nchunk, s_words, [&](int b, int it0, int it1, bool first, bool last) {
    fast2_unit<real, NX, NY, R, KIND, EQ, GF>(A, b, it0, it1, first, last, smem);
  });
}

template <typename real, int NX, int NY, int R, int KIND, bool EQ, int GF>
int launch_fast2_eq(const NS2DArgs<real>& a, int batch, hipStream_t s) {
  using G = Fast2Geom<NX, NY, R, GF>;
  const size_t lds = G::lds_elems() * sizeof(real);
  if (GF && (!a.fscr || a.fscr_stride < G::scratch_elems())) { bcn_set_error("fast2 path: field scratch missing"); return BCN_ERR_UNSUPPORTED; }
  NS2DArgs<real> c = a;
  if (!c.sweeps) c.sweeps = c.sweeps_int;
  const SchedParams sp = ns2d_sched_params(a);
  const int q = sp.q_set ? sp.q : 20;   // 100x100: 20 timesteps per chunk measured best (37.7 vs 38.2 ms at 10)
  if (sp.mode == 2 && batch > sp.grid && a.ndt_act >= 2 * q && a.sched_ctl) {
    auto ks = ns2d_fast2_sched<real, NX, NY, R, KIND, EQ, GF>;
    static unsigned long long set2 = 0;
    if (ns2d_first_on_device(set2)) BCN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ks), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int nchunk = a.ndt_act / q;   // uniform chunks (the long-chunks-first order of ns2d_fast_impl.h measured slower here)
    c.sched_nbig = 0;
    c.sched_q = q;
    BCN_HIP(hipMemsetAsync(a.sched_ctl, 0, a.sched_bytes, s));
    hipLaunchKernelGGL(ks, dim3(sp.grid), dim3(G::NT), lds, s, c, static_cast<SchedCtl*>(a.sched_ctl), batch, nchunk);
    BCN_HIP(hipGetLastError());
    if (a.host) a.host->launched = "ns2d_fast2_sched";
    return BCN_OK;
  }
  if (a.sched_ctl) BCN_HIP(hipMemsetAsync(a.sched_ctl, 0, a.sched_bytes, s));   // cycle counters
  auto k = ns2d_fast2_step<real, NX, NY, R, KIND, EQ, GF>;
  static unsigned long long set = 0;
  if (ns2d_first_on_device(set)) BCN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k, dim3(batch), dim3(G::NT), lds, s, c);
  BCN_HIP(hipGetLastError());
  if (a.host) a.host->launched = "ns2d_fast2_step";
  return BCN_OK;
}

template <typename real, int NX, int NY, int R, int KIND, int GF = 0>
int launch_fast2(const NS2DArgs<real>& a, int batch, hipStream_t s) {
  // dx == dy (every reference configuration): one multiply per cell instead of two
  if (a.cx == a.cy) return launch_fast2_eq<real, NX, NY, R, KIND, true, GF>(a, batch, s);
  return launch_fast2_eq<real, NX, NY, R, KIND, false, GF>(a, batch, s);
}

}  // namespace
