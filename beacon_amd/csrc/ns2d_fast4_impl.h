// ns2d_fast4_impl.h -- rayleigh / mixing action step for TALL grids (128 < ny <= 256): Jacobi sweeps in registers with
// RPL = 2..4 rows per lane, the other phases from HBM/L2 with lanes along x, the ordered transport as a register walk
// along anti-diagonals fed from LDS.
//
// Why not "everything in registers" like ns2d_fast_impl.h / ns2d_fast2_impl.h: 100x200 cells x 6 live fields x 4 B is
// 480 KB, the whole vector register file of a CU (512 KB).  What does fit is the Poisson solve -- phi, phi' and the rhs
// (3 x 80 KB at 100x200 float32) -- and that is where the sweeps are (rayleigh.py:419-454 / mixing.py:428-463).  So:
//
//  * u, v, p, S, us, vs stay in HBM/L2 ([ny+2][nx+2], x fastest): boundary conditions, predictor, p += phi, corrector
//    and the explicit part of the transport run over them with lanes along x (coalesced), as in ns2d_generic.hip.
//  * Poisson: wave w owns the column strip [w R + 1, w R + R], lane l the rows [l RPL + 1, (l+1) RPL], RPL = ceil(ny / 64)
//    (rows past ny in the last lane hold zeros): the vertical neighbours of a cell are in the lane's own registers except
//    across lanes (one DPP move up, one down per COLUMN), the horizontal ones in the wave's own registers except at the
//    strip edges, which go through LDS once per sweep (RPL contiguous values per lane).  A sweep copies nothing: the R
//    columns live in R + 1 register slots and shift by one slot per sweep, east -> west and back (f4_sweep); the barrier of
//    the edge exchange sits in mid-sweep, so that neither an LDS latency nor the barrier's skew is exposed between two
//    sweeps.  The residual of the reference's stop test (sum over the whole array incl. ghosts, rayleigh.py:448-449) is
//    evaluated where the plan says the test can pass (conv_plan 0: every sweep, as the reference; 1 proven, 2 extrapolated,
//    3 extrapolated and guarded -- as in ns2d_fast2_impl.h); an evaluated sweep ends with the barrier of its workgroup
//    reduction.  Up to 16 waves of <= 128 VGPRs: four per SIMD (8 fatter waves measured slower).
//  * the fields are addressed through ONE buffer descriptor (the host lays them out as [6][B][ncell]) with an SGPR offset
//    per field and one 32-bit offset per lane (F4Field): 64-bit addresses per lane and field did not fit the 128 registers.
//  * the reference's IN-PLACE transport sweep (rayleigh.py:468-487): S' = A + aW S'(i-1,j) + aS S'(i,j-1).  A, aW, aS are
//    computed by all waves into LDS (as many rows at a time as fit: two blocks at 100x200 float32); then ONE wave walks
//    the anti-diagonals d = i + j with lanes along x: the west value is the neighbouring lane's previous result (one
//    DPP move), the south value the lane's own previous result -- two dependent FMAs per diagonal, no barrier and no
//    memory round trip inside the chain; the coefficients of the next diagonal are fetched from LDS (odd pitch: no bank
//    conflicts along a diagonal) while the current one is computed.
//
// More replicas than CUs: the kernel body is a unit over a range of timesteps (the state is in HBM between timesteps anyway)
// and runs under the ticket scheduler of ns2d_sched.h.  DESIGN.md 4.2c has the measurements and what bounds each phase.
// Same argument block, state layout, status / sweep-count outputs and episode bookkeeping as the other kernels.
#pragma once
#include <stdlib.h>

#include <type_traits>

#include "bcn_dpp.h"
#include "ns2d.h"
#include "ns2d_device.h"
#include "ns2d_sched.h"

namespace {

using namespace bcn_dpp;

template <int NX, int NY, int R, int RPL>
struct Fast4Geom {
  static constexpr int SX = NX + 2;
  static constexpr int NCELL = SX * (NY + 2);
  static constexpr int NW = (NX + R - 1) / R;       // waves = column strips
  static constexpr int RL = NX - (NW - 1) * R;      // columns of the last strip (1..R)
  static constexpr int NT = NW * BCN_WAVE;
  static constexpr int NL = (NY + RPL - 1) / RPL;   // lanes that hold rows (the last one RT + 1 <= RPL of them)
  static constexpr int RT = (NY - 1) % RPL;         // the top row is row RT of lane NL - 1
  static constexpr int P = SX | 1;                  // LDS pitch of a natural-layout array (odd)
  static constexpr int CPL = (NX + BCN_WAVE - 1) / BCN_WAVE;  // columns per lane of the transport walk
  static constexpr int HROWS = BCN_WAVE * RPL;      // one edge column in the exchange buffer (lane-major)
  static constexpr int FIXED = 2 * 32 + 64 + 16;    // reduction scratch [2][2][16] + conditioned actions + the scheduler's two words
  static constexpr int HAL = 2 * NW * 2 * HROWS;    // [parity][wave][west|east][HROWS]
  static constexpr int WARR = P * (NY + 2);
  static constexpr int LDS_ELEMS_MAX(int esz) { return 160 * 1024 / esz; }
  // transport: rows per block and number of blocks
  static constexpr int br_cap(int esz) { return (LDS_ELEMS_MAX(esz) - FIXED - 2) / (3 * P); }
  static constexpr int nblk(int esz) { return (NY + br_cap(esz) - 1) / br_cap(esz); }
  static constexpr int br(int esz) { return (NY + nblk(esz) - 1) / nblk(esz); }
  static constexpr int lds_elems(int esz) {
    const int jac = FIXED + HAL + WARR, tr = FIXED + 3 * P * br(esz) + 2;
    return jac > tr ? jac : tr;
  }
  static_assert(NL <= BCN_WAVE, "ny <= 64 rows per lane");
  static_assert(NW >= 2 && NW <= 16 && RL >= 1, "2..16 column strips");
};

// Cells (i, j), j_lo <= j <= j_hi, with lanes along x and the rows dealt round-robin to the waves, U rows (U * CPL cells) of
// a lane at a time: ld(j, i) reads what a cell needs (clamped indices: always in range), st(values, j, i, ok) computes and
// writes.  All loads of the U * CPL cells are issued before the first store: the fields of 256 replicas do not fit L2, a
// load costs more than a microsecond, and one cell at a time (the loop of ns2d_generic.hip) pays that once per cell.
template <int NX, int NW, int U, class LD, class ST>
__device__ __forceinline__ void f4_cells(int w, int tx, int j_lo, int j_hi, LD&& ld, ST&& st) {
  constexpr int CPL = (NX + BCN_WAVE - 1) / BCN_WAVE;
  // the lane's indices are recomputed in every phase (a few VALU instructions): left to itself hipcc hoists them out of
  // the timestep loop, keeps them live across the Poisson solve -- where every register is taken -- and reloads them from
  // scratch at each use
  asm volatile("" : "+v"(tx));
  for (int jb = j_lo + w; jb <= j_hi; jb += NW * U) {
    decltype(ld(0, 0)) vals[U][CPL];
#pragma unroll
    for (int uu = 0; uu < U; uu++)
#pragma unroll
      for (int a = 0; a < CPL; a++) {
        const int j = jb + uu * NW, i = 1 + tx + BCN_WAVE * a;
        vals[uu][a] = ld(j <= j_hi ? j : j_hi, i <= NX ? i : NX);
      }
#pragma unroll
    for (int uu = 0; uu < U; uu++)
#pragma unroll
      for (int a = 0; a < CPL; a++) {
        const int j = jb + uu * NW, i = 1 + tx + BCN_WAVE * a;
        const bool ok = (j <= j_hi) && (i <= NX);
        st(vals[uu][a], j <= j_hi ? j : j_hi, i <= NX ? i : NX, ok);
      }
  }
}

// A replica's field in HBM behind ONE buffer descriptor for its six fields (four SGPRs: base = the replica's u, which
// the host lays out as [6][B][ncell], so field f starts f * B * ncell elements further on: `so`, an SGPR byte offset) and
// ONE 32-bit byte offset per lane: `buffer_load_dword v, v_off, s[rsrc], s_so offen offset:imm`.  As plain `real*` a lane
// keeps a 64-bit address per field; the 128-register budget spilled them, and every access then reloaded its address from
// scratch behind an s_waitcnt vmcnt(0), i.e. the loads of a batch went out one by one (round 4).
template <typename real> struct F4Field {
  __amdgpu_buffer_rsrc_t rs;
  int so;
  struct Ref {
    const F4Field& f;
    int i;
    __device__ __forceinline__ operator real() const { return f.ld(i); }
    __device__ __forceinline__ void operator=(real x) const { f.st(i, x); }
    __device__ __forceinline__ void operator=(const Ref& o) const { f.st(i, real(o)); }
  };
  __device__ __forceinline__ real ld(int i) const {
    if constexpr (sizeof(real) == 4) return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, i * 4, so, 0));
    else return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, i * 8, so, 0));
  }
  __device__ __forceinline__ void st(int i, real x) const {
    typedef unsigned int u2 __attribute__((ext_vector_type(2)));
    if constexpr (sizeof(real) == 4) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, x), rs, i * 4, so, 0);
    else __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, x), rs, i * 8, so, 0);
  }
  __device__ __forceinline__ Ref operator[](int i) const { return Ref{*this, i}; }
};
// the layout the descriptor relies on, checked on the host before a launch: the five other fields behind u, everything a
// replica touches within 2 GB of its u (the byte offsets are 32-bit)
template <typename real> inline bool f4_fields_ok(const NS2DArgs<real>& a, size_t ncell) {
  const real* f[6] = {a.u, a.v, a.p, a.S, a.us, a.vs};
  for (int k = 1; k < 6; k++)
    if (f[k] < a.u || ((size_t)(f[k] - a.u) + ncell) * sizeof(real) >= 0x7fffffffull) return false;
  return true;
}
template <typename real> struct F4Pred { real u[6], v[6], p[3], s; };
template <typename real> struct F4Rhs { real u0, u1, v0, v1; };
template <typename real> struct F4Corr { real p, us, vs, gx, gy; };
template <typename real> struct F4Tr { real uW, uE, vS, vN, Tc, TE, TN, TW; };

// One Jacobi sweep of a wave's column strip in registers, WITHOUT a register copy: the columns shift by one slot per sweep.
// An in-place sweep keeps a column's new values in temporaries until its east neighbour has used the old ones and moves
// them home afterwards (70 v_mov of 240 instructions at 7 x 4 cells per lane).  Instead, with R + 1 slots for R columns:
//   DIRB = false (state X -> Y): east -> west, column k from slots k - 1, k, k + 1 into slot k + 1 -- the old column k + 1
//                                there is dead by then (both its neighbours and the column itself are done);
//   DIRB = true  (state Y -> X): west -> east, column k from slots k, k + 1, k + 2 into slot k -- which holds the old
//                                column k - 1, dead for the same reason.
// Same arithmetic per cell as ever (rayleigh.py:428-444), in another order of the columns only.
//
// Edge exchange of a sweep without residual: ONE barrier, in the MIDDLE of the sweep.  The first column a sweep computes is an edge column: it goes
// to the neighbour strip's halo slot at once (the "early" product), the last column when the sweep ends (the "late" one).
// The next sweep runs the other way round, so it needs FIRST the halo its neighbour produced early and LAST the one
// produced late: the early halo was written before the previous sweep's barrier (readable with no wait at all), the late
// one before the neighbour reaches this sweep's barrier -- it is read right behind that barrier and used half a sweep
// later.  No LDS latency and no barrier skew is exposed between two sweeps.  Halo slots: hal[parity][wave][west|east]; a
// sweep reads parity `par` and writes `par ^ 1`, and parity == state (0: X), so every slot has one writer phase and one
// reader phase per double sweep, separated by the barriers.
//   first_rd / last_rd: the halo needed by the first / last column of this direction; first_wr / last_wr: where the first /
//   last column goes.  RC: columns of this strip (R, or RL in the last wave).  RC == 1 (last wave only, whose east "halo"
//   is its own slot): the barrier sits where the one foreign dependence needs it.
// EV: accumulates the squared increments (acc per row, accW / accE of the strip's first / last column) for the stop test.
template <typename real, int R, int RPL, int RC, int KIND, bool EQ, int RT, bool DIRB, bool EV>
__device__ __forceinline__ void f4_sweep(real (&Pv)[R + 1][RPL], const real (&Bv)[R][RPL], const real (&cxr)[RPL],
                                         const real (&cyr)[RPL], const real tmask, const real* first_rd, const real* last_rd,
                                         real* first_wr, real* last_wr, real (&acc)[RPL], real& accW, real& accE) {
  static_assert(RC >= 1 && RC <= R, "columns of the strip");
  // position (in processing order) in front of which the barrier sits; -1: none -- an evaluated sweep ends with a barrier
  // of its own (the workgroup reduction), and the caller puts one between a run of plain sweeps and an evaluated one, so
  // that an evaluated sweep finds both halos complete when it starts
  constexpr int PB = EV ? -1 : (RC >= 2) ? (RC / 2 > 1 ? RC / 2 : 1) : (DIRB ? 1 : 0);
  real hf[RPL], hl[RPL];
#define BCN_F4_SYNC() __syncthreads()
  constexpr bool XCH = true;
  if (PB == 0) BCN_F4_SYNC();
#pragma unroll
  for (int r = 0; r < RPL; r++) hf[r] = XCH ? first_rd[r] : real(0);
  if (PB <= 0 || PB >= RC) {
#pragma unroll
    for (int r = 0; r < RPL; r++) hl[r] = XCH ? last_rd[r] : real(0);
  }
  // A column's cells STAGE BY STAGE (e + w of all of them, then n + s, the sum, the fma) with scheduling barriers in
  // between: left to itself hipcc emits a cell as add, add, add, fma on the registers just written, and the dependent
  // instructions cost more issue time than the four waves of a SIMD hide (plain sweep of 7 x 4 cells: 2 196 cycles cell by
  // cell, 1 916 a column at a time; scripts/sweep_cost.py).  With four rows per lane, that is: with three the two orders
  // measure the same, with two a stage is four columns (200x100: 63.9 ms; 66.4 a column at a time or unstaged), and with
  // one row per lane the scheduler keeps its freedom (300x50: 38.7 ms; 40.5 / 39.2 / 39.6 staged by 1 / 2 / 4 columns).
  // All reads of a group precede its writes.
#ifndef BCN_F4_GROUP   // columns per stage
#define BCN_F4_GROUP(RPL) ((RPL) == 2 ? 4 : 1)
#endif
#ifndef BCN_F4_STAGE_ROWS
#define BCN_F4_STAGE_ROWS 2
#endif
  constexpr bool STAGED = RPL >= BCN_F4_STAGE_ROWS;
  constexpr int G = (STAGED && RC >= 2 * BCN_F4_GROUP(RPL)) ? BCN_F4_GROUP(RPL) : 1;
  constexpr int PBG = (PB > 0 && PB < RC) ? ((PB + G - 1) / G) * G : PB;   // the barrier, moved to the next group boundary
  static_assert(!(PB > 0 && PB < RC) || (PBG >= 1 && PBG <= ((RC - 1) / G) * G), "barrier between the first and the last column");
#pragma unroll
  for (int pos0 = 0; pos0 < RC; pos0 += G) {
    if (pos0 == PBG && PBG > 0 && PBG < RC) {
      BCN_F4_SYNC();
#pragma unroll
      for (int r = 0; r < RPL; r++) hl[r] = XCH ? last_rd[r] : real(0);
    }
    real cv[G][RPL], hs[G][RPL], vs_[G][RPL], out[G][RPL];
    // stage 1: e + w, and the lane-crossing neighbours of each column
    real sdn[G], nup[G];
#pragma unroll
    for (int g = 0; g < G; g++) {
      const int pos = pos0 + g;
      if (pos < RC) {
        const int k = DIRB ? pos : RC - 1 - pos;
#pragma unroll
        for (int r = 0; r < RPL; r++) {
          cv[g][r] = DIRB ? Pv[k + 1][r] : Pv[k][r];
          real wv, ev;
          if (k == 0) wv = DIRB ? hf[r] : hl[r];
          else wv = DIRB ? Pv[k][r] : Pv[k > 0 ? k - 1 : 0][r];
          if (k == RC - 1) ev = DIRB ? hl[r] : hf[r];
          else ev = DIRB ? Pv[k + 2 <= R ? k + 2 : R][r] : Pv[k + 1][r];
          hs[g][r] = ev + wv;
        }
        sdn[g] = from_below<real>(cv[g][0], cv[g][RPL - 1]);
        nup[g] = dpp<0x130, 0xf, 0xf, true>(real(0), cv[g][0]);
      }
    }
    if constexpr (STAGED) __builtin_amdgcn_sched_barrier(0);
    // stage 2: n + s
#pragma unroll
    for (int g = 0; g < G; g++)
      if (pos0 + g < RC) {
#pragma unroll
        for (int r = 0; r < RPL; r++) {
          const real s = (r == 0) ? sdn[g] : cv[g][r > 0 ? r - 1 : 0];
          real n = (r == RPL - 1) ? nup[g] : cv[g][r + 1 < RPL ? r + 1 : r];
          if (KIND == 0 && r == RT) n = tmask * cv[g][r] + n;
          vs_[g][r] = n + s;
        }
      }
    if constexpr (STAGED) __builtin_amdgcn_sched_barrier(0);
    // stage 3: (e + w) + (n + s), or the inner fma where dx != dy
#pragma unroll
    for (int g = 0; g < G; g++)
      if (pos0 + g < RC) {
        const int k = DIRB ? pos0 + g : RC - 1 - (pos0 + g);
#pragma unroll
        for (int r = 0; r < RPL; r++) {
          if constexpr (EQ) hs[g][r] = hs[g][r] + vs_[g][r];
          else hs[g][r] = cxr[r] * hs[g][r] + Bv[k][r];
        }
      }
    if constexpr (STAGED) __builtin_amdgcn_sched_barrier(0);
    // stage 4: the new values
#pragma unroll
    for (int g = 0; g < G; g++)
      if (pos0 + g < RC) {
        const int k = DIRB ? pos0 + g : RC - 1 - (pos0 + g);
#pragma unroll
        for (int r = 0; r < RPL; r++) {
          if constexpr (EQ) out[g][r] = cxr[r] * hs[g][r] + Bv[k][r];
          else out[g][r] = cyr[r] * vs_[g][r] + hs[g][r];
        }
      }
    if constexpr (STAGED) __builtin_amdgcn_sched_barrier(0);
    if (EV) {
#pragma unroll
      for (int g = 0; g < G; g++)
        if (pos0 + g < RC) {
          const int k = DIRB ? pos0 + g : RC - 1 - (pos0 + g);
#pragma unroll
          for (int r = 0; r < RPL; r++) {
            const real d = out[g][r] - cv[g][r];
            acc[r] += d * d;
            if (k == 0) accW += d * d;
            if (k == RC - 1) accE += d * d;
          }
        }
    }
#pragma unroll
    for (int g = 0; g < G; g++)
      if (pos0 + g < RC) {
        const int pos = pos0 + g, k = DIRB ? pos : RC - 1 - pos;
#pragma unroll
        for (int r = 0; r < RPL; r++) {
          if (DIRB) Pv[k][r] = out[g][r]; else Pv[k + 1][r] = out[g][r];
          if (XCH && pos == 0) first_wr[r] = out[g][r];
          if (XCH && pos == RC - 1) last_wr[r] = out[g][r];
        }
      }
  }
  if (PB >= RC) BCN_F4_SYNC();
#undef BCN_F4_SYNC
}

// timesteps [it_begin, it_end) of replica b: the whole action step (plain launch) or one chunk of it (ticket scheduler,
// ns2d_sched.h: the replica's state lives in HBM between timesteps anyway, so a chunk needs no load / store of its own)
template <typename real, int NX, int NY, int R, int RPL, int KIND, bool EQ>
__device__ __forceinline__ void fast4_unit(const NS2DArgs<real>& A, const int b, const int it_begin, const int it_end,
                                           const bool first_chunk, const bool last_chunk, char* smem) {
  using G = Fast4Geom<NX, NY, R, RPL>;
  constexpr int NW = G::NW, NT = G::NT, SX = G::SX, P = G::P, RL = G::RL, NL = G::NL, RT = G::RT, CPL = G::CPL, HROWS = G::HROWS;
  constexpr int BR = G::br(sizeof(real)), NBLK = G::nblk(sizeof(real));
  static_assert(R >= 2, "f4_sweep: a strip other than the last one has at least two columns");
#ifndef BCN_F4_U
#define BCN_F4_U 2
#endif
#ifndef BCN_F4_STATIC_DIR   // shape of the Jacobi loop (see there)
#define BCN_F4_STATIC_DIR(real) (sizeof(real) == 8)
#endif
  constexpr int U = (CPL >= BCN_F4_U) ? 1 : BCN_F4_U / CPL;   // rows of a lane in flight in the HBM/L2 phases
#ifndef BCN_F4_UP
#define BCN_F4_UP 2
#endif
  constexpr int UP = BCN_F4_UP * U;   // ... in the predictor and the corrector (measured: twice as many pay there, not in the transport)
  const int tid = threadIdx.x;
  const int tx = tid & (BCN_WAVE - 1), w = tid >> 6;
  const size_t off = (size_t)b * G::NCELL;

  real* red = reinterpret_cast<real*>(smem);  // [2][2][16]
  // (the transport walk addresses LDS by integer offsets from 0)
  if ((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem != 0u) __builtin_trap();
  real* sact = red + 2 * 32;                  // [64] conditioned actions ([128..129]: the scheduler's words)
  real* hal = red + G::FIXED;                     // Poisson: edge-column exchange
  real* W = hal + G::HAL;                     // Poisson: -rhs in, phi out (natural layout, pitch P)
  real* TX = red + G::FIXED;                     // transport (overlays the two above): A -> S', aW, aS of one row block
  struct alignas(2 * sizeof(real)) F4YZ { real y, z; };   // aW, aS of a cell side by side: one LDS access in the walk
  F4YZ* TYZ = reinterpret_cast<F4YZ*>(TX + ((BR * P + 1) & ~1));

  // (the descriptor must be provably wave-uniform, or hipcc wraps every access in a waterfall loop: b comes out of LDS in
  //  the ticket scheduler)
  const int bu = __builtin_amdgcn_readfirstlane(b);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(A.u + (size_t)bu * G::NCELL, 0, 0x7fffffff, 0x00020000);
  const F4Field<real> u{rs, 0}, v{rs, (int)((A.v - A.u) * (long)sizeof(real))}, p{rs, (int)((A.p - A.u) * (long)sizeof(real))},
      S{rs, (int)((A.S - A.u) * (long)sizeof(real))}, us{rs, (int)((A.us - A.u) * (long)sizeof(real))},
      vs{rs, (int)((A.vs - A.u) * (long)sizeof(real))};

  const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
  unsigned long long t_jac = 0, n_eval = 0, n_late = 0, n_redo = 0;
#ifdef BCN_F4_STAMP   // experiments: shader cycles per phase; replica b reports phase b % 8 in its third counter
  unsigned long long t_ph = 0, t_last = t_begin;
#define BCN_F4_PH(k) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if ((b & 7) == (k)) t_ph += t_ - t_last; t_last = t_; }
#else
#define BCN_F4_PH(k)
#endif

  if (tid < 2 * 32) red[tid] = 0;
  // ---- action conditioning (rayleigh.py:162-171) / wall speeds (mixing.py:212-234) ----
  real u_t = 0, u_b = 0, v_l = 0, v_r = 0;
  if (KIND == 0 && !first_chunk) {   // later chunks of a scheduled step reuse the conditioned vector
    if (tid < A.n_sgts) sact[tid] = A.a_last[(size_t)b * A.n_sgts + tid];
  } else if constexpr (KIND == 0) {
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
    __syncthreads();  // all reads of a_last done before it is rewritten
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

  // ---- Poisson mapping: wave = column strip, lane = RPL rows ----
  const bool lastw = (w == NW - 1);
  bool rowok[RPL];                 // rows past ny (the last lane's, where RPL does not divide ny) hold phi = 0 and stay there
  real cxr[RPL], cyr[RPL];
#pragma unroll
  for (int r = 0; r < RPL; r++) {
    rowok[r] = tx * RPL + r < NY;
    cxr[r] = rowok[r] ? A.cx : real(0);
    cyr[r] = rowok[r] ? A.cy : real(0);
  }
  const int i0 = 1 + w * R, j0l = 1 + tx * RPL;
  const real tmask = (KIND == 0 && tx == NL - 1) ? real(1) : real(0);   // Neumann top wall (rayleigh); mixing: phi = 0 above
  const real bmask = (tx == 0) ? real(1) : real(0);
  // exchange slots: west edge -> east halo of the strip to the left (own west halo at the wall: Neumann), and vice versa
  real* const hw_rd0 = hal + (w * 2 + 0) * HROWS + tx * RPL;
  real* const he_rd0 = hal + (w * 2 + 1) * HROWS + tx * RPL;
  real* const hw_wr0 = hal + ((w == 0 ? 0 : (w - 1) * 2 + 1)) * HROWS + tx * RPL;
  real* const he_wr0 = hal + ((lastw ? w * 2 + 1 : (w + 1) * 2 + 0)) * HROWS + tx * RPL;
  constexpr int HPAR = NW * 2 * HROWS;

  int status = first_chunk ? 0 : A.status[b];   // a replica that overflowed stays stopped (status is never NULL: capi.hip)
  // sweeps of the previous timestep's solve (wave-uniform; 0: unknown): where a solve under an extrapolating plan opens
  // (spec_start, as in ns2d_fast_impl.h)
  int prev_itp = (it_begin > 0 && A.sweeps) ? __builtin_amdgcn_readfirstlane(A.sweeps[(size_t)b * A.ndt_act + it_begin - 1]) : 0;
  int prev2_itp = (it_begin > 1 && A.sweeps) ? __builtin_amdgcn_readfirstlane(A.sweeps[(size_t)b * A.ndt_act + it_begin - 2]) : 0;
  for (int it = it_begin; it < it_end && status == 0; it++) {
    // ---- boundary conditions (rayleigh.py:180-202 / mixing.py:153-171) ----
    BCN_F4_PH(7)
    for (int j = 1 + tid; j <= NY; j += NT) {
      u[j * SX + 1] = 0;
      u[j * SX + NX + 1] = 0;
      if (j >= 2) {
        v[j * SX + 0] = 2 * v_l - v[j * SX + 1];
        v[j * SX + NX + 1] = 2 * v_r - v[j * SX + NX];
      }
      S[j * SX + 0] = S[j * SX + 1];
      S[j * SX + NX + 1] = S[j * SX + NX];
    }
    for (int i = 1 + tid; i <= NX + 1; i += NT) {
      // u[1,.] and u[nx+1,.] are zeroed by the loop above in the same phase: use the value they will have
      const bool wall = (i == 1) || (i == NX + 1);
      real utop = wall ? real(0) : u[NY * SX + i];
      real ubot = wall ? real(0) : u[1 * SX + i];
      u[(NY + 1) * SX + i] = 2 * u_t - utop;
      u[0 * SX + i] = 2 * u_b - ubot;
      if (i <= NX) {
        v[(NY + 1) * SX + i] = 0;
        v[1 * SX + i] = 0;
        if constexpr (KIND == 0) {
          S[(NY + 1) * SX + i] = 2 * A.Tc - S[NY * SX + i];
          int k = (i - 1) / A.nx_sgts;
          if (k < A.n_sgts) S[0 * SX + i] = 2 * (A.Th + sact[k]) - S[1 * SX + i];
        } else {
          S[(NY + 1) * SX + i] = S[NY * SX + i];
          S[0 * SX + i] = S[1 * SX + i];
        }
      }
    }
    __syncthreads();
    BCN_F4_PH(0)

    auto pred_load = [&](int j, int i) {
      const int c = j * SX + i;
      F4Pred<real> q;
      q.u[0] = u[c]; q.u[1] = u[c + 1]; q.u[2] = u[c - 1]; q.u[3] = u[c + SX]; q.u[4] = u[c - SX]; q.u[5] = u[c + 1 - SX];
      q.v[0] = v[c]; q.v[1] = v[c + 1]; q.v[2] = v[c - 1]; q.v[3] = v[c + SX]; q.v[4] = v[c - SX]; q.v[5] = v[c + SX - 1];
      q.p[0] = p[c]; q.p[1] = p[c - 1]; q.p[2] = p[c - SX];
      q.s = (KIND == 0) ? S[c] : real(0);
      return q;
    };
    // u* (cells with i >= 2) and v* (j >= 2) of one cell; 0 where the predictor computes nothing (walls, outside the grid)
    auto pred_cell = [&](const F4Pred<real>& q, int j, int i, bool ok, real& us_o, real& vs_o) {
      const real uc = q.u[0], uE_ = q.u[1], uW_ = q.u[2], uN_ = q.u[3], uS_ = q.u[4];
      const real vc = q.v[0], vE_ = q.v[1], vW_ = q.v[2], vN_ = q.v[3], vS_ = q.v[4];
      const real pc = q.p[0];
      us_o = 0; vs_o = 0;
      if (ok && i >= 2) {
        real uE = real(0.5) * (uE_ + uc), uW = real(0.5) * (uc + uW_);
        real uN = real(0.5) * (uN_ + uc), uS = real(0.5) * (uc + uS_);
        real vN = real(0.5) * (vN_ + q.v[5]), vS = real(0.5) * (vc + vW_);
        real conv = (uE * uE - uW * uW) * A.rdx + (uN * vN - uS * vS) * A.rdy;
        real diff = ((uE_ - 2 * uc + uW_) * A.rdx2 + (uN_ - 2 * uc + uS_) * A.rdy2) * A.kmom;
        real pres = (pc - q.p[1]) * A.rdx;
        us_o = uc + A.dt * (diff - conv - pres);
      }
      if (ok && j >= 2) {
        real vE = real(0.5) * (vE_ + vc), vW = real(0.5) * (vc + vW_);
        real uE = real(0.5) * (uE_ + q.u[5]), uW = real(0.5) * (uc + uS_);
        real vN = real(0.5) * (vN_ + vc), vS = real(0.5) * (vc + vS_);
        real conv = (uE * vE - uW * vW) * A.rdx + (vN * vN - vS * vS) * A.rdy;
        real diff = ((vE_ - 2 * vc + vW_) * A.rdx2 + (vN_ - 2 * vc + vS_) * A.rdy2) * A.kmom;
        real pres = (pc - q.p[2]) * A.rdy;
        vs_o = vc + A.dt * (diff - conv - pres + q.s);
      }
    };
    if constexpr (CPL <= 2) {
      // ---- predictor (rayleigh.py:370-407 / mixing.py:381-416) AND Poisson rhs (rayleigh.py:424-426) in one pass ----
      // The rhs of a cell needs u* of its east neighbour and v* of its north neighbour.  With lanes along x the east value is
      // the next lane's (one DPP move; the last lane's is the first lane's second column) -- and with the rows dealt to the
      // waves in CONTIGUOUS chunks instead of round-robin, the north value is what the wave computes in its next iteration:
      // the rhs of row j - 1 follows the predictor of row j, from registers.  A chunk's last row needs v* of the next
      // chunk's first row: every wave leaves that row of v* in LDS (the exchange buffer is idle here) and the chunks' last
      // rows get their rhs behind a barrier.  u*, v* still go to HBM for the corrector;
      // what goes away is the pass that read them back (2 of the 17 passes over the fields) and its load latency.
      // u*(1, j) = u*(nx + 1, j) = 0 and v*(i, 1) = v*(i, ny + 1) = 0 (walls: the predictor computes nothing there and
      // the arrays hold the zeros they were created with -- what the two-pass form reads).
      constexpr int RB = (NY + NW - 1) / NW;
      static_assert(NW * P <= G::HAL, "one row of v* per wave in the exchange buffer");
      const int ja = 1 + w * RB;
      const int jz = (ja + RB - 1 < NY) ? ja + RB - 1 : NY;      // (ja > NY: no rows)
      real usP[CPL], vsP[CPL];   // u*, v* of the last row done
#pragma unroll
      for (int a = 0; a < CPL; a++) { usP[a] = 0; vsP[a] = 0; }
      // two rows per iteration (the loads of both -- four cells per lane at 100 columns -- go out before the arithmetic);
      // branch-free like f4_cells: a row past the chunk is loaded from a clamped index and computed for nothing
      for (int j = ja; j <= jz; j += 2) {
        F4Pred<real> q[2][CPL];
        real usC[2][CPL], vsC[2][CPL];
        const bool two = j + 1 <= jz;
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
          for (int a = 0; a < CPL; a++) {
            const int jr = j + r, i = 1 + tx + BCN_WAVE * a;
            q[r][a] = pred_load(jr <= jz ? jr : jz, i <= NX ? i : NX);
          }
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
          for (int a = 0; a < CPL; a++) {
            const int jr = j + r, i = 1 + tx + BCN_WAVE * a;
            const int jc = jr <= jz ? jr : jz;
            const bool ok = (i <= NX) && (jr <= jz);
            pred_cell(q[r][a], jc, i <= NX ? i : NX, ok, usC[r][a], vsC[r][a]);
            const int c = jc * SX + (i <= NX ? i : NX);
            if (ok && i >= 2) us[c] = usC[r][a];      // u*, v* to HBM for the corrector
            if (ok && jr >= 2) vs[c] = vsC[r][a];
            if (r == 0 && j == ja && i <= NX) hal[w * P + i] = vsC[0][a];   // the chunk's first row of v*, for the wave below
          }
        // rhs of rows j - 1 (u*, v* carried) and j
#pragma unroll
        for (int r = 0; r < 2; r++) {
          const int jr = j - 1 + r;                                   // the row whose rhs is complete now
          const bool have = (r == 0) ? (j > ja) : two;
#pragma unroll
          for (int a = 0; a < CPL; a++) {
            const int i = 1 + tx + BCN_WAVE * a;
            const real u0 = (r == 0) ? usP[a] : usC[0][a], v0 = (r == 0) ? vsP[a] : vsC[0][a];
            const real v1 = (r == 0) ? vsC[0][a] : vsC[1][a];
            const real un = (r == 0) ? usP[a + 1 < CPL ? a + 1 : a] : usC[0][a + 1 < CPL ? a + 1 : a];
            real usE = dpp<0x130, 0xf, 0xf, true>(real(0), u0);                        // the next lane's column
            if (a + 1 < CPL) usE = (tx == BCN_WAVE - 1) ? read_lane(un, 0) : usE;     // (lane 63: the first lane's next column)
            if (have && i <= NX) W[jr * P + i] = -(A.cb * ((usE - u0) * A.rdx + (v1 - v0) * A.rdy));
          }
        }
#pragma unroll
        for (int a = 0; a < CPL; a++) { usP[a] = two ? usC[1][a] : usC[0][a]; vsP[a] = two ? vsC[1][a] : vsC[0][a]; }
      }
      __syncthreads();
      // the chunk's last row: v* of the row above it is the next wave's first row (the top wall's 0 for the last chunk)
      if (ja <= NY) {
#pragma unroll
        for (int a = 0; a < CPL; a++) {
          const int i = 1 + tx + BCN_WAVE * a;
          const real v1 = (jz < NY && i <= NX) ? hal[(w + 1) * P + (i <= NX ? i : NX)] : real(0);
          const real un = usP[a + 1 < CPL ? a + 1 : a];
          real usE = dpp<0x130, 0xf, 0xf, true>(real(0), usP[a]);
          if (a + 1 < CPL) usE = (tx == BCN_WAVE - 1) ? read_lane(un, 0) : usE;
          if (i <= NX) W[jz * P + i] = -(A.cb * ((usE - usP[a]) * A.rdx + (v1 - vsP[a]) * A.rdy));
        }
      }
      __syncthreads();
      BCN_F4_PH(1)
    } else {
    // ---- predictor (rayleigh.py:370-407 / mixing.py:381-416) ----
    f4_cells<NX, NW, UP>(w, tx, 1, NY, pred_load,
      [&](const F4Pred<real>& q, int j, int i, bool ok) {
        const int c = j * SX + i;
        real us_o, vs_o;
        pred_cell(q, j, i, ok, us_o, vs_o);
        if (ok && i >= 2) us[c] = us_o;
        if (ok && j >= 2) vs[c] = vs_o;
      });
    __syncthreads();
    BCN_F4_PH(1)

    // ---- Poisson rhs (rayleigh.py:424-426), negated, into LDS with lanes along x ----
    f4_cells<NX, NW, 2 * U>(w, tx, 1, NY,
      [&](int j, int i) {
        const int c = j * SX + i;
        return F4Rhs<real>{us[c], us[c + 1], vs[c], vs[c + SX]};
      },
      [&](const F4Rhs<real>& q, int j, int i, bool ok) {
        if (ok) W[j * P + i] = -(A.cb * ((q.u1 - q.u0) * A.rdx + (q.v1 - q.v0) * A.rdy));
      });
    __syncthreads();
    }

    // ---- Jacobi sweeps in registers (rayleigh.py:419-454 / mixing.py:428-463) ----
    const unsigned long long tj0 = __builtin_amdgcn_s_memtime();
    real Pv[R + 1][RPL], Bv[R][RPL];   // R columns in R + 1 slots: f4_sweep shifts them by one per sweep
#pragma unroll
    for (int k = 0; k < R; k++)
#pragma unroll
      for (int r = 0; r < RPL; r++) {
        const bool ok = rowok[r] && (k < RL || !lastw);
        Bv[k][r] = ok ? W[(j0l + r) * P + i0 + k] : real(0);
      }
    int itp = 0, par = 0;
    real err = 0, eU = 0;
    // The sweeps (f4_sweep above; state == parity of the halo buffers: 0 = X).  Evaluated: with the residual of the reference's stop test
    // (weighted: ghosts copy their interior neighbour, one more count per Neumann side) and the plain sum of squares the
    // evaluation plan works with, reduced over the workgroup behind a barrier of its own at the end of the sweep.
    // (a macro around the calls: the wave-uniform branch on `lastw` picks the instantiation, so that no register array is
    //  ever indexed by a run-time value)
#define BCN_F4_CALL(RC_, DIRB_, EV_)                                                                                               \
      f4_sweep<real, R, RPL, RC_, KIND, EQ, RT, DIRB_, EV_>(Pv, Bv, cxr, cyr, tmask,                                               \
          (DIRB_ ? hw_rd0 + HPAR : he_rd0), (DIRB_ ? he_rd0 + HPAR : hw_rd0),                                                      \
          (DIRB_ ? hw_wr0 : he_wr0 + HPAR), (DIRB_ ? he_wr0 : hw_wr0 + HPAR), acc, accW, accE);
    // an evaluated sweep of direction DIRB_ (state X -> Y: false)
#define BCN_F4_EVAL(DIRB_) {                                                                                                       \
      real acc[RPL], accW = 0, accE = 0;                                                                                           \
      _Pragma("unroll")                                                                                                            \
      for (int r = 0; r < RPL; r++) acc[r] = 0;                                                                                    \
      if (RL < R && lastw) BCN_F4_CALL(RL, DIRB_, true) else BCN_F4_CALL(R, DIRB_, true)                                           \
      constexpr int np = DIRB_ ? 0 : 1;                                                                                            \
      real plain = 0;                                                                                                              \
      _Pragma("unroll")                                                                                                            \
      for (int r = 0; r < RPL; r++) plain += acc[r];                                                                               \
      real loc = plain + bmask * acc[0];                                                                                           \
      if constexpr (KIND == 0) loc += tmask * acc[RT];                                                                             \
      if (w == 0) loc += accW;                                                                                                     \
      if (lastw) loc += accE;                                                                                                      \
      loc = wave_sum_lane63<real>(loc);                                                                                            \
      plain = wave_sum_lane63<real>(plain);                                                                                        \
      if (tx == BCN_WAVE - 1) { red[np * 32 + w] = loc; red[np * 32 + 16 + w] = plain; }                                           \
      __syncthreads();                                                                                                             \
      par = np;                                                                                                                    \
      itp++;                                                                                                                       \
      const real eW_ = red[np * 32 + (tx & 15)], eU_ = red[np * 32 + 16 + (tx & 15)];                                              \
      err = read_lane(row16_sum<real>(eW_), 15);                                                                                   \
      eU = read_lane(row16_sum<real>(eU_), 15);                                                                                    \
    }
    // sweeps without residual, two at a time from state DIRB_ (false: X) while n >= 2: straight-line code, no merge of the
    // two directions' register assignments inside the loop; and a single one
#define BCN_F4_PAIRS(DIRB_) {                                                                                                      \
      real acc[RPL], accW = 0, accE = 0;   /* (unused without residual) */                                                         \
      if (RL < R && lastw) {                                                                                                       \
        for (; n >= 2; n -= 2) { BCN_F4_CALL(RL, DIRB_, false) BCN_F4_CALL(RL, !DIRB_, false) itp += 2; }                          \
      } else {                                                                                                                     \
        for (; n >= 2; n -= 2) { BCN_F4_CALL(R, DIRB_, false) BCN_F4_CALL(R, !DIRB_, false) itp += 2; }                            \
      }                                                                                                                            \
    }
#define BCN_F4_PLAIN1(DIRB_) {                                                                                                     \
      real acc[RPL], accW = 0, accE = 0;                                                                                           \
      if (RL < R && lastw) BCN_F4_CALL(RL, DIRB_, false) else BCN_F4_CALL(R, DIRB_, false)                                         \
      par = DIRB_ ? 0 : 1;                                                                                                         \
      itp++;                                                                                                                       \
    }
    // Which sweeps evaluate the residual (conv_plan 0: all, as the reference) -- the plans of ns2d_fast2_impl.h:
    //  1 (proven): the increments obey d(k+1) = J d(k) with J symmetric, so log |d(k)|^2 (plain sum of squares) is convex in
    //    k and never increases: the slope between two evaluations bounds every later slope, and the weighted sum of the stop
    //    test is >= the plain one.  While that bound stays above 1.02 tol (the margin covers the rounding of the sums) no
    //    sweep can pass the test: those sweeps run without residual, reduction and test.
    //  2: the same extrapolation on the weighted sum itself (not proven convex; it is when the residual hovers within the
    //    weights' 10 % of tol for a hundred sweeps, where plan 1 cannot skip anything), one sweep and 1/16 short of the estimate.
    //  3 (float32 default): plan 2 with PROVEN landings: the first evaluation behind skipped sweeps must find the residual above
    //    BCN_CONV_GUARD * tol -- then none of them passed (bcn_common.h) -- and the plan aims there; a landing below it is
    //    counted ("late stop") and the solve -- it starts from phi = 0 and its rhs is still in registers -- repeated under plan 1.
    int plan = A.conv_plan;
    for (;;) {
#pragma unroll
      for (int k = 0; k < R; k++)
#pragma unroll
        for (int r = 0; r < RPL; r++) Pv[k][r] = 0;
      for (int k = tid; k < G::HAL; k += NT) hal[k] = 0;   // both parities of the exchange buffer start from phi = 0
      __syncthreads();
      itp = 0; par = 0;
      // tolL: what a landing evaluation -- the first one behind skipped sweeps -- must exceed for the skip to be verified: under
      // plan 3 BCN_CONV_GUARD * tol, which proves that no skipped sweep passed (bcn_common.h); under plan 2 tol itself
      const real tolL = (A.conv_plan == 3) ? A.tol * real(BCN_CONV_GUARD) : A.tol;
      const float l2tol_u = __log2f((float)A.tol * 1.02f), l2tol_w = __log2f((float)tolL * 1.003f);
      int k_prev = -1, skip_left = 0;
      float l2u_prev = 0, l2w_prev = 0;
      constexpr int JMAX = 256;
      // what follows an evaluated sweep: the stop / overflow tests, then the number n of sweeps that need no residual
#define BCN_F4_AFTER_EVAL(EVEN)                                                                                                    \
        n_eval++;                                                                                                                  \
        /* the reference tests the sweep count FIRST: a solve that reaches sweep itmax + 1 overflows even if that sweep passes */  \
        /* (rayleigh.py:451-454, in front of the loop condition) */                                                                \
        if (itp > A.itmax) { status |= BCN_ST_ITMAX; break; }                                                                      \
        /* a landing (the sweep before this one was not evaluated) that does not clear tolL leaves the skipped sweeps unverified */ \
        if (!(err > A.tol) || (plan >= 2 && !A.verify_conv && itp >= 2 && k_prev != itp - 2 && !(err > tolL))) {                   \
          if (skip_left > 0) status |= BCN_ST_PLAN;                                                                                \
          break;                                                                                                                   \
        }                                                                                                                          \
        n = 0;                                                                                                                     \
        if (skip_left > 0) {                                                                                                       \
          skip_left--;                                                                                                             \
        } else if (plan > 0) {                                                                                                     \
          const float l2u = __log2f((float)eU), l2w = __log2f((float)err);                                                         \
          int j = 0;                                                                                                               \
          if (k_prev >= 0) {                                                                                                       \
            const float rg = 1.f / (float)(itp - 1 - k_prev);                                                                      \
            if (plan == 1) {                                                                                                       \
              const float room = l2u - l2tol_u, rho = (l2u - l2u_prev) * rg;                                                       \
              if (room > 0.f) j = (rho < 0.f) ? (int)fminf(room / -rho, (float)JMAX) : JMAX;                                       \
            } else {                                                                                                               \
              const float room = l2w - l2tol_w, rho = (l2w - l2w_prev) * rg;                                                       \
              int jw = 0;                                                                                                          \
              if (room > 0.f) jw = (rho < 0.f) ? (int)fminf(room / -rho, (float)JMAX) : JMAX;                                      \
              j = jw - 1 - (jw >> 4) + A.plan_overshoot;                                                                           \
              j = j > 0 ? j : 0;                                                                                                   \
            }                                                                                                                      \
          }                                                                                                                        \
          j = __builtin_amdgcn_readfirstlane(j);                                                                                   \
          l2u_prev = l2u; l2w_prev = l2w;                                                                                          \
          k_prev = itp - 1;                                                                                                        \
          if (A.verify_conv) skip_left = j; else n = j;                                                                            \
        }                                                                                                                          \
        if (n > A.itmax - itp) n = A.itmax - itp > 0 ? A.itmax - itp : 0;   /* the overflow test sits in the evaluated sweeps */   \
        if (EVEN) n &= ~1;   /* pairs only (one evaluation earlier than planned changes no result): the evaluated sweeps */       \
                             /* then alternate X -> Y, Y -> X whatever n is: every sweep's direction is a compile-time fact */
      int n = 0;
      // Speculative opening (spec_start > 0, extrapolating plans): with the previous timestep's count to go by, the solve
      // opens with plain sweeps up to spec_start / 8 of it and evaluates the residual there for the first time -- sweeps 1
      // and 2, evaluated only to start the plan, are two of a solve's eight evaluations.  If that first evaluation fails,
      // no earlier sweep passed; if it passes, the guess was too far and the solve is repeated without it, under the same plan
      // (whatever the plan: the results never depend on spec_start).
      bool spec_open = false;
      // (the smaller of the last TWO counts: one solve that took long -- the first after an action, a spike in a flow at
      //  rest -- is followed by short ones, and a guess from it costs a whole solve)
      const int prev_min = prev_itp < prev2_itp ? prev_itp : prev2_itp;
      // (compiled in where it measured a gain: rayleigh float32 with >= 2 rows per lane, 50x150 12.7 -> 12.2 ms; its mere
      //  presence cost mixing 100x200 2.4 % and rayleigh 300x50 6 % -- code size -- and mixing's counts drop too fast after an
      //  action for any guess: 60.1 -> 66.1 / 73.0 ms at spec_start 6 / 7)
      constexpr bool SPEC_OPEN = KIND == 0 && RPL >= 2 && std::is_same<real, float>::value;
      if (SPEC_OPEN && A.spec_start > 0 && plan > 1 && !A.verify_conv && prev_min >= 16) {
        n = ((prev_min * (A.spec_start <= 16 ? A.spec_start : 6)) >> 3) & ~1;   // (17 = the zone-aware opening of ns2d_fast_impl.h: 6/8 here)
        if (n > A.itmax) n = A.itmax & ~1;
        if (n > 0) { spec_open = true; BCN_F4_PAIRS(false) __syncthreads(); }
      }
      // Two shapes of the loop, picked by what measured faster (the register allocation of the evaluated sweeps decides;
      // mixing 100x200 float32: 63.4 against 67.6 ms; rayleigh 50x150 float64, every sweep evaluated: 27.9 against 27.2 ms):
      // float64: directions known at compile time, an even number of plain sweeps between two evaluations;
      // float32: the direction of an evaluated sweep dispatched on the parity, any number of plain sweeps -- the odd ones
      //          around the pairs.  A barrier closes every run of plain sweeps: the last one's late edge column, for the
      //          evaluated sweep that follows.
      if constexpr (BCN_F4_STATIC_DIR(real)) {
        for (;;) {
          BCN_F4_EVAL(false)
          BCN_F4_AFTER_EVAL(true)
          BCN_F4_PH(3)
          if (n > 0) { BCN_F4_PAIRS(true) __syncthreads(); }
          BCN_F4_PH(2)   // (stamps: phase 2 = the sweeps without residual, phase 3 = the evaluated ones)
          BCN_F4_EVAL(true)
          BCN_F4_AFTER_EVAL(true)
          BCN_F4_PH(3)
          if (n > 0) { BCN_F4_PAIRS(false) __syncthreads(); }
          BCN_F4_PH(2)
        }
      } else {
        for (;;) {
          if (par) BCN_F4_EVAL(true) else BCN_F4_EVAL(false)
          BCN_F4_AFTER_EVAL(false)
          BCN_F4_PH(3)
          if (n > 0) {
            if (par) { BCN_F4_PLAIN1(true) n--; }
            BCN_F4_PAIRS(false)
            if (n > 0) BCN_F4_PLAIN1(false)
            __syncthreads();
          }
          BCN_F4_PH(2)
        }
      }
#undef BCN_F4_AFTER_EVAL
      if (spec_open && k_prev < 0 && !(status & BCN_ST_ITMAX)) {   // the opening's first evaluation passed: too far
        prev_itp = 0;
        continue;
      }
      const bool late = plan >= 2 && itp >= 2 && k_prev != itp - 2 && !(status & BCN_ST_ITMAX);
      if (late) n_late++;
      if (!(late && A.conv_plan == 3)) break;
      plan = 1;
      n_redo++;
    }
#undef BCN_F4_PAIRS
#undef BCN_F4_PLAIN1
#undef BCN_F4_EVAL
#undef BCN_F4_CALL
    if (par) {   // state Y (column k in slot k + 1): back home
#pragma unroll
      for (int k = 0; k < R; k++)
#pragma unroll
        for (int r = 0; r < RPL; r++) Pv[k][r] = Pv[k + 1][r];
    }
    t_jac += __builtin_amdgcn_s_memtime() - tj0;
    BCN_F4_PH(3)
    if (A.sweeps && tid == 0) A.sweeps[(size_t)b * A.ndt_act + it] = itp;
    prev2_itp = prev_itp;
    prev_itp = (status & BCN_ST_ITMAX) ? 0 : itp;

    // phi -> LDS (natural layout): the corrector runs with lanes along x
#pragma unroll
    for (int k = 0; k < R; k++)
#pragma unroll
      for (int r = 0; r < RPL; r++)
        if (rowok[r] && (k < RL || !lastw)) W[(j0l + r) * P + i0 + k] = Pv[k][r];
    __syncthreads();

    // ---- p += phi incl. ghosts (rayleigh.py:219), corrector (rayleigh.py:460-464) ----
    f4_cells<NX, NW, UP>(w, tx, 1, NY,
      [&](int j, int i) {
        const int c = j * SX + i;
        // the ghost cells next to an edge cell take the same increment (an interior cell re-reads itself: no extra traffic)
        const int cgx = (i == 1) ? c - 1 : (i == NX) ? c + 1 : c;
        const int cgy = (j == 1) ? c - SX : (j == NY && KIND == 0) ? c + SX : c;
        return F4Corr<real>{p[c], us[c], vs[c], p[cgx], p[cgy]};
      },
      [&](const F4Corr<real>& q, int j, int i, bool ok) {
        if (!ok) return;
        const int c = j * SX + i;
        const real ph = W[j * P + i];
        p[c] = q.p + ph;
        if (i == 1) p[c - 1] = q.gx + ph;
        if (i == NX) p[c + 1] = q.gx + ph;
        if (j == 1) p[c - SX] = q.gy + ph;
        if (j == NY && KIND == 0) p[c + SX] = q.gy + ph;
        if (i >= 2) u[c] = q.us - A.dt * (ph - W[j * P + i - 1]) * A.rdx;
        if (j >= 2) v[c] = q.vs - A.dt * (ph - W[(j - 1) * P + i]) * A.rdy;
      });
    __syncthreads();
    BCN_F4_PH(4)

    // ---- transport (rayleigh.py:468-487), one block of rows at a time ----
#pragma unroll 1
    for (int blk = 0; blk < NBLK; blk++) {
      const int jb0 = 1 + blk * BR;
      const int jb1 = (jb0 + BR - 1 < NY) ? jb0 + BR - 1 : NY;
      // explicit part and the two in-place coefficients, lanes along x
      f4_cells<NX, NW, U>(w, tx, jb0, jb1,
        [&](int j, int i) {
          const int c = j * SX + i;
          return F4Tr<real>{u[c], u[c + 1], v[c], v[c + SX], S[c], S[c + 1], S[c + SX], S[c - 1]};
        },
        [&](const F4Tr<real>& q, int j, int i, bool ok) {
          const real uE = q.uE, uW = q.uW, vN = q.vN, vS = q.vS;
          const real Tc_ = q.Tc, TE = q.TE, TN = q.TN;
          real expl = A.ksc * ((TE - 2 * Tc_) * A.rdx2 + (TN - 2 * Tc_) * A.rdy2) -
                      (uE * real(0.5) * (TE + Tc_) - uW * real(0.5) * Tc_) * A.rdx -
                      (vN * real(0.5) * (TN + Tc_) - vS * real(0.5) * Tc_) * A.rdy;
          real aw = A.dt * (A.ksc * A.rdx2 + real(0.5) * uW * A.rdx);
          real as = A.dt * (A.ksc * A.rdy2 + real(0.5) * vS * A.rdy);
          real xv = Tc_ + A.dt * expl;
          if (i == 1) { xv = xv + aw * q.TW; aw = 0; }   // west ghost (old BC value) folded in: same operation order
          if (ok) {
            const int t = (j - jb0) * P + i;
            TX[t] = xv;
            TYZ[t] = F4YZ{aw, as};
          }
        });
      __syncthreads();
      BCN_F4_PH(5)
      // ordered part: one wave, lanes along x, walking the anti-diagonals d = i + j of the block
      if (w == 0) {
        // lane l: columns i = l CPL + 1 + q; t = j - jb0 of the cell a column has on the current diagonal.  A lone wave issues
        // an instruction every ~5 cycles, so the walk is branch-free and keeps its indices incrementally: a column that is
        // outside the block on this diagonal reads and writes the unused cell 0 of the arrays (column 0 of the first row) and
        // keeps its value; the coefficients are fetched two diagonals ahead of their use.
        const int len = jb1 - jb0;
        real x[CPL], a_c[CPL], y_c[CPL], z_c[CPL], a_n[CPL], y_n[CPL], z_n[CPL];
        int t[CPL];
        unsigned ib[CPL], sel_c[CPL], sel_n[CPL];   // BYTE offsets into TX (twice that into TYZ): no shift per access
        // (whether a column is inside the block on a diagonal is carried as sel != 0 -- an offset the walk holds anyway -- not
        //  as a lane mask: four masks across the loop are eight SGPRs, and when those ran short hipcc kept them as 0 / 1
        //  in VGPRs at four instructions per use)
        // LDS addresses as integers (the kernel's shared memory is all dynamic and starts at 0 -- checked where the kernel
        // starts): through `smem + ...` hipcc adds the array's base symbol, a literal 0, to every address in the walk
        typedef __attribute__((address_space(3))) real lds_real;
        constexpr unsigned TX0 = (unsigned)(G::FIXED * sizeof(real));
        constexpr unsigned TYZ0 = TX0 + (unsigned)(((BR * P + 1) & ~1) * sizeof(real));
        auto fetch = [&](real* a_, real* y_, real* z_, unsigned* sel_) {
#pragma unroll
          for (int q = 0; q < CPL; q++) {
            sel_[q] = ((unsigned)t[q] <= (unsigned)len) ? ib[q] : 0u;   // (cell index >= 1 inside the block)
            a_[q] = *reinterpret_cast<lds_real*>(TX0 + sel_[q]);
            const lds_real* yz = reinterpret_cast<lds_real*>(TYZ0 + 2u * sel_[q]);   // (one 64-bit access: F4YZ is aligned)
            y_[q] = yz[0]; z_[q] = yz[1];
            t[q] += 1;
            ib[q] += (unsigned)(P * sizeof(real));
          }
        };
#pragma unroll
        for (int q = 0; q < CPL; q++) {
          const int i = tx * CPL + 1 + q;
          x[q] = (i <= NX) ? S[(jb0 - 1) * SX + i] : real(0);   // row below the block: the south ghost or the rows done before
          t[q] = (i <= NX) ? 1 - i : -(1 << 20);
          ib[q] = (unsigned)((t[q] * P + i) * (int)sizeof(real));
        }
        fetch(a_c, y_c, z_c, sel_c);
        fetch(a_n, y_n, z_n, sel_n);
        auto advance = [&](real* a_, real* y_, real* z_, unsigned* sel_) {
          // one diagonal with the coefficients in (a_, y_, z_), which are then refilled for the diagonal after the next
          const real wl = dpp<0x138, 0xf, 0xf, true>(real(0), x[CPL - 1]);   // last column of the lane to the left
          real xn[CPL];
#pragma unroll
          for (int q = 0; q < CPL; q++) {
            const real wv = (q == 0) ? wl : x[q - 1];
            xn[q] = z_[q] * x[q] + (y_[q] * wv + a_[q]);
          }
#pragma unroll
          for (int q = 0; q < CPL; q++) {
            x[q] = (sel_[q] != 0) ? xn[q] : x[q];
            // (stored BEFORE the refill: the registers of a_ and sel_ are free for it -- no copies at the loop's end)
            *reinterpret_cast<lds_real*>(TX0 + sel_[q]) = xn[q];
          }
          fetch(a_, y_, z_, sel_);
        };
        const int nsteps = len + NX;
#pragma unroll 1
        for (int st = 0; st < nsteps; st += 2) {
          advance(a_c, y_c, z_c, sel_c);
          advance(a_n, y_n, z_n, sel_n);   // (an odd count runs one diagonal past the block: no column is inside it)
        }
      }
      __syncthreads();
      BCN_F4_PH(6)
      for (int j = jb0 + w; j <= jb1; j += NW)
        for (int i = 1 + tx; i <= NX; i += BCN_WAVE) S[j * SX + i] = TX[(j - jb0) * P + i];
      __syncthreads();
    }
  }

  if (A.cyc && tid == 0) {   // a replica's chunks run one after the other (hand-off through the progress word): plain +=
    A.cyc[(size_t)b * 4 + 0] += t_jac;
    A.cyc[(size_t)b * 4 + 1] += __builtin_amdgcn_s_memtime() - t_begin;
#ifdef BCN_F4_STAMP
    A.cyc[(size_t)b * 4 + 2] += t_ph;
    A.cyc[(size_t)b * 4 + 3] += n_eval;
#else
    A.cyc[(size_t)b * 4 + 2] += n_late;   // stops the extrapolating plan did not foresee
    A.cyc[(size_t)b * 4 + 3] += n_redo;   // solves repeated under the proven plan (conv_plan 3)
#endif
  }
  if (last_chunk) {
    ns2d_finish<real, NT>(A, b, A.u + off, A.v + off, A.S + off, status, red);
  } else if (tid == 0) {
    A.status[b] = status;
  }
}

template <typename real, int NX, int NY, int R, int RPL, int KIND, bool EQ>
__global__ __launch_bounds__((Fast4Geom<NX, NY, R, RPL>::NT)) void ns2d_fast4_step(NS2DArgs<real> A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = blockIdx.x;
  if (A.mask && !A.mask[b]) return;
  fast4_unit<real, NX, NY, R, RPL, KIND, EQ>(A, b, 0, A.ndt_act, true, true, smem);
}

// ticketed chunk scheduler (ns2d_sched.h): persistent workgroups draw (chunk, replica) units
template <typename real, int NX, int NY, int R, int RPL, int KIND, bool EQ>
__global__ __launch_bounds__((Fast4Geom<NX, NY, R, RPL>::NT)) void ns2d_fast4_sched(NS2DArgs<real> A, SchedCtl* ctl, int batch, int nchunk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned int* s_words = reinterpret_cast<unsigned int*>(reinterpret_cast<real*>(smem) + 128);
  ns2d_sched_loop<real>(A, ctl, batch, nchunk, s_words, [&](int b, int it0, int it1, bool first, bool last) {
    fast4_unit<real, NX, NY, R, RPL, KIND, EQ>(A, b, it0, it1, first, last, smem);
  });
}

template <typename real, int NX, int NY, int R, int RPL, int KIND, bool EQ>
int launch_fast4_eq(const NS2DArgs<real>& a, int batch, hipStream_t s) {
  using G = Fast4Geom<NX, NY, R, RPL>;
  const size_t lds = (size_t)G::lds_elems(sizeof(real)) * sizeof(real);
  if (!f4_fields_ok(a, (size_t)G::NCELL)) return BCN_ERR_UNSUPPORTED;   // (the caller falls back to the generic kernel)
  NS2DArgs<real> c = a;
  if (!c.sweeps) c.sweeps = c.sweeps_int;
  const SchedParams sp = ns2d_sched_params(a);
  const int q = sp.q_set ? sp.q : 20;   // timesteps per chunk
  if (sp.mode == 2 && batch > sp.grid && a.ndt_act >= 2 * q && a.sched_ctl) {
    // more replicas than CUs: persistent workgroups share the replicas' timesteps chunk by chunk, so that a CU is not
    // stuck with the sum of whichever two replicas' sweep counts it was dealt
    auto ks = ns2d_fast4_sched<real, NX, NY, R, RPL, KIND, EQ>;
    static unsigned long long set2 = 0;
    if (ns2d_first_on_device(set2)) BCN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ks), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    c.sched_nbig = 0;
    c.sched_q = q;
    BCN_HIP(hipMemsetAsync(a.sched_ctl, 0, a.sched_bytes, s));
    hipLaunchKernelGGL(ks, dim3(sp.grid), dim3(G::NT), lds, s, c, static_cast<SchedCtl*>(a.sched_ctl), batch, a.ndt_act / q);
    BCN_HIP(hipGetLastError());
    if (a.host) a.host->launched = "ns2d_fast4_sched";
    return BCN_OK;
  }
  if (a.sched_ctl) BCN_HIP(hipMemsetAsync(a.sched_ctl, 0, a.sched_bytes, s));   // cycle counters
  auto k = ns2d_fast4_step<real, NX, NY, R, RPL, KIND, EQ>;
  static unsigned long long set = 0;
  if (ns2d_first_on_device(set)) BCN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k, dim3(batch), dim3(G::NT), lds, s, c);
  BCN_HIP(hipGetLastError());
  if (a.host) a.host->launched = "ns2d_fast4_step";
  return BCN_OK;
}

template <typename real, int NX, int NY, int R, int RPL, int KIND>
int launch_fast4(const NS2DArgs<real>& a, int batch, hipStream_t s) {
  // dx == dy (every reference configuration): one multiply per cell instead of two
  if (a.cx == a.cy) return launch_fast4_eq<real, NX, NY, R, RPL, KIND, true>(a, batch, s);
  return launch_fast4_eq<real, NX, NY, R, RPL, KIND, false>(a, batch, s);
}

}  // namespace
