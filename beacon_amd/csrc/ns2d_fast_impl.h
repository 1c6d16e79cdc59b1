// ns2d_fast_impl.h -- register-resident CDNA4 action step, one row per lane (ny <= 64): kernel templates.
// Instantiated by ns2d_fast.hip (the grids built into libbeacon_hip.so) and by ns2d_jit.hip (any other grid, compiled
// on demand by beacon_amd/jit.py).
//
// One workgroup of NW = NX/R waves per replica runs the whole action step on chip:
//   * lanes run along y (lane l <-> row j = l+1, NY <= 64), wave w owns the R columns
//     i = w*R+1 .. w*R+R, so every thread holds R consecutive x-cells of row j in VGPRs;
//   * p, u*, v*, the Poisson rhs and phi live in registers for the whole step; u, v, T live
//     in LDS in [i][j] order (lanes consecutive in j: bank-conflict free, also for the
//     skewed accesses of the transport sweep); HBM is read once on entry, written once on exit;
//   * Jacobi sweep: x-neighbours are the thread's own registers, y-neighbours come from the
//     adjacent lanes by DPP wave shifts (whose boundary lanes give the Neumann ghost for
//     free), only the two strip-edge columns of each wave go through a double-buffered LDS
//     exchange -- ONE barrier per sweep, which also carries the convergence test: per-wave
//     DPP-reduced partial sums of the squared increment, combined in a fixed order by every
//     wave (deterministic sweep counts);
//   * transport (the reference's in-place sweep, rayleigh.py:468-487): explicit part of all
//     cells in parallel, then ONE wave walks the nx+ny-1 anti-diagonals with the two
//     dependent FMAs per cell in registers (west = own previous value, south = DPP from the
//     lane below), coefficients prefetched from LDS PD diagonals ahead.
//
// Semantics are those of ns2d_generic.hip (same citations); tests compare both variants.
#pragma once
#include <stdlib.h>

#include <type_traits>

#include "bcn_dpp.h"
#include "ns2d.h"
#include "ns2d_device.h"
#include "ns2d_sched.h"


#ifndef BCN_R128
#define BCN_R128 16
#endif
#ifndef BCN_R50
#define BCN_R50 5     // columns per lane of the 50x50 kernels
#endif
#ifndef BCN_GFD
#define BCN_GFD 2   // float64 128x64: 1 = u, v, T in global scratch, 2 = u, v in LDS and T in global scratch
#endif
#ifndef BCN_PDG
#define BCN_PDG 8    // global fields, diagonals ahead.  Round 3 (T through flat pointers): 4 / 8 / 12 -> 52.9 / 52.1 / 51.3 ms per step
                     // (rayleigh 128x64 float64); round 4 (global pointers): 4 / 8 / 10 / 12 / 16 / 20 -> 53.0 / 51.2 / 51.9 / 52.0 / 51.9 / 61.0.
                     // (measured depths only: an intermediate round-4 build with generic pointers stopped converging at a depth of 6)
#endif
static_assert(BCN_PDG == 4 || BCN_PDG == 8 || BCN_PDG == 10 || BCN_PDG == 12 || BCN_PDG == 16 || BCN_PDG == 20,
              "BCN_PDG: only the prefetch depths that were measured AND verified against the oracle (4, 8, 10, 12, 16, 20)");
#ifndef BCN_PDF
#define BCN_PDF 8    // fields in LDS: diagonals per block of the transport wave (two register sets: it runs PDF..2 PDF diagonals ahead;
                     // measured 4 / 6 / 8 / 12 / 16: 26.2 / 25.7 / 24.8 / 25.3 / 26.3 k cycles per timestep outside the solve)
#endif
#ifndef BCN_R128D
#define BCN_R128D 16   // columns per lane of the float64 128x64 kernel
#endif



namespace {

using namespace bcn_dpp;

// GF ("global fields", float64 at 128x64: 3 x 68.6 KB do not fit LDS): 1 = u, v, T (with the same pads) live in
// a per-workgroup global scratch and LDS holds only the exchange buffers; 2 = u, v in LDS without pads (the
// transport wave clamps its skewed reads instead) and only T, with its pads, in the global scratch.
template <int NX, int NY, int R, int GF = 0>
struct FastGeom {
  static_assert(NY <= 64, "lanes run along y");
  static constexpr int NW = (NX + R - 1) / R;
  static constexpr int RL = NX - (NW - 1) * R;   // columns of the last wave (a second instantiation of the body when < R)
  static_assert(NW <= 16 && RL >= 3 && RL <= R, "at most 16 waves, at least 3 columns each");
  static constexpr int NT = NW * 64;
  static constexpr int SY = NY + 2;
  static constexpr int SX = NX + 2;
  // +16: lanes >= NY read (never write) past the array; a multiple of 64 elements, so that the transport wave fetches
  // u and v of a cell (the same index in two consecutive arrays) with ONE ds_read2st64_b32
  // GF == 2 with a narrow last strip: ONE body serves every strip (fast_body: DEADC), the last one R columns wide like the
  // others with RL of them live; its dead columns read (never write) up to R - RL + 1 columns past the east ghost column
  static constexpr int DEADPAD = (GF == 2 && RL != R) ? (R - RL + 2) * SY : 0;
  static constexpr int SZ = (SX * SY + 16 + DEADPAD + 63) / 64 * 64;
  static constexpr int PD = GF ? BCN_PDG : BCN_PDF;   // transport prefetch depth (diagonals); deeper for global fields
  // LDS map (elements): [ exchange 2*NW*2*64 | errp 128 | sact 64 | red 32 | sched 16 | .. FRONT ) U V T [ BACK )
  // The transport wave reads cell (t-lane+1, lane+1) for every lane without range checks: columns
  // -62..NX+NY+PD fall into FRONT / the neighbouring arrays / BACK, always inside this allocation.
  // columns per strip in the exchange buffer: 0, 1, R-2, R-1 (depth-2 halos: two sweeps per barrier), or the two edge
  // columns only where LDS is short (GF == 2: two float64 fields in LDS)
  static constexpr int XC = (GF == 2) ? 2 : 4;
  static constexpr int EXCH = 2 * NW * XC * 64;             // [2 buffers][NW][XC][64]
  static constexpr int MISC = EXCH + 224 + 16;              // + 16: scheduler words (ns2d_fast_sched)
  static constexpr int FRONT = ((MISC > 63 * SY + 1 ? MISC : 63 * SY + 1) + 15) / 16 * 16;
  static constexpr int BACK = (NY + 3 * PD + 2) * SY;   // the transport wave prefetches two blocks of PD diagonals ahead
  static constexpr int FRONTG = (63 * SY + 1 + 15) / 16 * 16;          // front pad of the global variant
  static constexpr int MISCA = (MISC + 15) / 16 * 16;
  static constexpr size_t base_elems() {
    return GF == 1 ? (size_t)MISCA : GF == 2 ? (size_t)MISCA + 2 * (size_t)SZ : (size_t)FRONT + 3 * (size_t)SZ + BACK;
  }
  // While wave 0 walks the ordered part of the transport step, the other waves compute the next timestep's predictor -- and,
  // where LDS has room for it, wave 0's strip as well: p of strip 0 in, u*, and the v* increment without buoyancy out,
  // [3][R][64] elements behind everything else (fast_body)
  static constexpr size_t SCR = 3 * (size_t)R * 64;
  template <typename real> static constexpr bool offload() {
    return NW >= 3 && (R + 3) / 4 <= NW - 1 && (base_elems() + SCR) * sizeof(real) <= 160 * 1024;
  }
  template <typename real> static constexpr size_t lds_bytes() {
    return (base_elems() + (offload<real>() ? SCR : 0)) * sizeof(real);
  }
  static constexpr size_t scratch_elems() {
    return GF == 1 ? (size_t)FRONTG + 3 * (size_t)SZ + BACK : GF == 2 ? (size_t)FRONTG + SZ + BACK : 0;
  }
};

// Ordered part of the transport step, run by ONE wave (kept out of line: its unrolled, software-pipelined loops would
// otherwise inflate the register pressure of the whole kernel).  At step t lane l works on cell (i, j) = (t - l + 1, l + 1),
// element cb + t*SY of the [i][j] arrays: S' = A + aW S'(i-1,j) + aS S'(i,j-1) with the west value in a register and the
// south value from the lane below by DPP.  Reads are unmasked (see the LDS map), only writes are.
//
// float32 with the fields in LDS (GF == 0): the south term is ONE v_fmac_f32_dpp -- tn = (A + aW tp) + aS * tp(lane-1);
// lane 0 has no lane below and keeps the first sum: its south value is the ghost row T[i][0], which the chain never
// updates, so its term is folded into A beforehand.  Blocks of four diagonals with two register sets: a block first
// requests the next block's A, u, v (two diagonals are SY elements apart: ds_read2_b32; V starts SZ elements behind U),
// then computes its four steps from registers and stores them pairwise.  A lone wave issues one instruction per ~4.6
// cycles whatever its kind: 14 -> 10 instructions per steady step; the ragged first and last thirds of the sweep carry three
// mask instructions more.
typedef float v2f __attribute__((ext_vector_type(2)));
// The wait state in front of the chain's DPP step.  A VALU write of a VGPR needs two wait states before a DPP read of it; the DPP
// source is the previous step's result, and between its last writer (the previous v_fmac_f32_dpp, or the v_cndmask of the masked
// phases) and this DPP read there is always the plain v_fmac that forms A + aW * west, which reads it too: ONE more wait state
// suffices.  s_nop 1 -> s_nop 0: 24.27 k -> 23.68 k cycles per timestep outside the solve (3 cycles x 191 steps of a lone wave; round 6).
#ifndef BCN_CHAIN_NOP
#define BCN_CHAIN_NOP "s_nop 0\n\t"
#endif
template <int NX, int NY, int R>
__device__ __attribute__((noinline)) void transport_chain_f32(float* Tl, const float* Ul, float* dummy, float c0x, float c1x,
                                                              float c0y, float c1y) {
  using G = FastGeom<NX, NY, R, 0>;
  constexpr int SY = G::SY, PD = G::PD, NSTEP = NX + NY - 1;
  static_assert(PD % 2 == 0, "blocks of an even number of diagonals (packed coefficient fmas)");
  const int lane = threadIdx.x & 63;
  const int j = lane + 1;
  const bool active = lane < NY;
  const int cb = (1 - lane) * SY + j;
  float* Tb = Tl + cb;
  // row 1: A += aS * T[i][0] (LDS accesses of one wave are in program order)
  for (int i = 1 + lane; i <= NX; i += 64) Tl[i * SY + 1] += (c0y + c1y * Ul[i * SY + 1 + G::SZ]) * Tl[i * SY];
  float tp = Tl[0 * SY + j];                        // west ghost
  float aA[PD], uA[PD], vA[PD], aB[PD], uB[PD], vB[PD];
  // Tc / Uc: this lane's element of the first diagonal of the current trip (T and A share an array, V sits SZ behind U);
  // OFF: diagonal offset inside the trip (compile time: every LDS access of a trip is base register + immediate)
  // (LDS pointers, 32 bit: as generic pointers of this out-of-line function every access of the rolled loops paid a
  // 64-bit add and a null-checked address-space conversion)
  typedef __attribute__((address_space(3))) float lds_f;
  lds_f* Tc = (lds_f*)Tb;
  const lds_f* Uc = (const lds_f*)(Ul + cb);
  lds_f* const dummyL = (lds_f*)dummy;
#define BCN_LOAD(RA, RU, RV, OFF)                                                             \
  {                                                                                           \
    const lds_f* const Tq = Tc + (OFF) * SY;                                                  \
    const lds_f* const Uq = Uc + (OFF) * SY;                                                  \
    const lds_f* const Vq = Uq + G::SZ;                                                       \
    _Pragma("unroll") for (int q = 0; q < PD; q++) { RA[q] = Tq[q * SY]; RU[q] = Uq[q * SY]; RV[q] = Vq[q * SY]; } \
  }
  // MASK 0: every lane inside, or running on behind the domain; 1: lanes <= t; 2: lanes > t - NX; 3: both tests; 4: as 1
  // for NY == 64, by value
#define BCN_BLOCK(RA, RU, RV, NA, NU, NV, OFF, MASK)                                          \
  {                                                                                           \
    BCN_LOAD(NA, NU, NV, (OFF) + PD)                                                          \
    lds_f* const Tq = Tc + (OFF) * SY;                                                        \
    float tq[PD];                                                                             \
    bool okq[PD];                                                                             \
    /* the coefficients of two diagonals per packed fma: a lone wave issues a v_pk_fma_f32 in the time of a v_fma_f32 */ \
    v2f awp[PD / 2], asp[PD / 2];                                                             \
    _Pragma("unroll") for (int q = 0; q < PD / 2; q++) {                                      \
      const v2f ru = {RU[2 * q], RU[2 * q + 1]}, rv = {RV[2 * q], RV[2 * q + 1]};             \
      awp[q] = c0x + c1x * ru;                                                                \
      asp[q] = c0y + c1y * rv;                                                                \
    }                                                                                         \
    _Pragma("unroll") for (int q = 0; q < PD; q++) {                                          \
      const int t = t0 + (OFF) + q;                                                           \
      const float aw = awp[q / 2][q & 1], as = asp[q / 2][q & 1];                             \
      float t1 = RA[q] + aw * tp;                                                             \
      asm volatile(BCN_CHAIN_NOP "v_fmac_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf" \
                   : "+v"(t1) : "v"(tp), "v"(as));                                            \
      okq[q] = (MASK == 0) || (MASK == 4 ? (lane <= t) : (active && (MASK != 2 ? (lane <= t) : true) && \
                               (MASK != 1 ? (lane > t - NX && t < NSTEP) : true)));           \
      tp = okq[q] ? t1 : tp;                                                                  \
      /* MASK 4 (every row active, lanes <= t): a lane in front of the domain stores the value it loaded back (its cell \
         lies in column 0 or in the array in front of T, which nothing writes during the sweep): the stores keep their \
         immediate offsets and pair up, instead of a selected address per step */            \
      tq[q] = (MASK == 4 && !okq[q]) ? RA[q] : t1;                                            \
    }                                                                                         \
    if (MASK == 0 || MASK == 4) {                                                             \
      _Pragma("unroll") for (int q = 0; q < PD; q++) Tq[q * SY] = tq[q];                      \
    } else {                                                                                  \
      _Pragma("unroll") for (int q = 0; q < PD; q++) { lds_f* dst = okq[q] ? Tq + q * SY : dummyL; *dst = tq[q]; } \
    }                                                                                         \
  }
  // The loops stay rolled (a trip = two blocks of four diagonals, two register sets): unrolled, the sweep is 18 KB of
  // straight-line code that the lone wave streams through the instruction cache once per timestep -- next to the other
  // waves' predictor code (fast_body) that doubled its time.
#define BCN_CHAIN2(T0, T1, MASK)                                                              \
  _Pragma("nounroll") for (int t0 = (T0); t0 < (T1); t0 += 2 * PD) {                          \
    BCN_BLOCK(aA, uA, vA, aB, uB, vB, 0, MASK)                                                \
    BCN_BLOCK(aB, uB, vB, aA, uA, vA, PD, MASK)                                               \
    Tc += 2 * PD * SY;                                                                        \
    Uc += 2 * PD * SY;                                                                        \
  }
  BCN_LOAD(aA, uA, vA, 0)
  // lanes 0..NY-1 are all inside the domain for t in [NY-1, NX); phase bounds are multiples of 2 PD
  constexpr int P2 = 2 * PD;
  constexpr int TA2 = ((NY - 1 + P2 - 1) / P2) * P2, TB2 = (NX / P2) * P2, TE2 = ((NSTEP + P2 - 1) / P2) * P2;
  if constexpr (TA2 <= TB2 && NY == 64) {
    // Behind the domain (i > NX) a lane simply runs on: its garbage lands in the pad behind T -- and, in its first step
    // out, in the east ghost column, which is saved here and put back (no masks at all in the last third of the sweep)
    const float east = Tl[(NX + 1) * SY + j];
    BCN_CHAIN2(0, TA2, 4)
    BCN_CHAIN2(TA2, TE2, 0)
    Tl[(NX + 1) * SY + j] = east;
  } else {
    BCN_CHAIN2(0, TE2, 3)
  }
#undef BCN_CHAIN2
#undef BCN_BLOCK
#undef BCN_LOAD
}

// The same for float64 and for fields in the global scratch (GF): plain form -- the south value by a DPP move (lane 0: none, the
// ghost row's term is folded into the explicit part of row 1 first), diagonals prefetched PD steps ahead one by one.
template <typename real, int NX, int NY, int R, int GF>
__device__ __attribute__((noinline)) void transport_chain(real* Tl, const real* Ul, const real* Vl, real* dummy,
                                                          real c0x, real c1x, real c0y, real c1y) {
  using G = FastGeom<NX, NY, R, GF>;
  constexpr int SY = G::SY, PD = G::PD, NSTEP = NX + NY - 1;
  const int lane = threadIdx.x & 63;
  const int j = lane + 1;
  const bool active = lane < NY;
  const int cb = (1 - lane) * SY + j;
  // Every field through a pointer of its OWN address space -- T: LDS (GF == 0) or the global scratch (the masked lanes' sink
  // then lies in the scratch's front pad); u, v: the global scratch with GF == 1, LDS otherwise.  As generic pointers of this
  // out-of-line function every access of the global scratch was a flat_load / flat_store, which count on BOTH wait counters
  // (the waits of the LDS reads also waited for the prefetch of T), and every LDS access paid a 64-bit add and a
  // null-checked address-space conversion.
  typedef typename std::conditional<GF != 0, __attribute__((address_space(1))) real, __attribute__((address_space(3))) real>::type treal;
  typedef typename std::conditional<GF == 1, __attribute__((address_space(1))) real, __attribute__((address_space(3))) real>::type ureal;
  treal* const Tb = (treal*)(Tl + cb);
  treal* const dummy_t = GF != 0 ? (treal*)(Tl - G::FRONTG) : (treal*)dummy;
  const ureal* const Ug = (const ureal*)Ul;
  const ureal* const Vg = (const ureal*)Vl;
  auto ldu = [&](int t) -> real { const int x = cb + t * SY; return GF == 2 ? Ug[x < 0 ? 0 : (x >= G::SZ ? G::SZ - 1 : x)] : Ug[x]; };
  auto ldv = [&](int t) -> real { const int x = cb + t * SY; return GF == 2 ? Vg[x < 0 ? 0 : (x >= G::SZ ? G::SZ - 1 : x)] : Vg[x]; };
  // The south ghost row folded into the explicit part of row 1 beforehand, as transport_chain_f32 does: A[i][1] += aS(i, 1) T[i][0].
  // Lane 0 then has no south term (DPP with zero fill) and a step loads ONE value of T instead of two -- with T in the global
  // scratch a load less per step of a lone wave: float64 128x64 104.9 k -> 96.9 k cycles per timestep outside the solve (round 6;
  // prefetch depths 4 / 8 / 12 / 16 measured again with it: 47.5 / 45.7 / 46.5 / 46.8 ms).
  for (int i = 1 + lane; i <= NX; i += 64) {
    const int x = i * SY + 1;
    ((treal*)Tl)[x] = ((treal*)Tl)[x] + (c0y + c1y * Vg[x]) * ((treal*)Tl)[x - 1];
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // (the first diagonals read what other lanes of this wave just stored)
  real ra[PD], ru[PD], rv[PD];
#pragma unroll
  for (int q = 0; q < PD; q++) { ra[q] = Tb[q * SY]; ru[q] = ldu(q); rv[q] = ldv(q); }
  real tp = Tl[0 * SY + j];                        // west ghost
  // MASK 0: every lane inside (steady state); 1: lanes <= t; 2: lanes > t - NX; 3: both tests
#define BCN_CHAIN(T0, T1, MASK)                                                             \
  _Pragma("nounroll") for (int t0 = (T0); t0 < (T1); t0 += PD) {                            \
    _Pragma("unroll") for (int q = 0; q < PD; q++) {                                        \
      const int t = t0 + q;                                                                 \
      const real s = dpp<0x138, 0xf, 0xf, true>(real(0), tp);   /* the lane below's value; lane 0: zero */ \
      const real aw = c0x + c1x * ru[q], as = c0y + c1y * rv[q];                            \
      const real tn = ra[q] + aw * tp + as * s;                                             \
      if (MASK == 0) {                                                                      \
        tp = tn;                                                                            \
        Tb[t * SY] = tn;                                                                    \
      } else {                                                                              \
        const bool ok = active && (MASK != 2 ? (lane <= t) : true) &&                       \
                        (MASK != 1 ? (lane > t - NX && t < NSTEP) : true);                  \
        tp = ok ? tn : tp;                                                                  \
        treal* dst = ok ? Tb + t * SY : dummy_t;                                            \
        *dst = tn;                                                                          \
      }                                                                                     \
      ra[q] = Tb[(t + PD) * SY];                                                            \
      ru[q] = ldu(t + PD);                                                                  \
      rv[q] = ldv(t + PD);                                                                  \
    }                                                                                       \
  }
  // lanes 0..NY-1 are all inside the domain for t in [NY-1, NX); phase bounds are multiples of PD
  constexpr int TA = ((NY - 1 + PD - 1) / PD) * PD;      // first steady step (rounded up)
  constexpr int TB = (NX / PD) * PD;                      // end of the steady phase (rounded down)
  if constexpr (TA <= TB) {
    BCN_CHAIN(0, TA, 1)
    if (NY == 64) { BCN_CHAIN(TA, TB, 0) } else { BCN_CHAIN(TA, TB, 1) }
    BCN_CHAIN(TB, NSTEP, 2)
  } else {                                               // (nearly) square grid: no steady phase
    constexpr int TL = ((NY - 1) / PD) * PD, TH = ((NX + PD - 1) / PD) * PD;
    BCN_CHAIN(0, TL, 1)
    BCN_CHAIN(TL, TH, 3)
    BCN_CHAIN(TH, NSTEP, 2)
  }
#undef BCN_CHAIN
}

// One unit of work: timesteps [it_begin, it_end) of replica b (state HBM -> chip -> HBM).
// R0: columns of a full strip (the geometry), R = RW: columns of THIS wave's strip (R0, or RL for the last wave)
template <typename real, int NX, int NY, int R0, int RW, int KIND, bool EQ, int GF>
__device__ __forceinline__ void fast_body(const NS2DArgs<real>& A, const int w, const int b, const int it_begin, const int it_end,
                                          const bool first_chunk, const bool last_chunk, char* smem) {
  using G = FastGeom<NX, NY, R0, GF>;
  constexpr int R = RW;
  constexpr int NW = G::NW, NT = G::NT, SY = G::SY, SX = G::SX, SZ = G::SZ, PD = G::PD;
  real* exch = reinterpret_cast<real*>(smem);  // [2][NW][2][64]
  real* errp = exch + G::EXCH;                 // [2][4][16]: partials of the reference norm / the unweighted norm (x 2 sweeps)
  real* sact = errp + 128;                     // [64]
  real* red = sact + 64;                       // [32]
  real* gscr = GF ? A.fscr + (size_t)blockIdx.x * A.fscr_stride + G::FRONTG : nullptr;
  real* Ul = GF == 1 ? gscr : GF == 2 ? exch + G::MISCA : exch + G::FRONT;
  real* Vl = Ul + SZ;
  real* Tl = GF == 2 ? gscr : Vl + SZ;

  const int tid = threadIdx.x, lane = tid & 63;
  int j = lane + 1;   // GF: laundered at the phase boundaries (hipcc would hoist and spill a timestep's 64-bit addresses)
  const bool active = lane < NY;
  // ONE body for every strip (float64 with T in the global scratch; VERDICT r05 item 2: the seam between two instantiations of
  // this body in one kernel is where round 4's wrong float64 kernel had its wrong word).  The last strip is R0 columns wide like
  // the others; its first RL columns exist, the others are DEAD: their registers hold finite values that nothing stores, their
  // u*, v*, rhs are zero, the live east edge (column RL - 1) sees its Neumann ghost as a mirror cell in column RL, and the
  // residual leaves them out.  LIVE(k): column k of THIS wave exists (compile time for k < RL and for every other grid).
  constexpr int RL = G::RL;
  constexpr bool DEADC = GF == 2 && RW == R0 && RL != R0;
  const bool lastw = DEADC && w == NW - 1;
#define LIVE(k) (!DEADC || (k) < RL || !lastw)
  const int i0 = w * R0 + 1;
  const size_t off = (size_t)b * A.ncell;
  real* __restrict__ gu = A.u + off;
  real* __restrict__ gv = A.v + off;
  real* __restrict__ gp = A.p + off;
  real* __restrict__ gS = A.S + off;
  constexpr int XC = G::XC, XL = XC - 1;   // XL: slot of the strip's last column
  // exchange buffer [2 buffers][NW][XC columns of the strip][64 lanes]: column-major, one 32-bit (64-bit) access per column.
  // (Measured in round 4 and dropped: lane-major with one 128-bit store and two 64-bit loads per double sweep -- 750 against 738
  // cycles per sweep on the bench workload: the four narrow stores drain behind the interior cells, the wide one does not.)
  auto exl = [&](int buf, int wave, int which) -> real& { return exch[((buf * NW + wave) * XC + which) * 64 + lane]; };

  // ---- load: HBM [j][i] -> LDS [i][j]; p -> registers --------------------------------------
  for (int c = tid; c < SX * SY; c += NT) {
    const int jj = c / SX, ii = c - jj * SX;
    Ul[ii * SY + jj] = gu[c];
    Vl[ii * SY + jj] = gv[c];
    Tl[ii * SY + jj] = gS[c];
  }
  for (int c = SX * SY + tid; c < SZ; c += NT) { Ul[c] = 0; Vl[c] = 0; Tl[c] = 0; }
  if (tid < 128) errp[tid] = 0;
  real p[R];
#pragma unroll
  for (int k = 0; k < R; k++) p[k] = (active && LIVE(k)) ? gp[j * SX + i0 + k] : real(0);

  // ---- action conditioning (rayleigh.py:162-171); later chunks reuse the conditioned vector ----
  if (!first_chunk) {
    if (tid < A.n_sgts) sact[tid] = A.a_last[(size_t)b * A.n_sgts + tid];
  } else {
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
  }
  __syncthreads();

  const real dt = A.dt, rdx = A.rdx, rdy = A.rdy, rdx2 = A.rdx2, rdy2 = A.rdy2;
  const real cx = A.cx, cy = A.cy;
  // err weight of this lane's row: ghosts copy their interior neighbour (rayleigh.py:432-449)
  const real wl = active ? real(1) + (j == 1 ? 1 : 0) + ((j == NY && KIND == 0) ? 1 : 0) : real(0);
  const real fW = (active && w == 0) ? real(1) : real(0);
  const real fE = (active && w == NW - 1) ? real(1) : real(0);
  // (one body, last strip: its register column R - 1 is dead -- weight 0 --, and the east ghost column counts with column RL - 1: fEl)
  const real cW = wl + fW, cE = lastw ? real(0) : wl + fE;   // weights of the strip's first / last column (ghost columns included)
  const real fEl = lastw ? fE : real(0);
  // coefficient of the lane's own value from the y-ghosts (bottom: always Neumann; top: rayleigh)
  const real cBy = cy * (real)((lane == 0 ? 1 : 0) + ((lane == NY - 1 && KIND == 0) ? 1 : 0));
  const real actf = active ? real(1) : real(0);

  int status = 0;
  int xb = 0;
#ifdef BCN_STAMP
  const unsigned long long kt0 = __builtin_amdgcn_s_memtime(), kr0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long seg[7] = {0, 0, 0, 0, 0, 0, 0};
  unsigned long long tl = kt0;
#define BCN_PH(x) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t__ = __builtin_amdgcn_s_memtime(); seg[x] += t__ - tl; tl = t__; __builtin_amdgcn_sched_barrier(0); }
#else
#define BCN_PH(x)
#endif
  if (!first_chunk) status = A.status[b];   // a replica that overflowed stays stopped (status is never NULL: capi.hip)
  const unsigned long long cyc_u0 = __builtin_amdgcn_s_memtime();
  unsigned long long cyc_j = 0;
  // sweeps of the previous timestep's solve, kept in LDS (red[20]; 0 = unknown: the first timestep of an action step
  // starts without a guess) -- see "Speculative jump" below
  real* const prev_sweeps = red + 20;
  // float32 only: the float64 default is the proven plan, which evaluates every ~4th sweep anyway and needs the unweighted
  // norm above its threshold behind a jump (measured: no gain, and the code costs the float64 instantiation 6 %)
  constexpr bool SPEC = std::is_same<real, float>::value;
#if !defined(BCN_STAMP) && !defined(BCN_DBG_NCHK)
  if (tid == 0) prev_sweeps[0] = (it_begin > 0 && A.sweeps) ? (real)A.sweeps[(size_t)b * A.ndt_act + it_begin - 1] : real(0);
#else
  if (tid == 0) prev_sweeps[0] = 0;
#endif
  // prev_sweeps[1]: this solve runs under the proven plan (a repeat, conv_plan 3); [2]: solves whose stop sweep the
  // extrapolating plan could not verify ("late stops"); [3]: solves repeated (speculative jump too far, or conv_plan 3)
  if (tid == 0) { prev_sweeps[1] = 0; prev_sweeps[2] = 0; prev_sweeps[3] = 0; }
  // slow_k (red[24..29]): [0] per timestep, log2 sqrt(3 |d_1|^2 / tol); [1], [3]: log2 of the two mode cutoffs of the slow-mode
  // landing guard, [2], [4]: their growth bounds (NS2DArgs::slow_*; +inf = none: the guard stays BCN_CONV_GUARD); [5]: -log2 of the
  // decay of err per sweep at the last planned pair (the next solve's opening).  In LDS, not in registers: read at the ~8
  // evaluated pairs of a solve only.
  real* const slow_k = red + 24;
  if (tid == 0) {
    slow_k[0] = 0; slow_k[5] = 0;
    slow_k[1] = (real)A.slow_l2lc[0]; slow_k[2] = (real)A.slow_cl[0];
    slow_k[3] = (real)A.slow_l2lc[1]; slow_k[4] = (real)A.slow_cl[1];
  }

  // A timestep is software-pipelined against its successor: the ordered part of the scalar transport (rayleigh.py:468-487)
  // is a chain of nx+ny-1 dependent steps that ONE wave walks (transport_chain*), and nothing in the next timestep's
  // predictor depends on the new T except the buoyancy term of v*, which is linear in T[i,j] (rayleigh.py:370-407).  So
  // while wave 0 walks the chain of timestep n, the other waves apply the velocity boundary conditions of timestep n+1
  // (in the phase before) and compute its whole predictor without buoyancy: u*, and X = diff - conv - dp/dy of v* (kept
  // with the old v: v* = v + dt (X + T) is finished behind the chain's barrier -- the same operations in the same order
  // as the unsplit form).  Where LDS has room (OFFLOAD), the helper waves also compute strip 0 (from p of strip 0 that
  // wave 0 left in LDS) and wave 0 only loads its u*, X; otherwise wave 0 computes its own strip behind the chain.
  constexpr bool OFFLOAD = G::template offload<real>();
  real* const P0 = exch + G::base_elems();   // [R][64] p of strip 0; then u* [R][64] and X [R][64] of strip 0
  real* const US0 = P0 + R0 * 64;
  real* const VX0 = US0 + R0 * 64;
  // one cell of the predictor (rayleigh.py:370-407) without the buoyancy term: u* (complete) and X of v*
  auto pred = [&](const int i, real uc, real uE_, real uW_, real uN_, real uS_, real uSE_, real vc, real vE_, real vW_, real vN_,
                  real vS_, real vNW_, real pc, real pW, real pS, real& us_out, real& vx_out) {
    {
      real uE = real(0.5) * (uE_ + uc), uW = real(0.5) * (uc + uW_);
      real uN2 = real(0.5) * (uN_ + uc), uS2 = real(0.5) * (uc + uS_);
      real vN2 = real(0.5) * (vN_ + vNW_), vS2 = real(0.5) * (vc + vW_);
      real conv = (uE * uE - uW * uW) * rdx + (uN2 * vN2 - uS2 * vS2) * rdy;
      real diff = ((uE_ - 2 * uc + uW_) * rdx2 + (uN_ - 2 * uc + uS_) * rdy2) * A.kmom;
      real pres = (pc - pW) * rdx;
      us_out = (i >= 2 && (!DEADC || i <= NX) && active) ? uc + dt * (diff - conv - pres) : real(0);
    }
    {
      real vE = real(0.5) * (vE_ + vc), vW = real(0.5) * (vc + vW_);
      real uE = real(0.5) * (uE_ + uSE_), uW = real(0.5) * (uc + uS_);
      real vN2 = real(0.5) * (vN_ + vc), vS2 = real(0.5) * (vc + vS_);
      real conv = (uE * vE - uW * vW) * rdx + (vN2 * vN2 - vS2 * vS2) * rdy;
      real diff = ((vE_ - 2 * vc + vW_) * rdx2 + (vN_ - 2 * vc + vS_) * rdy2) * A.kmom;
      real pres = (pc - pS) * rdy;
      vx_out = diff - conv - pres;
    }
  };
  // velocity boundary conditions (rayleigh.py:180-202, the u, v part): they read interior values the corrector has
  // finished and write ghost cells / wall faces that neither the corrector nor the transport step reads
  auto bc_uv = [&]() {
    for (int jj = 1 + tid; jj <= NY; jj += NT) {
      Ul[1 * SY + jj] = 0;
      Ul[(NX + 1) * SY + jj] = 0;
      if (jj >= 2) {
        Vl[0 * SY + jj] = -Vl[1 * SY + jj];
        Vl[(NX + 1) * SY + jj] = -Vl[NX * SY + jj];
      }
    }
    for (int ii = 1 + tid; ii <= NX + 1; ii += NT) {
      const bool wall = (ii == 1) || (ii == NX + 1);
      const real utop = wall ? real(0) : Ul[ii * SY + NY];
      const real ubot = wall ? real(0) : Ul[ii * SY + 1];
      Ul[ii * SY + NY + 1] = -utop;
      Ul[ii * SY + 0] = -ubot;
      if (ii <= NX) {
        Vl[ii * SY + NY + 1] = 0;
        Vl[ii * SY + 1] = 0;
      }
    }
  };
  // what the next predictor needs from the other strips / from wave 0: the strip's last p column, p of strip 0
  auto publish_p = [&]() {
    exl(xb, w, 1) = p[R - 1];
    if (OFFLOAD && w == 0) {
#pragma unroll
      for (int k = 0; k < R; k++) P0[k * 64 + lane] = p[k];
    }
  };

  // prologue of this unit: boundary conditions and p exchange of its first timestep
  bc_uv();
  publish_p();
  __syncthreads();

  real us[R], vs[R];
  for (int it = it_begin;; it++) {
    asm volatile("" : "+v"(j));   // keep hipcc from hoisting (and spilling) a whole timestep's LDS addresses out of the loop
    const bool have_chain = it > it_begin;                     // the ordered transport part of timestep it - 1 is pending
    const bool have_pred = it < it_end && status == 0;         // timestep it runs in this unit
    // ======== ordered part of the transport step (wave 0)  ||  predictor without buoyancy of the next timestep ========
    real vx[R];   // X of v* of this thread's cells
    // own strip: every wave in the unit's first timestep; behind that the helper waves (wave 0: below)
    const bool pred_own = have_pred && (w != 0 || !have_chain);
    if (w == 0 && have_chain) {
#ifndef BCN_CHAIN_PRIO
#define BCN_CHAIN_PRIO 3
#endif
      // the chain is the critical path of this region: its wave outranks the predictor of the wave it shares a SIMD with
      __builtin_amdgcn_s_setprio(BCN_CHAIN_PRIO);
      if constexpr (std::is_same<real, float>::value && GF == 0)
        transport_chain_f32<NX, NY, R0>(Tl, Ul, red + 16, dt * A.ksc * rdx2, real(0.5) * dt * rdx, dt * A.ksc * rdy2,
                                        real(0.5) * dt * rdy);
      else
        transport_chain<real, NX, NY, R0, GF>(Tl, Ul, Vl, red + 16, dt * A.ksc * rdx2, real(0.5) * dt * rdx,
                                              dt * A.ksc * rdy2, real(0.5) * dt * rdy);
      __builtin_amdgcn_s_setprio(0);
      BCN_PH(6)
    }
#define BCN_OWN_PRED                                                                                            \
    {                                                                                                           \
      const real pWh = (w > 0) ? exl(xb, w - 1, 1) : real(0);                                              \
      /* float64 (GF) and wide strips: in chunks of CH columns, so that the neighbour arrays fit the register file */ \
      constexpr bool WIDE = (R > 16) || (R > 12 && NW > 8);   /* more live values than the wave's register budget */ \
      constexpr int CH = GF ? 4 : (!WIDE ? R : (R % 4 == 0 ? 4 : 5));   /* the last chunk may be shorter (kk < R below) */ \
      _Pragma("unroll") for (int c0 = 0; c0 < R; c0 += CH) {                                                    \
        if (CH != R) __builtin_amdgcn_sched_barrier(0);   /* finish one chunk before loading the next */        \
        real ur[CH + 2], uS[CH + 1], uN[CH], vr[CH + 2], vN[CH + 1], vS[CH];                                    \
        const int ic = i0 + c0;                                                                                 \
        _Pragma("unroll") for (int k = 0; k < CH + 2; k++) { ur[k] = Ul[(ic - 1 + k) * SY + j]; vr[k] = Vl[(ic - 1 + k) * SY + j]; } \
        _Pragma("unroll") for (int k = 0; k < CH + 1; k++) { uS[k] = Ul[(ic + k) * SY + j - 1]; vN[k] = Vl[(ic - 1 + k) * SY + j + 1]; } \
        _Pragma("unroll") for (int k = 0; k < CH; k++) { uN[k] = Ul[(ic + k) * SY + j + 1]; vS[k] = Vl[(ic + k) * SY + j - 1]; } \
        _Pragma("unroll") for (int k = 0; k < CH; k++) {                                                        \
          const int kk = c0 + k;                                                                                \
          if (kk >= R) continue;   /* remainder chunk (resolved at compile time) */                             \
          const real pc = p[kk < R ? kk : 0];                                                                   \
          const real pW = (kk > 0) ? p[kk > 0 && kk <= R ? kk - 1 : 0] : pWh;                                   \
          const real pS = from_below(pc, pc);                                                                   \
          pred(ic + k, ur[k + 1], ur[k + 2], ur[k], uN[k], uS[k], uS[k + 1], vr[k + 1], vr[k + 2], vr[k], vN[k + 1], vS[k], \
               vN[k], pc, pW, pS, us[kk < R ? kk : 0], vx[kk < R ? kk : 0]);                                    \
        }                                                                                                       \
      }                                                                                                         \
    }
    if (pred_own) BCN_OWN_PRED
    if constexpr (OFFLOAD) {
      // strip 0 for wave 0, in chunks of four columns dealt to the helper waves (in an 8-wave workgroup wave 4 shares its
      // SIMD with wave 0: it comes last)
      if (have_pred && have_chain && w != 0) {
        // (ONE copy of the chunk code, the chunk a run-time index: the kernel has to stay inside the instruction cache)
        constexpr int CH0 = 4, NCH0 = (R0 + CH0 - 1) / CH0;
        static_assert(NCH0 <= NW - 1, "one chunk of strip 0 per helper wave");
        const int c = (NW == 8) ? (w < 4 ? w - 1 : (w == 4 ? 6 : w - 2)) : w - 1;
        if (c < NCH0) {
          const int kc = c * CH0;                   // first column of the chunk inside strip 0
          real ur[CH0 + 2], uS[CH0 + 1], uN[CH0], vr[CH0 + 2], vN[CH0 + 1], vS[CH0], pp[CH0 + 1];
          const int ic = 1 + kc;
#pragma unroll
          for (int k = 0; k < CH0 + 2; k++) { ur[k] = Ul[(ic - 1 + k) * SY + j]; vr[k] = Vl[(ic - 1 + k) * SY + j]; }
#pragma unroll
          for (int k = 0; k < CH0 + 1; k++) { uS[k] = Ul[(ic + k) * SY + j - 1]; vN[k] = Vl[(ic - 1 + k) * SY + j + 1]; }
#pragma unroll
          for (int k = 0; k < CH0; k++) { uN[k] = Ul[(ic + k) * SY + j + 1]; vS[k] = Vl[(ic + k) * SY + j - 1]; }
#pragma unroll
          for (int k = 0; k <= CH0; k++) {
            const int kk = kc + k - 1;               // p column of strip 0 (-1: the wall, no pressure gradient there)
            const int kq = kk < 0 ? 0 : (kk < R0 ? kk : R0 - 1);
            const real pv = P0[kq * 64 + lane];
            pp[k] = (kk >= 0 && kk < R0) ? pv : real(0);
          }
#pragma unroll
          for (int k = 0; k < CH0; k++) {
            const int kk = kc + k;                   // (columns past the strip, in its last chunk, are computed and dropped)
            const real pc = pp[k + 1];
            const real pS = from_below(pc, pc);
            real uo, xo;
            pred(ic + k, ur[k + 1], ur[k + 2], ur[k], uN[k], uS[k], uS[k + 1], vr[k + 1], vr[k + 2], vr[k], vN[k + 1], vS[k], vN[k],
                 pc, pp[k], pS, uo, xo);
            if (kk < R0) {
              US0[kk * 64 + lane] = uo;
              VX0[kk * 64 + lane] = xo;
            }
          }
        }
      }
    }
    if (w != 0) { BCN_PH(6) }
    if (have_pred) xb ^= 1;                          // (the p exchange buffer has been read)
    if (have_pred && w != 0) exl(xb, w, 0) = us[0];   // for the rhs of the strip to the west (nobody needs wave 0's)
    __syncthreads();
    BCN_PH(5)
    if (!have_pred) break;
    asm volatile("" : "+v"(j));   // keep hipcc from hoisting (and spilling) a whole timestep's LDS addresses out of the loop
    if (w == 0 && have_chain) {
      if constexpr (OFFLOAD) {
#pragma unroll
        for (int k = 0; k < R; k++) { us[k] = US0[k * 64 + lane]; vx[k] = VX0[k * 64 + lane]; }
      } else {
        BCN_OWN_PRED
      }
    }
#undef BCN_OWN_PRED
    // ---- boundary conditions of T (rayleigh.py:180-202, the T part) on the transported field --------------
    // TREG (T in the global scratch, GF == 2): every thread loads its own row of T once (one coalesced load per column, all
    // issued before the first use) and the ghost cells are WRITTEN from those registers by the lanes / waves at the walls --
    // the same expressions on the same values as the strided read-modify-write loops below, without their two rounds of
    // global-load latency in front of the buoyancy term
    constexpr bool TREG = GF == 2 && KIND == 0;
    real Tr[TREG ? R : 1];
    if constexpr (TREG) {
#pragma unroll
      for (int k = 0; k < R; k++) Tr[k] = Tl[(i0 + k) * SY + j];
      if (active) {   // (wave-uniform branches)
        if (w == 0) Tl[0 * SY + j] = Tr[0];
        if (w == NW - 1) Tl[(NX + 1) * SY + j] = Tr[(DEADC ? RL : R) - 1];
      }
      // top and bottom ghost rows: the wall rows' values (lanes NY - 1 and 0) are broadcast, lane k < R stores column k of the
      // strip -- one store instruction per ghost row, no lane-divergent block
      {
        real gt = 0, gb = 0;
#pragma unroll
        for (int k = 0; k < R; k++) {
          const real tt = read_lane(Tr[k], NY - 1), tb = read_lane(Tr[k], 0);
          gt = (lane == k) ? 2 * A.Tc - tt : gt;
          gb = (lane == k) ? tb : gb;
        }
        const int ii = i0 + (lane < R ? lane : 0);
        const int sg = (ii - 1) / A.nx_sgts;
        const real sa = sact[sg < A.n_sgts ? sg : 0];
        if (lane < (lastw ? RL : R)) {
          Tl[ii * SY + NY + 1] = gt;
          if (sg < A.n_sgts) Tl[ii * SY + 0] = 2 * (A.Th + sa) - gb;
        }
      }
    } else {
    for (int jj = 1 + tid; jj <= NY; jj += NT) {
      Tl[0 * SY + jj] = Tl[1 * SY + jj];
      Tl[(NX + 1) * SY + jj] = Tl[NX * SY + jj];
    }
    for (int ii = 1 + tid; ii <= NX; ii += NT) {
      Tl[ii * SY + NY + 1] = 2 * A.Tc - Tl[ii * SY + NY];
      const int k = (ii - 1) / A.nx_sgts;
      if (k < A.n_sgts) Tl[ii * SY + 0] = 2 * (A.Th + sact[k]) - Tl[ii * SY + 1];
    }
    }
    // ---- buoyancy: v* = v + dt (X + T) (rayleigh.py:405) ----------------------------------------------------
#pragma unroll
    for (int k = 0; k < R; k++) {
      const real buoy = (KIND == 0) ? (TREG ? Tr[TREG ? k : 0] : Tl[(i0 + k) * SY + j]) : real(0);
      vs[k] = (j >= 2 && active && LIVE(k)) ? Vl[(i0 + k) * SY + j] + dt * (vx[k] + buoy) : real(0);   // (the old v: re-read, not kept)
    }
    BCN_PH(0)

    // ---- Poisson rhs (rayleigh.py:424-426) ---------------------------------------------------
    real nb[R];   // minus the scaled rhs
    {
      const real usE = (w < NW - 1) ? exl(xb, w + 1, 0) : real(0);   // u*[nx+1,.] = 0
      xb ^= 1;
#pragma unroll
      for (int k = 0; k < R; k++) {
        const real ue = (k < R - 1) ? us[k < R - 1 ? k + 1 : 0] : usE;
        const real vn = from_above(real(0), vs[k]);                        // v*[., ny+1] = 0 (lanes >= NY hold 0)
        nb[k] = (active && LIVE(k)) ? -A.cb * ((ue - us[k]) * rdx + (vn - vs[k]) * rdy) : real(0);
      }
    }
    // rhs of the neighbouring strips' edge columns: the double sweeps below recompute those columns (depth-2 halos)
    const int wm = (w > 0) ? w - 1 : 0, wp = (w < NW - 1) ? w + 1 : NW - 1;
    real nbW = 0, nbE = 0;
    if constexpr (XC == 4) {
      exl(xb, w, 0) = nb[0];
      exl(xb, w, 3) = nb[R - 1];
      // |d_1|^2 = |phi_1 - 0|^2 = sum nb^2 (plain, interior): what the slow-mode landing guard below scales with.  The partials
      // travel in the exchange buffer's column slot 1, which the rhs exchange leaves unused, behind the same barrier.
      real a1p = 0;
#pragma unroll
      for (int k = 0; k < R; k++) a1p += nb[k] * nb[k];
      const real a1w = wave_sum_lane63<real>(a1p);
      if (lane == 63) exch[((xb * NW + w) * XC + 1) * 64] = a1w;
      __syncthreads();
      nbW = exl(xb, wm, 3);
      nbE = exl(xb, wp, 0);
      const real a1q = (lane < NW) ? exch[((xb * NW + (lane < NW ? lane : 0)) * XC + 1) * 64] : real(0);
      const real a1 = read_lane(row16_sum<real>(a1q), 15);
      // slow_k[0] = log2(sqrt(3 |d_1|^2 / tol)): read again where a landing guard is computed, at least one barrier from here
      if (tid == 0) slow_k[0] = (real)(0.5f * __log2f(3.f * (float)a1 / (float)A.tol));
      xb ^= 1;
    } else if (A.conv_plan == 3) {
      // (single-sweep exchange: no rhs exchange to ride on -- one barrier per solve; the partials go to the errp half that the
      // first evaluated sweep does not write)
      real a1p = 0;
#pragma unroll
      for (int k = 0; k < R; k++) a1p += nb[k] * nb[k];
      const real a1w = wave_sum_lane63<real>(a1p);
      if (lane == 63) errp[(xb ^ 1) * 64 + w] = a1w;
      __syncthreads();
      const real a1 = read_lane(row16_sum<real>(errp[(xb ^ 1) * 64 + (lane & 15)]), 15);
      if (tid == 0) slow_k[0] = (real)(0.5f * __log2f(3.f * (float)a1 / (float)A.tol));
    }

    BCN_PH(1)
    const unsigned long long cyc_j0 = __builtin_amdgcn_s_memtime();
    // ---- Jacobi sweeps (rayleigh.py:419-454): one barrier per sweep --------------------------
    // phi ping-pongs between two register arrays (no copies).  Behind the barrier of a sweep the LDS reads of the
    // strip-edge halos are issued first and the R-2 interior cells of the NEXT sweep are computed while they are in
    // flight; the two edge cells follow and go to the exchange buffer at once.
    //
    // Which sweeps evaluate the residual.  The reference evaluates err_k = sum((phi_k - phi_{k-1})^2) over the whole
    // array after EVERY sweep and stops at the first k with err_k <= tol (rayleigh.py:448-454).  A sweep is
    // d_{k+1} = J d_k for the increments d_k, with J = (Adj + G) / 4 symmetric (Adj: neighbour matrix of the interior
    // cells, G: diagonal count of mirrored ghost sides; a Dirichlet-zero ghost drops out), so the UNWEIGHTED interior
    // norm a_k = |d_k|^2 = sum_i lambda_i^{2k} c_i^2 is log-convex in k: its per-sweep decay factor a_{k+1}/a_k never
    // decreases.  Hence, from two evaluated sweeps kp < k, rho = (a_k / a_kp)^(1/(k-kp)) is a lower bound of every
    // later factor, and err_{k+i} >= a_{k+i} >= a_k rho^i (the reference's norm counts the ghost copies on top of the
    // interior: err = d'(I+G)d >= a).  While a_k rho^i > 1.02 tol the test cannot pass, and those sweeps run WITHOUT
    // the residual (5 instead of 7 instructions per cell, no wave reduction, no partials through LDS).
    // A.conv_plan: 0 = evaluate every sweep (the reference, literally); 1 = skip only what the bound above proves
    // (exact stop sweep: the float64 default); 2 = extrapolate the reference norm itself instead, whose decay factor is
    // observed -- not proven: I+G does not commute with J -- never to decrease either (10 000 sweeps of oracle traces,
    // scripts/plan_sim.py), ending the skip 1 + 1/16 of its length early (the float32 default: the evaluations drop
    // from ~28 % to ~9 % of the sweeps).  A.verify_conv evaluates every sweep anyway and raises
    // BCN_ST_PLAN if a sweep the plan would have skipped passes the test (tests/test_gpu_parity.py).
    auto cell = [&](real c, real e, real wv, real nbk) -> real {
      const real q = cBy * c + nbk;                       // Neumann ghosts in y copy the cell itself
      real ph;
      if constexpr (EQ && std::is_same<real, float>::value) {
        ph = jacobi_cell_eq(c, e, wv, nbk, cx, cBy);       // the whole cell as one statement (bcn_dpp.h)
      } else if (EQ) {
        real sum = e + wv;
        sum = add_above_below(sum, c);                    // + north (lane+1) + south (lane-1); 0 outside the wave
        ph = cx * sum + q;
      } else {
        const real ns = dpp<0x130, 0xf, 0xf, true>(real(0), c) + dpp<0x138, 0xf, 0xf, true>(real(0), c);
        ph = cx * (e + wv) + (cy * ns + q);
      }
      if (NY < 64) ph *= actf;                            // lanes past the top row stay 0
      return ph;
    };
    real hW = 0, hE = 0;            // halos of the array the last sweep read
    // columns -1, -2 / R, R+1 of the array the last sweep wrote (LDS reads issued behind its barrier)
    real hW1r, hW2r, hE1r, hE2r;
    int itp;
    real phA[R], phB[R];
    bool finalB;
    // tolL: what a LANDING evaluation -- the first one behind skipped sweeps -- must exceed for the skip to be verified.  Plan 3:
    // BCN_CONV_GUARD * tol, which PROVES that no skipped sweep passed (bcn_common.h), or less where the slow-mode constants of the
    // grid allow (tolL below, beacon_amd/stoprule.py); the plan aims its landings above it.
    // Plan 2: tol itself (only a landing that passes is noticed: the unguarded rule of round 2).
    const real tolL0 = (A.conv_plan == 3) ? A.tol * real(BCN_CONV_GUARD) : A.tol;
    const float l2tol_u = __log2f((float)A.tol * 1.02f), l2tol_w = __log2f((float)tolL0 * 1.003f);
    constexpr int JMAX = 256;
#ifdef BCN_DBG_NCHK
    int nchk = 0;
#define BCN_NCHK_INC nchk++;
#else
#define BCN_NCHK_INC
#endif
#ifdef BCN_STAMP
    const unsigned long long st0 = __builtin_amdgcn_s_memtime();
#endif
    for (;;) {   // the solve of this timestep: once -- again when a speculative jump went too far or, under conv_plan 3, when
                 // the extrapolating plan did not foresee the stop (below); u*, v* and the rhs are untouched by the sweeps
    hW1r = 0; hW2r = 0; hE1r = 0; hE2r = 0;
    itp = 0;
#pragma unroll
    for (int k = 0; k < R; k++) phA[k] = 0;
    finalB = false;
    real tolL = tolL0;              // the landing guard in force: BCN_CONV_GUARD * tol, or the slow-mode guard of the last evaluation
    int k_prev = -1;                // index of the planned evaluation before the last one, log2 of its two norms
    float l2u_prev = 0, l2w_prev = 0;
    int skip_left = 0;              // verify_conv: sweeps the plan would still skip; -2 / -1: speculative jump pending / failed
    // Speculative jump.  The norm never increases from one sweep to the next (|lambda_i| <= 1; the reference norm adds
    // the ghost copies), so behind the evaluations of sweeps 1 and 2 the next one may sit at sweep m: if it does not pass,
    // no sweep in between did.  m = spec_start/8 of the previous timestep's count (consecutive timesteps of the bench
    // workload differ by < 30 % in 611 000 solves).  If the evaluation at m DOES pass the test (or, under the proven plan,
    // its unweighted norm is not above the threshold), the solve is repeated without the jump -- the result never
    // depends on the guess.  The bookkeeping lives in `skip_left` (unused without verify_conv) and in LDS: every scalar
    // register more in this loop costs hipcc dozens of SGPR spills (v_readlane) around the sweeps.
    // the evaluation plan of THIS solve: conv_plan 3 is plan 2 whose unverified stops are repeated under plan 1
    const int plan = (A.conv_plan == 3) ? (__builtin_amdgcn_readfirstlane((int)prev_sweeps[1]) != 0 ? 1 : 2) : A.conv_plan;
    // all cells of one sweep; the two strip-edge cells come last (their halos were requested behind the previous
    // barrier) and go to the exchange buffer at once, in front of whatever else the sweep still has to do
#define BCN_CELLS(SRC, DST)                                                                  \
      _Pragma("unroll") for (int k = 1; k < R - 1; k++) DST[k] = cell(SRC[k], SRC[k + 1], SRC[k - 1], nb[k]); \
      /* keep the halo-dependent part behind the interior cells: hipcc otherwise sometimes hoists the edge cells \
         (and their s_waitcnt on the LDS reads) in front of them: +170 cycles per sweep */   \
      __builtin_amdgcn_sched_barrier(0);                                                     \
      hW = (w > 0) ? hW1r : SRC[0];                                                          \
      hE = (w < NW - 1) ? hE1r : SRC[R - 1];                                                 \
      const real p0 = cell(SRC[0], SRC[1], hW, nb[0]);                                       \
      const real pl = cell(SRC[R - 1], hE, SRC[R - 2], nb[R - 1]);                           \
      DST[0] = p0;                                                                           \
      DST[R - 1] = pl;                                                                       \
      /* one body, last strip: the Neumann ghost east of the live edge, as a cell of the register row */ \
      if constexpr (DEADC) DST[DEADC ? RL : 0] = lastw ? DST[DEADC ? RL - 1 : 0] : DST[DEADC ? RL : 0]; \
      BCN_PUBLISH(DST)
#define BCN_PUBLISH(DST)                                                                     \
      exl(xb, w, 0) = DST[0];                                                                \
      exl(xb, w, XL) = DST[R - 1];                                                           \
      if constexpr (XC == 4) { exl(xb, w, 1) = DST[1]; exl(xb, w, 2) = DST[R - 2]; }
#define BCN_HALO_READS                                                                       \
      hW1r = exl(xb, wm, XL);                                                                \
      hE1r = exl(xb, wp, 0);                                                                 \
      if constexpr (XC == 4) { hW2r = exl(xb, wm, 2); hE2r = exl(xb, wp, 1); }               \
      xb ^= 1;
#define BCN_SWEEP_END                                                                        \
      __syncthreads();                                                                       \
      itp++;                                                                                 \
      BCN_HALO_READS
    // a sweep that does not evaluate the residual
#define BCN_FAST(SRC, DST) { BCN_CELLS(SRC, DST) BCN_SWEEP_END }
    // TWO such sweeps (X -> Y -> X) behind ONE barrier: the first also advances the neighbours' edge columns -1 and R from
    // the depth-2 halos (a wall's ghost column mirrors the strip's own new edge), the second then needs nothing from other
    // waves.  Two redundant cells per 2 R, half the barriers and exchanges; every value is computed exactly as in two
    // single sweeps.
#define BCN_FAST2X(X, Y)                                                                     \
    {                                                                                        \
      _Pragma("unroll") for (int k = 1; k < R - 1; k++) Y[k] = cell(X[k], X[k + 1], X[k - 1], nb[k]); \
      __builtin_amdgcn_sched_barrier(0);                                                     \
      const real xw1 = (w > 0) ? hW1r : X[0], xe1 = (w < NW - 1) ? hE1r : X[R - 1];          \
      Y[0] = cell(X[0], X[1], xw1, nb[0]);                                                   \
      Y[R - 1] = cell(X[R - 1], xe1, X[R - 2], nb[R - 1]);                                   \
      real yw = cell(hW1r, X[0], hW2r, nbW), ye = cell(hE1r, hE2r, X[R - 1], nbE);           \
      yw = (w > 0) ? yw : Y[0];                                                              \
      ye = (w < NW - 1) ? ye : Y[R - 1];                                                     \
      __builtin_amdgcn_sched_barrier(0);                                                     \
      /* the second sweep needs no halo: its four exchanged columns go FIRST, so that their LDS stores complete behind \
         the interior cells instead of in front of the barrier */                             \
      X[0] = cell(Y[0], Y[1], yw, nb[0]);                                                    \
      X[1] = cell(Y[1], Y[2], Y[0], nb[1]);                                                  \
      X[R - 2] = cell(Y[R - 2], Y[R - 1], Y[R - 3], nb[R - 2]);                              \
      X[R - 1] = cell(Y[R - 1], ye, Y[R - 2], nb[R - 1]);                                    \
      hW = yw; hE = ye;                                                                      \
      BCN_PUBLISH(X)                                                                         \
      __builtin_amdgcn_sched_barrier(0);                                                     \
      _Pragma("unroll") for (int k = 2; k < R - 2; k++) X[k] = cell(Y[k], Y[k + 1], Y[k - 1], nb[k]); \
      __syncthreads();                                                                       \
      itp += 2;                                                                              \
      BCN_HALO_READS                                                                         \
    }
    // a sweep that does (the same arithmetic, in the same order, as when it was fused into the cells), evaluated right
    // behind its barrier; sets `n`: the number of following sweeps that cannot pass the test
#define BCN_CHECK(SRC, DST, DST_IS_B)                                                        \
    {                                                                                        \
      BCN_CELLS(SRC, DST)                                                                    \
      BCN_NCHK_INC                                                                           \
      real acc = 0;                                                                          \
      _Pragma("unroll") for (int k = 1; k < R - 1; k++) {                                    \
        real d = DST[k] - SRC[k];                                                            \
        if (DEADC && k >= RL) d = lastw ? real(0) : d;   /* dead columns of the last strip */ \
        acc += d * d;                                                                        \
      }                                                                                      \
      real pI = wl * acc;                                                                    \
      if constexpr (DEADC) { const real de = DST[DEADC ? RL - 1 : 0] - SRC[DEADC ? RL - 1 : 0]; pI += fEl * (de * de); } \
      const real d0 = p0 - SRC[0], dl = pl - SRC[R - 1];                                     \
      const real part = pI + cW * (d0 * d0) + cE * (dl * dl);                                \
      const real tot63 = wave_sum_lane63<real>(part);                                        \
      if (lane == 63) errp[xb * 64 + w] = tot63;                                             \
      if (plan == 1) {   /* the proven plan needs the unweighted interior norm too (lanes past the top row hold zeros) */ \
        const real totu63 = wave_sum_lane63<real>(acc + d0 * d0 + (lastw ? real(0) : dl * dl)); \
        if (lane == 63) errp[xb * 64 + 16 + w] = totu63;                                     \
      }                                                                                      \
      __syncthreads();                                                                       \
      itp++;                                                                                 \
      /* one read: lane q holds wave q's partial of the reference norm, lane 16 + q that of the unweighted norm; \
         summed in a fixed order, uniformly in every lane */                                 \
      const real epart = errp[xb * 64 + (lane & 31)];                                        \
      BCN_HALO_READS                                                                         \
      const real esum = row16_sum<real>(epart);                                              \
      const real err = read_lane(esum, 15);                                                  \
      /* behind the speculative jump the proven plan also needs the unweighted norm above its threshold */ \
      const bool amb = SPEC && skip_left == -2 && plan == 1 && !(read_lane(esum, 31) > A.tol * real(1.02)); \
      /* the reference tests the sweep count FIRST (rayleigh.py:451-454): sweep itmax + 1 overflows even if it passes */ \
      if (itp > A.itmax) { status |= BCN_ST_ITMAX; finalB = DST_IS_B; if (SPEC) skip_left = 0; break; } \
      /* a landing (the sweep before this one was not evaluated) that does not clear tolL leaves the skipped sweeps unverified */ \
      const bool unv = plan >= 2 && !A.verify_conv && itp >= 2 && k_prev != itp - 2 && !(err > tolL); \
      if (!(err > A.tol) || amb || unv) {                                                    \
        if (skip_left > 0) status |= BCN_ST_PLAN;                                            \
        if (SPEC) skip_left = skip_left == -2 ? -1 : 0;                                      \
        finalB = DST_IS_B; break;                                                            \
      }                                                                                      \
      if (SPEC) skip_left = skip_left < 0 ? 0 : skip_left;                                   \
      n = 0;                                                                                 \
      if (skip_left > 0) {                                                                   \
        skip_left--;                                                                         \
      } else if (plan > 0) {   /* plan the next evaluation (see above) */             \
        const float l2u = (plan == 1) ? __log2f((float)read_lane(esum, 31)) : 0.f; \
        const float l2w = __log2f((float)err);                                               \
        int j = 0;                                                                           \
        if (k_prev >= 0) {                                                                   \
          const float rg = 1.f / (float)(itp - 1 - k_prev);                                  \
          if (plan == 1) {                                                            \
            const float room_u = l2u - l2tol_u, rho_u = (l2u - l2u_prev) * rg;               \
            if (room_u > 0.f) j = (rho_u < 0.f) ? (int)fminf(room_u / -rho_u, (float)JMAX) : JMAX; \
          } else {                                                                           \
            float l2tl = l2tol_w;                                                            \
            if (A.conv_plan == 3) {   /* the slow-mode guard of the landing this skip ends in; this sweep, itp, is the last evaluated one */ \
              const float e0 = (float)slow_k[0], fi = (float)itp;                            \
              const float t0 = 1.f + 2.f * exp2f(e0 + fi * (float)slow_k[1]), t1 = 1.f + 2.f * exp2f(e0 + fi * (float)slow_k[3]); \
              const float g = fminf(fminf((float)slow_k[2] * t0 * t0, (float)slow_k[4] * t1 * t1) * 1.001f, (float)BCN_CONV_GUARD); \
              tolL = A.tol * (real)g;                                                        \
              l2tl = __log2f((float)tolL * 1.003f);                                          \
            }                                                                                \
            const float room_w = l2w - l2tl, rho_w = (l2w - l2w_prev) * rg;                  \
            int jw = 0;                                                                      \
            if (room_w > 0.f) jw = (rho_w < 0.f) ? (int)fminf(room_w / -rho_w, (float)JMAX) : JMAX; \
            j = jw - 1 - (jw >> 4) + A.plan_overshoot;                                                          \
            j = j > 0 ? j : 0;                                                               \
          }                                                                                  \
        }                                                                                    \
        j = __builtin_amdgcn_readfirstlane(j);                                               \
        l2u_prev = l2u; l2w_prev = l2w; k_prev = itp - 1;                                    \
        if (A.verify_conv) skip_left = j; else n = j;                                        \
      }                                                                                      \
    }
    // TWO evaluated sweeps (X -> Y -> X) behind ONE barrier and ONE decision: the double sweep above with the residual of
    // both of its sweeps -- the arithmetic of BCN_CHECK in the same order, so the norms are bit for bit those of two single
    // evaluated sweeps -- whose partials travel in one LDS word each (lane q / 16 + q: wave q's partial of the reference
    // norm of the first / second sweep; 32 + q / 48 + q: of the unweighted norm) and come back with one read and one
    // row sum.  An evaluated single sweep costs 2.6 fast ones (1 816 against 700 cycles: the chain residual -> wave
    // reduction -> LDS -> barrier -> LDS -> row sum -> decision -> plan is serial); the pair pays that chain once.  If the
    // first sweep passes, Y is the result (X holds one sweep more, which nothing reads).  The decay factor of the plan is
    // that of the pair itself: the most recent -- by log-convexity the largest -- lower bound of every later factor.
#define BCN_CHECK2X(X, Y)                                                                    \
    {                                                                                        \
      const int itp0 = itp;                                                                  \
      _Pragma("unroll") for (int k = 1; k < R - 1; k++) Y[k] = cell(X[k], X[k + 1], X[k - 1], nb[k]); \
      __builtin_amdgcn_sched_barrier(0);                                                     \
      const real xw1 = (w > 0) ? hW1r : X[0], xe1 = (w < NW - 1) ? hE1r : X[R - 1];          \
      Y[0] = cell(X[0], X[1], xw1, nb[0]);                                                   \
      Y[R - 1] = cell(X[R - 1], xe1, X[R - 2], nb[R - 1]);                                   \
      real yw = cell(hW1r, X[0], hW2r, nbW), ye = cell(hE1r, hE2r, X[R - 1], nbE);           \
      yw = (w > 0) ? yw : Y[0];                                                              \
      ye = (w < NW - 1) ? ye : Y[R - 1];                                                     \
      BCN_NCHK_INC BCN_NCHK_INC                                                              \
      real acc1 = 0;                                                                         \
      _Pragma("unroll") for (int k = 1; k < R - 1; k++) { const real d = Y[k] - X[k]; acc1 += d * d; } \
      const real d10 = Y[0] - X[0], d1l = Y[R - 1] - X[R - 1];                               \
      const real part1 = wl * acc1 + cW * (d10 * d10) + cE * (d1l * d1l);                    \
      const real upart1 = acc1 + d10 * d10 + d1l * d1l;                                      \
      __builtin_amdgcn_sched_barrier(0);                                                     \
      X[0] = cell(Y[0], Y[1], yw, nb[0]);                                                    \
      X[1] = cell(Y[1], Y[2], Y[0], nb[1]);                                                  \
      X[R - 2] = cell(Y[R - 2], Y[R - 1], Y[R - 3], nb[R - 2]);                              \
      X[R - 1] = cell(Y[R - 1], ye, Y[R - 2], nb[R - 1]);                                    \
      hW = yw; hE = ye;                                                                      \
      BCN_PUBLISH(X)                                                                         \
      __builtin_amdgcn_sched_barrier(0);                                                     \
      _Pragma("unroll") for (int k = 2; k < R - 2; k++) X[k] = cell(Y[k], Y[k + 1], Y[k - 1], nb[k]); \
      real acc2 = 0;                                                                         \
      _Pragma("unroll") for (int k = 1; k < R - 1; k++) { const real d = X[k] - Y[k]; acc2 += d * d; } \
      const real d20 = X[0] - Y[0], d2l = X[R - 1] - Y[R - 1];                               \
      const real part2 = wl * acc2 + cW * (d20 * d20) + cE * (d2l * d2l);                    \
      const real t1_63 = wave_sum_lane63<real>(part1), t2_63 = wave_sum_lane63<real>(part2); \
      if (lane == 63) { errp[xb * 64 + w] = t1_63; errp[xb * 64 + 16 + w] = t2_63; }         \
      if (plan == 1) {   /* the proven plan needs the unweighted interior norm too (lanes past the top row hold zeros) */ \
        const real u1_63 = wave_sum_lane63<real>(upart1);                                    \
        const real u2_63 = wave_sum_lane63<real>(acc2 + d20 * d20 + d2l * d2l);              \
        if (lane == 63) { errp[xb * 64 + 32 + w] = u1_63; errp[xb * 64 + 48 + w] = u2_63; }  \
      }                                                                                      \
      __syncthreads();                                                                       \
      itp += 2;                                                                              \
      const real epart = errp[xb * 64 + lane];                                               \
      BCN_HALO_READS                                                                         \
      const real esum = row16_sum<real>(epart);                                              \
      const real err1 = read_lane(esum, 15), err2 = read_lane(esum, 31);                     \
      /* the reference tests the sweep count FIRST (rayleigh.py:451-454): a sweep beyond itmax ends the solve as an overflow \
         whether it passes or not -- the pair's first sweep is number itp - 1, its second itp */ \
      const bool ov1 = itp - 1 > A.itmax, ov2 = itp > A.itmax;                               \
      const bool pass1 = ov1 || !(err1 > A.tol), pass2 = ov2 || !(err2 > A.tol);             \
      /* behind the speculative jump the proven plan also needs the unweighted norm above its threshold */ \
      const bool amb = SPEC && skip_left == -2 && plan == 1 && !(read_lane(esum, 47) > A.tol * real(1.02)); \
      /* the pair directly follows skipped sweeps (or the speculative opening) and its first sweep does not clear tolL: \
         the skipped sweeps are not verified -- plan 3 repeats the solve, plan 2 notices it only when that sweep passes */ \
      const bool unv = plan >= 2 && !A.verify_conv && itp0 > 0 && k_prev != itp0 && !(err1 > tolL); \
      if (pass1 || pass2 || amb || unv) {                                                    \
        const bool ovf = ov1 || (ov2 && (err1 > A.tol));                                     \
        if (ovf) status |= BCN_ST_ITMAX;                                                     \
        if (!ovf && skip_left > (pass1 ? 0 : 1)) status |= BCN_ST_PLAN;   /* verify_conv: a sweep the plan skips passes */ \
        if (SPEC) skip_left = (skip_left == -2 && !ovf) ? -1 : 0;                            \
        finalB = pass1;                                                                      \
        itp -= pass1 ? 1 : 0;                                                                \
        /* the first sweep was the last one: the west halo of ITS result is the neighbour's edge column as recomputed here */ \
        hW1r = pass1 ? yw : hW1r;                                                            \
        /* a stop the plan did not foresee: the passing sweep directly follows skipped ones */ \
        late_stop = unv && !ovf;                                                             \
        break;                                                                               \
      }                                                                                      \
      if (SPEC) skip_left = skip_left < 0 ? 0 : skip_left;                                   \
      n = 0;                                                                                 \
      if (skip_left > 0) {                                                                   \
        skip_left = skip_left > 2 ? skip_left - 2 : 0;                                       \
      } else if (plan > 0) {   /* plan the next evaluation (see above) */                    \
        int j = 0;                                                                           \
        if (plan == 1) {                                                                     \
          const float l2u1 = __log2f((float)read_lane(esum, 47)), l2u2 = __log2f((float)read_lane(esum, 63)); \
          const float room_u = l2u2 - l2tol_u, rho_u = l2u2 - l2u1;                          \
          if (room_u > 0.f) j = (rho_u < 0.f) ? (int)fminf(room_u / -rho_u, (float)JMAX) : JMAX; \
        } else {                                                                             \
          const float l2w1 = __log2f((float)err1), l2w2 = __log2f((float)err2);              \
          float l2tl = l2tol_w;                                                              \
          if (A.conv_plan == 3) {   /* the slow-mode guard of the landing this skip ends in; the last evaluated sweep is itp */ \
            const float e0 = (float)slow_k[0], fi = (float)itp;                              \
            const float t0 = 1.f + 2.f * exp2f(e0 + fi * (float)slow_k[1]), t1 = 1.f + 2.f * exp2f(e0 + fi * (float)slow_k[3]); \
            const float g = fminf(fminf((float)slow_k[2] * t0 * t0, (float)slow_k[4] * t1 * t1) * 1.001f, (float)BCN_CONV_GUARD); \
            tolL = A.tol * (real)g;                                                          \
            l2tl = __log2f((float)tolL * 1.003f);                                            \
          }                                                                                  \
          const float room_w = l2w2 - l2tl, rho_w = l2w2 - l2w1;                             \
          if (SPEC && tid == 0) slow_k[5] = (real)-rho_w;   /* the decay per sweep, for the next solve's opening */ \
          int jw = 0;                                                                        \
          if (room_w > 0.f) jw = (rho_w < 0.f) ? (int)fminf(room_w / -rho_w, (float)JMAX) : JMAX; \
          j = jw - 1 - (jw >> 4) + A.plan_overshoot;                                         \
          j = j > 0 ? j : 0;                                                                 \
        }                                                                                    \
        j = __builtin_amdgcn_readfirstlane(j) & ~1;                                          \
        k_prev = itp;                                                                        \
        if (A.verify_conv) skip_left = j; else n = j;                                        \
      }                                                                                      \
    }
    // Evaluations come in pairs wherever the exchange carries depth-2 halos (XC == 4).  Measured on the float32 128x64 kernel,
    // cycles per sweep on average: plan 0 1 816 -> 1 417, plan 1 1 133 -> 1 006; the default (plan 3 with the jump at the
    // start of the solve, whose landing is two consecutive evaluations anyway) 778 -> 740.
    constexpr bool PAIRS = XC == 4;
    bool late_stop = false;
    // Speculative jump (see above), taken at once: with a previous count to go by, the solve OPENS with double sweeps up to
    // spec_start/8 of it and evaluates the residual there for the first time -- sweeps 1 and 2, evaluated only to start the
    // plan, cost two of the solve's ~9 evaluations (an evaluated sweep costs 2.6 fast ones).  If that first evaluation
    // fails, no earlier sweep passed; if it passes, the solve is repeated without the guess.
    if constexpr (SPEC && XC == 4) {
      // (extrapolating plans only: the proven plan must find the UNWEIGHTED norm above the tolerance where it lands, which
      // at 7/8 of the previous count it almost never is -- measured: nearly every solve repeated)
      if (A.spec_start > 0 && plan > 1 && !A.verify_conv) {
        const int prev = __builtin_amdgcn_readfirstlane((int)prev_sweeps[0]);
        // spec_start 1..16: that many eighths of the previous count.  17 (the default): the count itself moves by a few per cent
        // from one timestep to the next (0.93 .. 1.05 in the bench workload) -- what the landing must stay clear of is the ZONE in
        // front of the stop in which err is already below the landing guard, log2(guard) / (decay per sweep) sweeps long (a dozen
        // at 1.035 and 0.3 % per sweep).  So: 15/16 of the previous count minus 1.25 zones, the decay taken from the previous
        // solve's last planned pair (slow_k[5]; unknown at the head of a chunk: 6/8 of the count).
        int n0 = (prev * (A.spec_start <= 16 ? A.spec_start : 6)) >> 3;
        if (A.spec_start == 17 && plan == 2 && A.conv_plan == 3) {
          const float rp = (float)slow_k[5];
          if (rp > 0.f) n0 = ((prev * 15) >> 4) - (int)fminf(ceilf(BCN_OPEN_ZONE_L2 / rp), 4096.f);
        }
        n0 = __builtin_amdgcn_readfirstlane(n0) & ~1;
        if (n0 > A.itmax) n0 = A.itmax & ~1;
        if (prev >= 16 && n0 > 0) {
          skip_left = -2;
          for (; n0 > 0; n0 -= 2) BCN_FAST2X(phA, phB)
        }
      }
    }
    if constexpr (PAIRS) {
      for (;;) {
        int n;
        BCN_CHECK2X(phA, phB)
        if (n > A.itmax - itp) n = (A.itmax - itp > 0 ? A.itmax - itp : 0) & ~1;   // the overflow test sits in the check sweeps
        for (; n > 0; n -= 2) BCN_FAST2X(phA, phB)
      }
    } else {
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
      if (SPEC && XC != 4 && itp == 2 && A.spec_start > 0 && plan > 0 && !A.verify_conv) {   // (single-sweep exchange: behind sweeps 1, 2)
        const int prev = __builtin_amdgcn_readfirstlane((int)prev_sweeps[0]);
        const int ns = ((prev * (A.spec_start <= 16 ? A.spec_start : 6)) >> 3) - 2;
        if (prev >= 16 && ns > n) { n = ns & ~1; skip_left = -2; }
      }
      if (n > A.itmax - itp) n = (A.itmax - itp > 0 ? A.itmax - itp : 0) & ~1;   // the overflow test sits in the check sweeps
      if constexpr (XC == 4) {
        for (; n > 0; n -= 2) BCN_FAST2X(phA, phB)
      } else {
        for (; n > 0; n -= 2) {
          BCN_FAST(phA, phB)
          BCN_FAST(phB, phA)
        }
      }
    }
    late_stop = itp >= 2 && k_prev != itp - 2;   // (single evaluated sweeps: k_prev is the index of the one before the last)
    }
#undef BCN_CHECK2X
#undef BCN_CHECK
#undef BCN_FAST2X
#undef BCN_HALO_READS
#undef BCN_PUBLISH
#undef BCN_FAST
#undef BCN_SWEEP_END
#undef BCN_CELLS
    {
      // Guard of the extrapolating plan (conv_plan 2, 3).  A stop at sweep s is the reference's stop sweep when sweep s - 1
      // was evaluated (and failed); when the passing evaluation directly follows SKIPPED sweeps the plan's own estimate was
      // wrong -- the decay accelerated -- and an earlier sweep may have passed as well: a "late stop", counted per replica
      // (bcn_get_counters).  conv_plan 3 repeats such a solve under the proven plan 1, as a jump that went too far
      // (skip_left == -1) repeats it without the guess.
      const bool toofar = SPEC && skip_left == -1;
      const bool late = plan >= 2 && !toofar && late_stop && !(status & BCN_ST_ITMAX);
      if (late && tid == 0) prev_sweeps[2] += 1;
      if (!(toofar || (late && A.conv_plan == 3))) {
        if (tid == 0) { prev_sweeps[0] = SPEC ? (real)itp : real(0); prev_sweeps[1] = 0; }
        break;
      }
      if (tid == 0) { prev_sweeps[0] = 0; prev_sweeps[1] = toofar ? real(0) : real(1); prev_sweeps[3] += 1; }
      __syncthreads();   // the words above are read at the head of the repeated solve
    }
    }
    if (finalB) {
#pragma unroll
      for (int k = 0; k < R; k++) phA[k] = phB[k];
    }
    hW = hW1r;   // west halo of the final phi (read behind the last barrier; unused by wave 0)
#ifdef BCN_STAMP   // diagnostic build only: cycles per sweep in the high half of the sweep count
    {
      const unsigned long long st1 = __builtin_amdgcn_s_memtime();
      const int cps = (int)((st1 - st0) / (unsigned long long)(itp > 0 ? itp : 1));
      if (A.sweeps && tid == 0) A.sweeps[(size_t)b * A.ndt_act + it] = itp | (cps << 16);
    }
#elif defined(BCN_DBG_NCHK)   // diagnostic build only: residual evaluations in the high half of the sweep count
    if (A.sweeps && tid == 0) A.sweeps[(size_t)b * A.ndt_act + it] = itp | (nchk << 16);
#else
    if (A.sweeps && tid == 0) A.sweeps[(size_t)b * A.ndt_act + it] = itp;
#endif

    cyc_j += __builtin_amdgcn_s_memtime() - cyc_j0;
    asm volatile("" : "+v"(j));   // keep hipcc from hoisting (and spilling) a whole timestep's LDS addresses out of the loop
    BCN_PH(2)
    // ---- p += phi (rayleigh.py:219), corrector (rayleigh.py:460-464) -> LDS u, v --------------
#pragma unroll
    for (int k = 0; k < R; k++) {
      const int i = i0 + k;
      const real ph = phA[k];
      const real pw = (k > 0) ? phA[k > 0 ? k - 1 : 0] : hW;
      const real ps = from_below(ph, ph);
      p[k] += ph;
      if (active) {
        if (i >= 2 && LIVE(k)) Ul[i * SY + j] = us[k] - dt * (ph - pw) * rdx;
        if (j >= 2 && LIVE(k)) Vl[i * SY + j] = vs[k] - dt * (ph - ps) * rdy;
      }
    }
    __syncthreads();

    asm volatile("" : "+v"(j));   // keep hipcc from hoisting (and spilling) a whole timestep's LDS addresses out of the loop
    BCN_PH(3)
    // ---- transport, explicit part of every cell (rayleigh.py:468-487); next to it, with the corrected u, v: the velocity
    //      boundary conditions and the p exchange of the NEXT timestep of this unit ------------------------------
    {
      real Ac[R];
      // TREG: the thread's row of T plus the column to its east in ONE round of loads (the neighbouring strip's first column,
      // or the east ghost the boundary conditions stored); the north value from the lane above, the top ghost computed
      real Tx[TREG ? R + 1 : 1];
      if constexpr (TREG) {
#pragma unroll
        for (int k = 0; k <= R; k++) Tx[k] = Tl[(i0 + k) * SY + j];
      }
#pragma unroll
      for (int k = 0; k < R; k++) {
        const int c = (i0 + k) * SY + j;
        const real uE = Ul[c + SY], uW = Ul[c], vN = Vl[c + 1], vS = Vl[c];
        real T0, TE, TN;
        if constexpr (TREG) {
          T0 = Tx[TREG ? k : 0]; TE = Tx[TREG ? k + 1 : 0];
          TN = from_above(real(0), T0);
          TN = (lane == NY - 1) ? 2 * A.Tc - T0 : TN;      // the top ghost (rayleigh.py:188), as the boundary conditions stored it
        } else {
          T0 = Tl[c]; TE = Tl[c + SY]; TN = Tl[c + 1];
        }
        const real expl = A.ksc * ((TE - 2 * T0) * rdx2 + (TN - 2 * T0) * rdy2) -
                          (uE * real(0.5) * (TE + T0) - uW * real(0.5) * T0) * rdx -
                          (vN * real(0.5) * (TN + T0) - vS * real(0.5) * T0) * rdy;
        Ac[k] = T0 + dt * expl;
      }
      if (it + 1 < it_end && status == 0) {
        bc_uv();
        publish_p();
      }
      __syncthreads();   // every read of the old T is done
      if (active) {
#pragma unroll
        for (int k = 0; k < R; k++) if (LIVE(k)) Tl[(i0 + k) * SY + j] = Ac[k];
      }
    }
    __syncthreads();
    BCN_PH(4)
  }
#undef BCN_NCHK_INC

  // ---- store: LDS [i][j] -> HBM [j][i]; p and its ghosts -------------------------------------
  for (int c = tid; c < SX * SY; c += NT) {
    const int jj = c / SX, ii = c - jj * SX;
    gu[c] = Ul[ii * SY + jj];
    gv[c] = Vl[ii * SY + jj];
    gS[c] = Tl[ii * SY + jj];
  }
  if (active) {
#pragma unroll
    for (int k = 0; k < R; k++) {
      const int i = i0 + k, c = j * SX + i;
      // a p ghost receives the same increments as its interior neighbour (phi ghosts copy it)
      if (!LIVE(k)) continue;
      const real dp = p[k] - gp[c];
      if (i == 1) gp[c - 1] += dp;
      if (i == NX) gp[c + 1] += dp;
      if (j == 1) gp[c - SX] += dp;
      if (j == NY && KIND == 0) gp[c + SX] += dp;
      gp[c] = p[k];
    }
  }
  __syncthreads();
#ifdef BCN_STAMP   // diagnostic build only: shader clock in MHz (s_memrealtime ticks at 100 MHz)
  {
    const unsigned long long kt1 = __builtin_amdgcn_s_memtime(), kr1 = __builtin_amdgcn_s_memrealtime();
    status = (int)((kt1 - kt0) * 100ull / (kr1 - kr0 + 1));
#ifndef BCN_STAMP_WAVE
#define BCN_STAMP_WAVE 0     // the wave whose phase times are reported
#endif
    if (tid == 64 * BCN_STAMP_WAVE && A.actions_norm)
      for (int q = 0; q < 7; q++) A.actions_norm[(size_t)b * A.n_sgts + q] = (real)seg[q] / (real)(it_end - it_begin);

  }
#endif
  if (last_chunk) {
    ns2d_finish<real, NT>(A, b, gu, gv, gS, status, red);
  } else if (tid == 0) {
    A.status[b] = status;
  }
#undef LIVE
  if (tid == 0 && A.cyc) {   // this replica's units run one after the other (chunk hand-off): plain read-modify-write
    A.cyc[4 * (size_t)b] += cyc_j;
    A.cyc[4 * (size_t)b + 1] += __builtin_amdgcn_s_memtime() - cyc_u0;
    A.cyc[4 * (size_t)b + 2] += (unsigned long long)prev_sweeps[2];
    A.cyc[4 * (size_t)b + 3] += (unsigned long long)prev_sweeps[3];
  }
}

// one unit of work, every wave with the body of its strip width (both bodies execute the same barriers)
template <typename real, int NX, int NY, int R, int KIND, bool EQ, int GF>
__device__ __forceinline__ void fast_unit(const NS2DArgs<real>& A, const int b, const int it_begin, const int it_end,
                                          const bool first_chunk, const bool last_chunk, char* smem) {
  using G = FastGeom<NX, NY, R, GF>;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // (GF == 2 -- float64, T in the global scratch --: ONE body, the narrow last strip's surplus columns dead: fast_body, DEADC)
  if (GF != 2 && G::RL != R && w == G::NW - 1) fast_body<real, NX, NY, R, G::RL, KIND, EQ, GF>(A, w, b, it_begin, it_end, first_chunk, last_chunk, smem);
  else fast_body<real, NX, NY, R, R, KIND, EQ, GF>(A, w, b, it_begin, it_end, first_chunk, last_chunk, smem);
}

// plain launch: one workgroup per replica, timesteps [A.it_begin, A.it_end)
template <typename real, int NX, int NY, int R, int KIND, bool EQ, int GF>
__global__ __launch_bounds__((FastGeom<NX, NY, R, GF>::NT)) void ns2d_fast_step(NS2DArgs<real> A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = A.order ? A.order[blockIdx.x] : (int)blockIdx.x;
  if (A.mask && !A.mask[b]) return;
  fast_unit<real, NX, NY, R, KIND, EQ, GF>(A, b, A.it_begin, A.it_end, A.first_chunk != 0, A.last_chunk != 0, smem);
}

// ---- ticketed chunk scheduler (ns2d_sched.h) ---------------------------------------------------
template <typename real, int NX, int NY, int R, int KIND, bool EQ, int GF>
__global__ __launch_bounds__((FastGeom<NX, NY, R, GF>::NT)) void ns2d_fast_sched(NS2DArgs<real> A, SchedCtl* ctl, int batch, int nchunk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // two words behind fast_body's scalars (no static __shared__ in front of the dynamic region)
  unsigned int* s_words = reinterpret_cast<unsigned int*>(reinterpret_cast<real*>(smem) +
                                                          FastGeom<NX, NY, R, GF>::EXCH + 224);
  ns2d_sched_loop<real>(A, ctl, batch, nchunk, s_words, [&](int b, int it0, int it1, bool first, bool last) {
    fast_unit<real, NX, NY, R, KIND, EQ, GF>(A, b, it0, it1, first, last, smem);
  });
}

// ---- LPT ordering between the two launches of one step ---------------------------------------
// A replica's Jacobi work varies ~10x across a batch (20..240 sweeps per timestep) and each
// replica is a serial chain on one CU, so with ~2 replicas per CU the step time is set by
// which CU happens to pick up a long replica late.  The work of the first Q timesteps
// predicts the rest (r ~ 0.8), so the step runs as [0,Q) in index order, then the remaining
// timesteps with replicas dispatched longest-first.
__global__ __launch_bounds__(1024) void ns2d_rank_by_work(const int32_t* sweeps, int ndt, int q, int batch,
                                                          int32_t* order, const uint8_t* mask) {
  __shared__ int key[2048];
  for (int b = threadIdx.x; b < batch; b += blockDim.x) {
    int s = 0;
    if (!mask || mask[b])
      for (int t = 0; t < q; t++) s += sweeps[(size_t)b * ndt + t];
    key[b] = s;
  }
  __syncthreads();
  for (int b = threadIdx.x; b < batch; b += blockDim.x) {
    const int mine = key[b];
    int rank = 0;
    for (int o = 0; o < batch; o++) rank += (key[o] > mine || (key[o] == mine && o < b)) ? 1 : 0;
    order[rank] = b;
  }
}

template <typename real, int NX, int NY, int R, int KIND, bool EQ, int GF>
int launch_fast_eq(const NS2DArgs<real>& a, int batch, hipStream_t s) {
  using G = FastGeom<NX, NY, R, GF>;
  if (GF && (!a.fscr || a.fscr_stride < G::scratch_elems())) { bcn_set_error("fast path: field scratch missing"); return BCN_ERR_UNSUPPORTED; }
  const size_t lds = G::template lds_bytes<real>();
  auto k = ns2d_fast_step<real, NX, NY, R, KIND, EQ, GF>;
  static unsigned long long attr_set = 0;
  if (ns2d_first_on_device(attr_set)) {
    BCN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  NS2DArgs<real> c = a;
  if (!c.sweeps) c.sweeps = c.sweeps_int;
  const SchedParams sp = ns2d_sched_params(a);
  const int mode = sp.mode, sched_grid = sp.grid, SQ = sp.q;
  if (mode == 2 && batch > sched_grid && a.ndt_act >= 2 * SQ && a.sched_ctl) {
    auto ks = ns2d_fast_sched<real, NX, NY, R, KIND, EQ, GF>;
    static unsigned long long attr_set2 = 0;
    if (ns2d_first_on_device(attr_set2)) {
      BCN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ks), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    int nchunk = 0;
    ns2d_sched_chunks(a.ndt_act, SQ, (a.host ? a.host->sched_tail : 0), &c.sched_nbig, &nchunk);
    c.sched_q = SQ; c.order = nullptr; c.first_chunk = 1; c.last_chunk = 1; c.it_begin = 0; c.it_end = a.ndt_act;
    BCN_HIP(hipMemsetAsync(a.sched_ctl, 0, a.sched_bytes, s));
    hipLaunchKernelGGL(ks, dim3(sched_grid), dim3(G::NT), lds, s, c, static_cast<SchedCtl*>(a.sched_ctl), batch, nchunk);
    BCN_HIP(hipGetLastError());
    if (a.host) a.host->launched = "ns2d_fast_sched";
    return BCN_OK;
  }
  // split only when replicas outnumber the CUs (otherwise every replica starts at once and
  // the order cannot matter); BCN_LPT_MIN_BATCH / bcn_set_sched override the threshold (tests)
  const int min_batch = sp.lpt_min_batch;
  constexpr int Q = 10;
  const bool split = mode >= 1 && batch >= min_batch && batch <= 2048 && a.ndt_act >= 4 * Q;
  c.first_chunk = 1; c.order = nullptr; c.it_begin = 0;
  if (a.sched_ctl) BCN_HIP(hipMemsetAsync(a.sched_ctl, 0, a.sched_bytes, s));   // cycle counters
  if (!split) {
    c.it_end = a.ndt_act; c.last_chunk = 1;
    hipLaunchKernelGGL(k, dim3(batch), dim3(G::NT), lds, s, c);
  } else {
    c.it_end = Q; c.last_chunk = 0;
    hipLaunchKernelGGL(k, dim3(batch), dim3(G::NT), lds, s, c);
    hipLaunchKernelGGL(ns2d_rank_by_work, dim3(1), dim3(1024), 0, s, c.sweeps, a.ndt_act, Q, batch, c.order_out, c.mask);
    c.first_chunk = 0; c.last_chunk = 1; c.it_begin = Q; c.it_end = a.ndt_act; c.order = c.order_out;
    hipLaunchKernelGGL(k, dim3(batch), dim3(G::NT), lds, s, c);
  }
  BCN_HIP(hipGetLastError());
  if (a.host) a.host->launched = "ns2d_fast_step";
  return BCN_OK;
}

template <typename real, int NX, int NY, int R, int KIND, int GF = 0>
int launch_fast(const NS2DArgs<real>& a, int batch, hipStream_t s) {
  static_assert(KIND == 0, "fast_body applies the rayleigh boundary conditions (no moving walls): mixing runs ns2d_fast2");
  // dx == dy (every reference configuration): one multiply per cell instead of two
  if (a.cx == a.cy) return launch_fast_eq<real, NX, NY, R, KIND, true, GF>(a, batch, s);
  return launch_fast_eq<real, NX, NY, R, KIND, false, GF>(a, batch, s);
}


}  // namespace
