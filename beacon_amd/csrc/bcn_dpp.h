// bcn_dpp.h -- cross-lane primitives for the register-resident kernels (gfx950, wave64).
#pragma once
#include "bcn_common.h"

namespace bcn_dpp {

// ---- DPP primitives (verified on gfx950 by scripts/dpp_test.hip) ----------------------------
template <int CTRL, int RM, int BM, bool BC>
__device__ __forceinline__ float dpp(float oldv, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, oldv),
                                                               __builtin_bit_cast(int, v), CTRL, RM, BM, BC));
}
template <int CTRL, int RM, int BM, bool BC>
__device__ __forceinline__ double dpp(double oldv, double v) {
  const long long o = __builtin_bit_cast(long long, oldv), s = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_update_dpp((int)o, (int)s, CTRL, RM, BM, BC);
  const int hi = __builtin_amdgcn_update_dpp((int)(o >> 32), (int)(s >> 32), CTRL, RM, BM, BC);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
// value of the lane below (row j-1); lane 0 keeps `oldv`
template <typename real> __device__ __forceinline__ real from_below(real oldv, real v) { return dpp<0x138, 0xf, 0xf, false>(oldv, v); }
// value of the lane above (row j+1); lane 63 keeps `oldv`
template <typename real> __device__ __forceinline__ real from_above(real oldv, real v) { return dpp<0x130, 0xf, 0xf, false>(oldv, v); }

// acc + north + south neighbour (values of the lanes above and below, 0 outside the wave) as two
// fused v_add_f32_dpp.  hipcc fuses the DPP move into the add only for some of the stencil cells, so
// the pair is spelled out.  hipcc does not model instructions inside an asm statement: it once
// scheduled the VALU write of `c` directly in front of this statement (VALU write -> DPP read of the
// same VGPR needs 2 wait states; symptom: replicas with identical inputs diverged), hence the
// leading s_nop 1.  The second add reads `c` by DPP again and `t` as a plain operand: no hazard.
__device__ __forceinline__ float add_above_below(float acc, float c) {
  float t, r;
  asm("s_nop 1\n\t"
      "v_add_f32_dpp %0, %2, %3 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %1, %2, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
      : "=&v"(t), "=v"(r)
      : "v"(c), "v"(acc));
  return r;
}
// One Jacobi cell of the one-row-per-lane kernels with dx == dy, as ONE statement: cx * ((e + w) + north + south) + (cBy * c + nb).
// The same operations in the same order as the separate statements (results are bit-identical), but hipcc can no longer
// re-order WITHIN the cell: left to itself it sometimes hoists the q-fma in front of the DPP pair and then has to put an
// `s_nop 0` between the second DPP add and the fma that consumes it (DPP result -> VALU read), and the whole double sweep
// came out anywhere between 817 and 895 cycles depending on unrelated code around the loop.  The q-fma sits between the
// DPP pair and its consumer, where it fills that wait state for free.
__device__ __forceinline__ float jacobi_cell_eq(float c, float e, float wv, float nbk, float cx, float cBy) {
  float t, t2, q, ph;
  asm("v_add_f32 %0, %5, %6\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %1, %4, %0 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %0, %4, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_fma_f32 %2, %9, %4, %7\n\t"
      "v_fma_f32 %3, %8, %0, %2"
      : "=&v"(t), "=&v"(t2), "=&v"(q), "=v"(ph)
      : "v"(c), "v"(e), "v"(wv), "v"(nbk), "v"(cx), "v"(cBy));
  return ph;
}
__device__ __forceinline__ double add_above_below(double acc, double c) {
  return acc + dpp<0x130, 0xf, 0xf, true>(0.0, c) + dpp<0x138, 0xf, 0xf, true>(0.0, c);
}

// Two-rows-per-lane stencils (ns2d_fast2.hip): lo + (hi of the lane below) and hi + (lo of the lane
// above), 0 outside the wave, as fused v_add_f32_dpp (same s_nop rule as above).
__device__ __forceinline__ void add_pair_neighbours(float lo, float hi, float& s_plus_hi, float& lo_plus_n) {
  asm("s_nop 1\n\t"
      "v_add_f32_dpp %0, %3, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %1, %2, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
      : "=&v"(s_plus_hi), "=&v"(lo_plus_n)
      : "v"(lo), "v"(hi));
}
__device__ __forceinline__ void add_pair_neighbours(double lo, double hi, double& s_plus_hi, double& lo_plus_n) {
  s_plus_hi = dpp<0x138, 0xf, 0xf, true>(0.0, hi) + hi;
  lo_plus_n = lo + dpp<0x130, 0xf, 0xf, true>(0.0, lo);
}

template <typename real>
__device__ __forceinline__ real row16_sum(real s) {  // lane 15 of each 16-lane row: sum of the row
  s += dpp<0x111, 0xf, 0xf, true>(real(0), s);
  s += dpp<0x112, 0xf, 0xf, true>(real(0), s);
  s += dpp<0x114, 0xf, 0xf, true>(real(0), s);
  s += dpp<0x118, 0xf, 0xf, true>(real(0), s);
  return s;
}
template <typename real>
__device__ __forceinline__ real wave_sum_lane63(real s) {  // lane 63: sum over the wave
  s = row16_sum(s);
  s += dpp<0x142, 0xa, 0xf, false>(real(0), s);
  s += dpp<0x143, 0xc, 0xf, false>(real(0), s);
  return s;
}
// the same with max, for NON-NEGATIVE values (lanes without a source read 0)
__device__ __forceinline__ float bcn_fmax(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ double bcn_fmax(double a, double b) { return __builtin_fmax(a, b); }
template <typename real>
__device__ __forceinline__ real row16_max(real s) {
  s = bcn_fmax(s, dpp<0x111, 0xf, 0xf, true>(real(0), s));
  s = bcn_fmax(s, dpp<0x112, 0xf, 0xf, true>(real(0), s));
  s = bcn_fmax(s, dpp<0x114, 0xf, 0xf, true>(real(0), s));
  s = bcn_fmax(s, dpp<0x118, 0xf, 0xf, true>(real(0), s));
  return s;
}
template <typename real>
__device__ __forceinline__ real wave_max_lane63(real s) {
  s = row16_max(s);
  s = bcn_fmax(s, dpp<0x142, 0xa, 0xf, false>(real(0), s));
  s = bcn_fmax(s, dpp<0x143, 0xc, 0xf, false>(real(0), s));
  return s;
}
__device__ __forceinline__ float read_lane(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ double read_lane(double v, int l) {
  const long long s = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_readlane((int)s, l), hi = __builtin_amdgcn_readlane((int)(s >> 32), l);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

}  // namespace bcn_dpp
