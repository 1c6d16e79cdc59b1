// ubench3.hip -- per-WAVE issue interval of instruction patterns that occur in the Jacobi sweep (gfx950), with
// 1, 2 and 4 waves per SIMD: cycles per instruction as seen by one wave (s_memtime around an unrolled block of
// inline asm, so the instruction stream is exactly what is written here).
// Build: hipcc --offload-arch=gfx950 -O3 scripts/ubench3.hip -o scripts/ubench3
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

// every pattern is 8 "work" instructions long (N_INSTR says how many instructions of any kind it holds)
#define P_IND    "v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n"
#define P_DEP    "v_add_f32 %0, %0, %4\n v_add_f32 %0, %0, %4\n v_add_f32 %0, %0, %4\n v_add_f32 %0, %0, %4\n v_add_f32 %0, %0, %4\n v_add_f32 %0, %0, %4\n v_add_f32 %0, %0, %4\n v_add_f32 %0, %0, %4\n"
#define P_FMA    "v_fma_f32 %0, %0, %4, %1\n v_fma_f32 %1, %1, %4, %2\n v_fma_f32 %2, %2, %4, %3\n v_fma_f32 %3, %3, %4, %0\n v_fma_f32 %0, %0, %4, %1\n v_fma_f32 %1, %1, %4, %2\n v_fma_f32 %2, %2, %4, %3\n v_fma_f32 %3, %3, %4, %0\n"
// DPP add whose plain operand was produced by the instruction right in front of it (the sweep's e+w -> +n -> +s)
#define P_DPPDEP "v_add_f32 %0, %1, %2\n v_add_f32_dpp %0, %3, %0 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %0, %3, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fma_f32 %1, %0, %4, %1\n v_add_f32 %0, %1, %2\n v_add_f32_dpp %0, %3, %0 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %0, %3, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fma_f32 %2, %0, %4, %2\n"
// the same with the kernel's s_nop 1 in front of each DPP pair (10 instructions)
#define P_DPPNOP "v_add_f32 %0, %1, %2\n s_nop 1\n v_add_f32_dpp %0, %3, %0 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %0, %3, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fma_f32 %1, %0, %4, %1\n v_add_f32 %0, %1, %2\n s_nop 1\n v_add_f32_dpp %0, %3, %0 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %0, %3, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fma_f32 %2, %0, %4, %2\n"
// DPP adds that do not depend on their predecessor
#define P_DPPIND "v_add_f32_dpp %0, %3, %0 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %3, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %2, %3, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %0, %3, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %3, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %2, %3, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %0, %3, %0 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %3, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
// row-local DPP (row_shr / row_shl inside 16 lanes) instead of the wave-wide shifts
#define P_DPPROW "v_add_f32_dpp %0, %3, %0 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %3, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %2, %3, %2 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %0, %3, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %3, %1 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %2, %3, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %0, %3, %0 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %3, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
// VALU with a scalar instruction after each (does SALU take one of the wave's issue turns?) (16 instructions)
#define P_SALU   "v_add_f32 %0, %0, %4\n s_add_u32 s20, s20, 1\n v_add_f32 %1, %1, %4\n s_add_u32 s21, s21, 1\n v_add_f32 %2, %2, %4\n s_add_u32 s20, s20, 1\n v_add_f32 %3, %3, %4\n s_add_u32 s21, s21, 1\n v_add_f32 %0, %0, %4\n s_add_u32 s20, s20, 1\n v_add_f32 %1, %1, %4\n s_add_u32 s21, s21, 1\n v_add_f32 %2, %2, %4\n s_add_u32 s20, s20, 1\n v_add_f32 %3, %3, %4\n s_add_u32 s21, s21, 1\n"
#define P_NOP    "v_add_f32 %0, %0, %4\n s_nop 0\n v_add_f32 %1, %1, %4\n s_nop 0\n v_add_f32 %2, %2, %4\n s_nop 0\n v_add_f32 %3, %3, %4\n s_nop 0\n v_add_f32 %0, %0, %4\n s_nop 0\n v_add_f32 %1, %1, %4\n s_nop 0\n v_add_f32 %2, %2, %4\n s_nop 0\n v_add_f32 %3, %3, %4\n s_nop 0\n"

// ---- whole 4-cell blocks of the sweep: s = %4..%7 (read only), t = %0..%3, d = %8..%11, acc = %12, coefficient %13 ----
#define UP " wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define DN " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
// B0: phase-ordered (4 adds, 4 fma, 4 DPP up, 4 DPP down, 4 fmac, 4 sub, 4 acc) = variant E
#define B0 "v_add_f32 %0, %5, %4\n v_add_f32 %1, %6, %4\n v_add_f32 %2, %7, %5\n v_add_f32 %3, %4, %6\n" \
           "v_fma_f32 %8, %13, %4, %5\n v_fma_f32 %9, %13, %5, %6\n v_fma_f32 %10, %13, %6, %7\n v_fma_f32 %11, %13, %7, %4\n" \
           "v_add_f32_dpp %0, %4, %0" UP "v_add_f32_dpp %1, %5, %1" UP "v_add_f32_dpp %2, %6, %2" UP "v_add_f32_dpp %3, %7, %3" UP \
           "v_add_f32_dpp %0, %4, %0" DN "v_add_f32_dpp %1, %5, %1" DN "v_add_f32_dpp %2, %6, %2" DN "v_add_f32_dpp %3, %7, %3" DN \
           "v_fmac_f32 %8, %13, %0\n v_fmac_f32 %9, %13, %1\n v_fmac_f32 %10, %13, %2\n v_fmac_f32 %11, %13, %3\n" \
           "v_sub_f32 %0, %8, %4\n v_sub_f32 %1, %9, %5\n v_sub_f32 %2, %10, %6\n v_sub_f32 %3, %11, %7\n" \
           "v_fmac_f32 %12, %0, %0\n v_fmac_f32 %12, %1, %1\n v_fmac_f32 %12, %2, %2\n v_fmac_f32 %12, %3, %3\n"
// B1: cell by cell as the kernel today: add, s_nop 1, DPP up, DPP down, fma, fmac, sub, acc
#define C1(t, s, e, w, d) "v_add_f32 " t ", " e ", " w "\n s_nop 1\n v_add_f32_dpp " t ", " s ", " t UP "v_add_f32_dpp " t ", " s ", " t DN \
           "v_fma_f32 " d ", %13, " s ", " e "\n v_fmac_f32 " d ", %13, " t "\n v_sub_f32 " t ", " d ", " s "\n v_fmac_f32 %12, " t ", " t "\n"
#define B1 C1("%0", "%4", "%5", "%7", "%8") C1("%1", "%5", "%6", "%4", "%9") C1("%2", "%6", "%7", "%5", "%10") C1("%3", "%7", "%4", "%6", "%11")
// B2: every DPP add separated from the next DPP add by one plain instruction
#define B2 "v_add_f32 %0, %5, %4\n v_add_f32 %1, %6, %4\n" \
           "v_add_f32_dpp %0, %4, %0" UP "v_add_f32 %2, %7, %5\n v_add_f32_dpp %1, %5, %1" UP "v_add_f32 %3, %4, %6\n" \
           "v_add_f32_dpp %0, %4, %0" DN "v_fma_f32 %8, %13, %4, %5\n v_add_f32_dpp %1, %5, %1" DN "v_fma_f32 %9, %13, %5, %6\n" \
           "v_add_f32_dpp %2, %6, %2" UP "v_fma_f32 %10, %13, %6, %7\n v_add_f32_dpp %3, %7, %3" UP "v_fma_f32 %11, %13, %7, %4\n" \
           "v_add_f32_dpp %2, %6, %2" DN "v_fmac_f32 %8, %13, %0\n v_add_f32_dpp %3, %7, %3" DN "v_fmac_f32 %9, %13, %1\n" \
           "v_sub_f32 %0, %8, %4\n v_fmac_f32 %10, %13, %2\n v_sub_f32 %1, %9, %5\n v_fmac_f32 %11, %13, %3\n" \
           "v_fmac_f32 %12, %0, %0\n v_sub_f32 %2, %10, %6\n v_fmac_f32 %12, %1, %1\n v_sub_f32 %3, %11, %7\n v_fmac_f32 %12, %2, %2\n v_fmac_f32 %12, %3, %3\n"
// B3: as B2 with TWO plain instructions between DPP adds where available
#define B3 "v_add_f32 %0, %5, %4\n v_add_f32 %1, %6, %4\n v_add_f32 %2, %7, %5\n" \
           "v_add_f32_dpp %0, %4, %0" UP "v_add_f32 %3, %4, %6\n v_fma_f32 %8, %13, %4, %5\n v_add_f32_dpp %1, %5, %1" UP "v_fma_f32 %9, %13, %5, %6\n v_fma_f32 %10, %13, %6, %7\n" \
           "v_add_f32_dpp %0, %4, %0" DN "v_fma_f32 %11, %13, %7, %4\n v_add_f32_dpp %2, %6, %2" UP "v_nop\n v_add_f32_dpp %1, %5, %1" DN "v_fmac_f32 %8, %13, %0\n" \
           "v_add_f32_dpp %3, %7, %3" UP "v_sub_f32 %0, %8, %4\n v_add_f32_dpp %2, %6, %2" DN "v_fmac_f32 %9, %13, %1\n v_fmac_f32 %12, %0, %0\n" \
           "v_add_f32_dpp %3, %7, %3" DN "v_sub_f32 %1, %9, %5\n v_fmac_f32 %10, %13, %2\n v_fmac_f32 %12, %1, %1\n v_sub_f32 %2, %10, %6\n" \
           "v_fmac_f32 %11, %13, %3\n v_fmac_f32 %12, %2, %2\n v_sub_f32 %3, %11, %7\n v_fmac_f32 %12, %3, %3\n"
// B4: only the 8 DPP adds of a block (what they cost alone), B5: only the 20 plain instructions
#define B4 "v_add_f32_dpp %0, %4, %0" UP "v_add_f32_dpp %1, %5, %1" UP "v_add_f32_dpp %2, %6, %2" UP "v_add_f32_dpp %3, %7, %3" UP \
           "v_add_f32_dpp %0, %4, %0" DN "v_add_f32_dpp %1, %5, %1" DN "v_add_f32_dpp %2, %6, %2" DN "v_add_f32_dpp %3, %7, %3" DN
#define B5 "v_add_f32 %0, %5, %4\n v_add_f32 %1, %6, %4\n v_add_f32 %2, %7, %5\n v_add_f32 %3, %4, %6\n" \
           "v_fma_f32 %8, %13, %4, %5\n v_fma_f32 %9, %13, %5, %6\n v_fma_f32 %10, %13, %6, %7\n v_fma_f32 %11, %13, %7, %4\n" \
           "v_fmac_f32 %8, %13, %0\n v_fmac_f32 %9, %13, %1\n v_fmac_f32 %10, %13, %2\n v_fmac_f32 %11, %13, %3\n" \
           "v_sub_f32 %0, %8, %4\n v_sub_f32 %1, %9, %5\n v_sub_f32 %2, %10, %6\n v_sub_f32 %3, %11, %7\n" \
           "v_fmac_f32 %12, %0, %0\n v_fmac_f32 %12, %1, %1\n v_fmac_f32 %12, %2, %2\n v_fmac_f32 %12, %3, %3\n"

// B6: B0 with s_nop 1 in front of the DPP group; B7: B1 without its s_nop; B8: two cells interleaved, one s_nop 1 per pair of cells
#define B6 "v_add_f32 %0, %5, %4\n v_add_f32 %1, %6, %4\n v_add_f32 %2, %7, %5\n v_add_f32 %3, %4, %6\n" \
           "v_fma_f32 %8, %13, %4, %5\n v_fma_f32 %9, %13, %5, %6\n v_fma_f32 %10, %13, %6, %7\n v_fma_f32 %11, %13, %7, %4\n s_nop 1\n" \
           "v_add_f32_dpp %0, %4, %0" UP "v_add_f32_dpp %0, %4, %0" DN "v_add_f32_dpp %1, %5, %1" UP "v_add_f32_dpp %1, %5, %1" DN \
           "v_add_f32_dpp %2, %6, %2" UP "v_add_f32_dpp %2, %6, %2" DN "v_add_f32_dpp %3, %7, %3" UP "v_add_f32_dpp %3, %7, %3" DN \
           "v_fmac_f32 %8, %13, %0\n v_fmac_f32 %9, %13, %1\n v_fmac_f32 %10, %13, %2\n v_fmac_f32 %11, %13, %3\n" \
           "v_sub_f32 %0, %8, %4\n v_sub_f32 %1, %9, %5\n v_sub_f32 %2, %10, %6\n v_sub_f32 %3, %11, %7\n" \
           "v_fmac_f32 %12, %0, %0\n v_fmac_f32 %12, %1, %1\n v_fmac_f32 %12, %2, %2\n v_fmac_f32 %12, %3, %3\n"
#define C7(t, s, e, w, d) "v_add_f32 " t ", " e ", " w "\n v_add_f32_dpp " t ", " s ", " t UP "v_add_f32_dpp " t ", " s ", " t DN \
           "v_fma_f32 " d ", %13, " s ", " e "\n v_fmac_f32 " d ", %13, " t "\n v_sub_f32 " t ", " d ", " s "\n v_fmac_f32 %12, " t ", " t "\n"
#define B7 C7("%0", "%4", "%5", "%7", "%8") C7("%1", "%5", "%6", "%4", "%9") C7("%2", "%6", "%7", "%5", "%10") C7("%3", "%7", "%4", "%6", "%11")
#define C8(t, s, e, w, d, t2, s2, e2, w2, d2) "v_add_f32 " t ", " e ", " w "\n v_add_f32 " t2 ", " e2 ", " w2 "\n s_nop 1\n" \
           "v_add_f32_dpp " t ", " s ", " t UP "v_add_f32_dpp " t ", " s ", " t DN "v_add_f32_dpp " t2 ", " s2 ", " t2 UP "v_add_f32_dpp " t2 ", " s2 ", " t2 DN \
           "v_fma_f32 " d ", %13, " s ", " e "\n v_fma_f32 " d2 ", %13, " s2 ", " e2 "\n v_fmac_f32 " d ", %13, " t "\n v_fmac_f32 " d2 ", %13, " t2 "\n" \
           "v_sub_f32 " t ", " d ", " s "\n v_sub_f32 " t2 ", " d2 ", " s2 "\n v_fmac_f32 %12, " t ", " t "\n v_fmac_f32 %12, " t2 ", " t2 "\n"
#define B8 C8("%0", "%4", "%5", "%7", "%8", "%1", "%5", "%6", "%4", "%9") C8("%2", "%6", "%7", "%5", "%10", "%3", "%7", "%4", "%6", "%11")
// B9: B1 with the ghost fma moved in front of the DPP pair instead of the s_nop (7 instructions per cell)
#define C9(t, s, e, w, d) "v_add_f32 " t ", " e ", " w "\n v_fma_f32 " d ", %13, " s ", " e "\n v_add_f32_dpp " t ", " s ", " t UP "v_add_f32_dpp " t ", " s ", " t DN \
           "v_fmac_f32 " d ", %13, " t "\n v_sub_f32 " t ", " d ", " s "\n v_fmac_f32 %12, " t ", " t "\n"
#define B9 C9("%0", "%4", "%5", "%7", "%8") C9("%1", "%5", "%6", "%4", "%9") C9("%2", "%6", "%7", "%5", "%10") C9("%3", "%7", "%4", "%6", "%11")
// B10: B9 with s_nop 0 between the fma and the DPP pair
#define C10(t, s, e, w, d) "v_add_f32 " t ", " e ", " w "\n v_fma_f32 " d ", %13, " s ", " e "\n s_nop 0\n v_add_f32_dpp " t ", " s ", " t UP "v_add_f32_dpp " t ", " s ", " t DN \
           "v_fmac_f32 " d ", %13, " t "\n v_sub_f32 " t ", " d ", " s "\n v_fmac_f32 %12, " t ", " t "\n"
#define B10 C10("%0", "%4", "%5", "%7", "%8") C10("%1", "%5", "%6", "%4", "%9") C10("%2", "%6", "%7", "%5", "%10") C10("%3", "%7", "%4", "%6", "%11")

template <int MODE>
__global__ __launch_bounds__(1024) void kb(unsigned long long* out, float* sink) {
  float t0 = 0, t1 = 0, t2 = 0, t3 = 0, d0 = 0, d1 = 0, d2 = 0, d3 = 0, acc = 0;
  const float s0 = threadIdx.x * 1e-3f, s1 = s0 + 1.f, s2 = s0 + 2.f, s3 = s0 + 3.f, cf = 0.25f;
  __syncthreads();
  const unsigned long long t_0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < 64; it++) {
#define KB_OPS : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3) : "v"(s0), "v"(s1), "v"(s2), "v"(s3), "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(acc), "v"(cf)
    if (MODE == 0) asm volatile(REP16(B0) KB_OPS);
    if (MODE == 1) asm volatile(REP16(B1) KB_OPS);
    if (MODE == 2) asm volatile(REP16(B2) KB_OPS);
    if (MODE == 3) asm volatile(REP16(B3) KB_OPS);
    if (MODE == 4) asm volatile(REP16(B4) KB_OPS);
    if (MODE == 5) asm volatile(REP16(B5) KB_OPS);
    if (MODE == 6) asm volatile(REP16(B6) KB_OPS);
    if (MODE == 7) asm volatile(REP16(B7) KB_OPS);
    if (MODE == 8) asm volatile(REP16(B8) KB_OPS);
    if (MODE == 9) asm volatile(REP16(B9) KB_OPS);
    if (MODE == 10) asm volatile(REP16(B10) KB_OPS);
  }
  const unsigned long long t_1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) { out[2 * (threadIdx.x >> 6)] = t_0; out[2 * (threadIdx.x >> 6) + 1] = t_1; }
  sink[threadIdx.x] = t0 + t1 + t2 + t3 + d0 + d1 + d2 + d3 + acc;
}
// ---- a whole sweep assembled from exact pieces (512 threads = 8 waves, 2 per SIMD): what does each piece add? ----
#define RSTEP(c) "s_nop 1\n v_add_f32_dpp %12, %12, %12 " c " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define REDUCE RSTEP("row_shr:1") RSTEP("row_shr:2") RSTEP("row_shr:4") RSTEP("row_shr:8") \
               "s_nop 1\n v_mov_b32_dpp %0, %12 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_add_f32 %12, %12, %0\n" \
               "s_nop 1\n v_mov_b32_dpp %0, %12 row_bcast:31 row_mask:0xc bank_mask:0xf\n v_add_f32 %12, %12, %0\n"
template <int MODE>
__global__ __launch_bounds__(512) void ks(unsigned long long* out, float* sink) {
  __shared__ float lds[4096];
  float t0 = 0, t1 = 0, t2 = 0, t3 = 0, d0 = 0, d1 = 0, d2 = 0, d3 = 0, acc = 0;
  const float s0 = threadIdx.x * 1e-3f, s1 = s0 + 1.f, s2 = s0 + 2.f, s3 = s0 + 3.f, cf = 0.25f;
  const int a0 = threadIdx.x * 4, a1 = ((threadIdx.x + 64) & 511) * 4;
  float r0 = 0, r1 = 0, r2 = 0, r3 = 0;
  lds[threadIdx.x] = 0; lds[threadIdx.x + 512] = 0;
  __syncthreads();
  const unsigned long long t_0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < 512; it++) {
    asm volatile(B1 B1 B1 B1 KB_OPS);                                                     // 16 cells
    if (MODE >= 3) asm volatile(REDUCE KB_OPS);                                           // wave reduction of the residual
    if (MODE >= 2) asm volatile("ds_write_b32 %0, %2\n ds_write_b32 %0, %3 offset:2048\n ds_write_b32 %1, %4 offset:4096\n s_waitcnt lgkmcnt(0)"
                                :: "v"(a0), "v"(a1), "v"(d0), "v"(d1), "v"(acc) : "memory");
    if (MODE >= 1) asm volatile("s_barrier" ::: "memory");
    if (MODE >= 2) asm volatile("ds_read_b32 %0, %4 offset:2048\n ds_read_b32 %1, %4\n ds_read_b128 %2, %5 offset:4096\n"
                                : "=v"(r0), "=v"(r1), "=v"(*(float4*)&t0) : "v"(a0), "v"(a1), "v"(a1 & ~15) : "memory");
    if (MODE >= 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     // reads waited for at once (no cover)
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long t_1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) { out[2 * (threadIdx.x >> 6)] = t_0; out[2 * (threadIdx.x >> 6) + 1] = t_1; }
  sink[threadIdx.x] = t0 + t1 + t2 + t3 + d0 + d1 + d2 + d3 + acc + r0 + r1 + r2 + r3;
}
#define UPR " row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define DNR " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define C1R(t, s, e, w, d) "v_add_f32 " t ", " e ", " w "\n s_nop 1\n v_add_f32_dpp " t ", " s ", " t UPR "v_add_f32_dpp " t ", " s ", " t DNR \
           "v_fma_f32 " d ", %13, " s ", " e "\n v_fmac_f32 " d ", %13, " t "\n v_sub_f32 " t ", " d ", " s "\n v_fmac_f32 %12, " t ", " t "\n"
#define C1P(t, s, e, w, d) "v_add_f32 " t ", " e ", " w "\n s_nop 1\n v_add_f32 " t ", " s ", " t "\n v_add_f32 " t ", " s ", " t "\n" \
           "v_fma_f32 " d ", %13, " s ", " e "\n v_fmac_f32 " d ", %13, " t "\n v_sub_f32 " t ", " d ", " s "\n v_fmac_f32 %12, " t ", " t "\n"
#define B1R C1R("%0", "%4", "%5", "%7", "%8") C1R("%1", "%5", "%6", "%4", "%9") C1R("%2", "%6", "%7", "%5", "%10") C1R("%3", "%7", "%4", "%6", "%11")
#define B1P C1P("%0", "%4", "%5", "%7", "%8") C1P("%1", "%5", "%6", "%4", "%9") C1P("%2", "%6", "%7", "%5", "%10") C1P("%3", "%7", "%4", "%6", "%11")
template <int CELLS, int LDSOP, bool NOBAR = false>
__global__ __launch_bounds__(512) void kx(unsigned long long* out, float* sink) {
  __shared__ float lds[4096];
  float t0 = 0, t1 = 0, t2 = 0, t3 = 0, d0 = 0, d1 = 0, d2 = 0, d3 = 0, acc = 0;
  const float s0 = threadIdx.x * 1e-3f, s1 = s0 + 1.f, s2 = s0 + 2.f, s3 = s0 + 3.f, cf = 0.25f;
  const int a0 = threadIdx.x * 4, a1 = ((threadIdx.x + 64) & 511) * 4;
  float r0 = 0;
  lds[threadIdx.x] = 0; lds[threadIdx.x + 512] = 0;
  __syncthreads();
  const unsigned long long t_0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < 512; it++) {
    if (LDSOP == 1) asm volatile("ds_read_b32 %0, %1\n" : "=v"(r0) : "v"(a1) : "memory");
    if (LDSOP == 2) asm volatile("global_load_dword %0, %1, off\n" : "=v"(r0) : "v"(sink + threadIdx.x) : "memory");
    if (LDSOP == 3) asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n" : "=v"(r0) : "v"(a1) : "memory");   // latency exposed at once
    if (LDSOP == 7 && (threadIdx.x >> 6) == 0) asm volatile("ds_read_b32 %0, %1\n" : "=v"(r0) : "v"(a1) : "memory");   // only wave 0 loads
    if (CELLS == 0 && LDSOP == 4) { asm volatile(B1 B1 KB_OPS); asm volatile("ds_read_b32 %0, %1\n" : "=v"(r0) : "v"(a1) : "memory"); asm volatile(B1 B1 KB_OPS); }
    else if (CELLS == 0 && LDSOP == 5) { asm volatile(B1 B1 B1 B1 KB_OPS); asm volatile("ds_read_b32 %0, %1\n" : "=v"(r0) : "v"(a1) : "memory"); }
    else if (CELLS == 0 && LDSOP == 6) { asm volatile("ds_read_b32 %0, %1\n" : "=v"(r0) : "v"(a1) : "memory"); asm volatile(B1 B1 KB_OPS); asm volatile("s_waitcnt lgkmcnt(0)"); asm volatile(B1 B1 KB_OPS); }
    else if (CELLS == 0 && LDSOP >= 10 && LDSOP < 30) {   // read issued after (LDSOP - 10) cells; 20 + n: waves 4-7 after n + 8 cells instead
      const int n = LDSOP >= 20 ? LDSOP - 20 : LDSOP - 10;
      const bool late = LDSOP >= 20 && (threadIdx.x >> 8);
#define ONECELL(i) { if (i == 0) asm volatile(C1("%0", "%4", "%5", "%7", "%8") KB_OPS); if (i == 1) asm volatile(C1("%1", "%5", "%6", "%4", "%9") KB_OPS); \
                     if (i == 2) asm volatile(C1("%2", "%6", "%7", "%5", "%10") KB_OPS); if (i == 3) asm volatile(C1("%3", "%7", "%4", "%6", "%11") KB_OPS); }
      _Pragma("unroll") for (int c = 0; c < 16; c++) {
        if (c == n && !late) asm volatile("ds_read_b32 %0, %1\n" : "=v"(r0) : "v"(a1) : "memory");
        if (c == n + 8 && late) asm volatile("ds_read_b32 %0, %1\n" : "=v"(r0) : "v"(a1) : "memory");
        ONECELL(c % 4)
      }
    }
    else if (CELLS == 0) asm volatile(B1 B1 B1 B1 KB_OPS);
    if (CELLS == 1) asm volatile(B1R B1R B1R B1R KB_OPS);
    if (CELLS == 2) asm volatile(B1P B1P B1P B1P KB_OPS);
    if (NOBAR) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_barrier" ::: "memory");
  }
  const unsigned long long t_1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) { out[2 * (threadIdx.x >> 6)] = t_0; out[2 * (threadIdx.x >> 6) + 1] = t_1; }
  sink[threadIdx.x] = t0 + t1 + t2 + t3 + d0 + d1 + d2 + d3 + acc + r0;
}
template <int CELLS, int LDSOP, bool NOBAR = false>
void runx(const char* name, unsigned long long* d, float* s) {
  hipLaunchKernelGGL((kx<CELLS, LDSOP, NOBAR>), 1, 512, 0, 0, d, s);
  unsigned long long h[32];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  unsigned long long mn = ~0ull, mx = 0;
  for (int w = 0; w < 8; w++) { if (h[2 * w] < mn) mn = h[2 * w]; if (h[2 * w + 1] > mx) mx = h[2 * w + 1]; }
  printf("%-70s %.0f cycles per sweep\n", name, (double)(mx - mn) / 512.0);
}

// the same pieces in the "edges in the middle" order: loads issued behind the barrier, 8 cells, wait for the loads, 2 edge
// cells (half a block), stores, 6 cells, [reduction], wait for the stores, barrier
#define HB C1("%0", "%4", "%5", "%7", "%8") C1("%1", "%5", "%6", "%4", "%9")
template <int MODE, bool NOBAR = false>
__global__ __launch_bounds__(512) void km(unsigned long long* out, float* sink) {
  __shared__ float lds[4096];
  float t0 = 0, t1 = 0, t2 = 0, t3 = 0, __attribute__((aligned(8))) d0 = 0, d1 = 0, d2 = 0, d3 = 0, acc = 0;
  const float s0 = threadIdx.x * 1e-3f, s1 = s0 + 1.f, s2 = s0 + 2.f, s3 = s0 + 3.f, cf = 0.25f;
  const int a0 = threadIdx.x * 4, a1 = ((threadIdx.x + 64) & 511) * 4;
  float r0 = 0, r1 = 0;
  float4 e4 = {0, 0, 0, 0};
  lds[threadIdx.x] = 0; lds[threadIdx.x + 512] = 0;
  __syncthreads();
  const unsigned long long t_0 = __builtin_amdgcn_s_memtime();
  // MODE: 0 reduction + its store at the end; 1 every store mid-sweep; 2 loads only; 3 stores only; 4 loads + stores, no reduction;
  //       5 = 4 with ONE b64 store and ONE b64 load for the halos + b128 partials; 6 = 4 without the mid-sweep wait
  constexpr bool LOADS = MODE != 3 && MODE < 7, STORES = MODE != 2 && MODE < 7, RED = MODE <= 1;
  for (int it = 0; it < 512; it++) {
    if (LOADS && MODE != 5) asm volatile("ds_read_b32 %0, %3 offset:2048\n ds_read_b32 %1, %3\n ds_read_b128 %2, %4 offset:4096\n"
                 : "=v"(r0), "=v"(r1), "=v"(e4) : "v"(a1), "v"(a1 & ~15) : "memory");
    if (MODE == 5) asm volatile("ds_read_b64 %0, %2\n ds_read_b128 %1, %3 offset:4096\n"
                 : "=v"(*(float2*)&r0), "=v"(e4) : "v"(a1 * 2), "v"(a1 & ~15) : "memory");
    if (MODE == 7 || MODE == 9) asm volatile("ds_read_b32 %0, %1\n" : "=v"(r0) : "v"(a1) : "memory");
    asm volatile(B1 B1 KB_OPS);                                                           // 8 cells
    if (MODE != 6) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (MODE == 8 || MODE == 9) asm volatile("ds_write_b32 %0, %1\n" :: "v"(a0), "v"(d0) : "memory");
    asm volatile(HB KB_OPS);                                                              // the 2 edge cells
    if (STORES && MODE != 5) asm volatile("ds_write_b32 %0, %1\n ds_write_b32 %0, %2 offset:2048\n" :: "v"(a0), "v"(d0), "v"(d1) : "memory");
    if (MODE == 5) asm volatile("ds_write_b64 %0, %1\n" :: "v"(a0 * 2), "v"(*(float2*)&d0) : "memory");
    if (MODE == 1 || (MODE >= 3 && MODE < 7)) asm volatile("ds_write_b32 %0, %1 offset:4096\n" :: "v"(a0), "v"(acc) : "memory");   // lagged: partial published here
    asm volatile(B1 HB KB_OPS);                                                           // 6 cells
    if (MODE == 0) { asm volatile(REDUCE KB_OPS); asm volatile("ds_write_b32 %0, %1 offset:4096\n" :: "v"(a0), "v"(acc) : "memory"); }
    if (MODE == 1) asm volatile(REDUCE KB_OPS);                                           // (its cost, wherever it is placed)
    if (NOBAR) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier" ::: "memory");
  }
  const unsigned long long t_1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) { out[2 * (threadIdx.x >> 6)] = t_0; out[2 * (threadIdx.x >> 6) + 1] = t_1; }
  sink[threadIdx.x] = t0 + t1 + t2 + t3 + d0 + d1 + d2 + d3 + acc + r0 + r1 + e4.x + e4.y + e4.z + e4.w;
}
template <int MODE, bool NOBAR = false>
void runm(const char* name, unsigned long long* d, float* s) {
  hipLaunchKernelGGL((km<MODE, NOBAR>), 1, 512, 0, 0, d, s);
  unsigned long long h[32];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  unsigned long long mn = ~0ull, mx = 0;
  for (int w = 0; w < 8; w++) { if (h[2 * w] < mn) mn = h[2 * w]; if (h[2 * w + 1] > mx) mx = h[2 * w + 1]; }
  printf("%-70s %.0f cycles per sweep\n", name, (double)(mx - mn) / 512.0);
}

template <int MODE>
void runs(const char* name, unsigned long long* d, float* s) {
  hipLaunchKernelGGL(ks<MODE>, 1, 512, 0, 0, d, s);
  unsigned long long h[32];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  unsigned long long mn = ~0ull, mx = 0;
  for (int w = 0; w < 8; w++) { if (h[2 * w] < mn) mn = h[2 * w]; if (h[2 * w + 1] > mx) mx = h[2 * w + 1]; }
  printf("%-70s %.0f cycles per sweep\n", name, (double)(mx - mn) / 512.0);
}

template <int MODE>
void runb(const char* name, unsigned long long* d, float* s) {
  for (int nt : {256, 512, 1024}) {
    hipLaunchKernelGGL(kb<MODE>, 1, nt, 0, 0, d, s);
    unsigned long long h[32];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    unsigned long long mn = ~0ull, mx = 0;
    for (int w = 0; w < nt / 64; w++) { if (h[2 * w] < mn) mn = h[2 * w]; if (h[2 * w + 1] > mx) mx = h[2 * w + 1]; }
    const double per_block = (double)(mx - mn) / (64.0 * 16.0);
    printf("%-58s %d waves/SIMD: %.1f cycles per 4-cell block per wave = %.1f per block per SIMD\n", name, nt / 256, per_block,
           per_block / (nt / 256));
  }
}

template <int MODE>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, float* sink) {
  float a = threadIdx.x * 0.001f, b = a + 1.f, c = a + 2.f, d = a + 3.f;
  const float m = 1.0001f;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < 16; it++) {
    if (MODE == 0) asm volatile(REP16(P_IND) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
    if (MODE == 1) asm volatile(REP16(P_DEP) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
    if (MODE == 2) asm volatile(REP16(P_FMA) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
    if (MODE == 3) asm volatile(REP16(P_DPPDEP) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
    if (MODE == 4) asm volatile(REP16(P_DPPNOP) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
    if (MODE == 5) asm volatile(REP16(P_DPPIND) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
    if (MODE == 6) asm volatile(REP16(P_DPPROW) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
    if (MODE == 7) asm volatile(REP16(P_SALU) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m) : "s20", "s21");
    if (MODE == 8) asm volatile(REP16(P_NOP) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) { out[2 * (threadIdx.x >> 6)] = t0; out[2 * (threadIdx.x >> 6) + 1] = t1; }
  sink[threadIdx.x] = a + b + c + d;
}

template <int MODE>
void run(const char* name, int n_instr, unsigned long long* d, float* s) {
  for (int nt : {256, 512, 768, 1024}) {
    hipLaunchKernelGGL(k<MODE>, 1, nt, 0, 0, d, s);
    unsigned long long h[32];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    unsigned long long mn = ~0ull, mx = 0;
    for (int w = 0; w < nt / 64; w++) { if (h[2 * w] < mn) mn = h[2 * w]; if (h[2 * w + 1] > mx) mx = h[2 * w + 1]; }
    const double per_wave = (double)(mx - mn) / (16.0 * 16.0 * n_instr);
    printf("%-44s %d waves/SIMD: %.2f cycles per instruction per wave = %.2f per instruction per SIMD\n", name, nt / 256,
           per_wave, per_wave / (nt / 256));
  }
}

int main() {
  unsigned long long* d; float* s;
  hipMalloc(&d, 64 * 8); hipMalloc(&s, 1024 * 4);
  run<0>("independent v_add (8)", 8, d, s);
  run<1>("dependent v_add chain (8)", 8, d, s);
  run<2>("v_fma, dependent at distance 3 (8)", 8, d, s);
  run<3>("add, 2 dependent DPP adds, fma (8)", 8, d, s);
  run<4>("add, s_nop 1, 2 dependent DPP adds, fma (10)", 10, d, s);
  run<5>("independent wave_shl/shr DPP adds (8)", 8, d, s);
  run<6>("independent row_shl/shr DPP adds (8)", 8, d, s);
  run<7>("v_add + s_add alternating (16)", 16, d, s);
  run<8>("v_add + s_nop 0 alternating (16)", 16, d, s);
  runb<0>("block: phase-ordered, DPP adds back to back (28)", d, s);
  runb<1>("block: cell by cell with s_nop 1 (kernel today, 32)", d, s);
  runb<2>("block: one plain instruction between DPP adds (28)", d, s);
  runb<3>("block: 1-2 plain instructions between DPP adds (29)", d, s);
  runb<6>("block: phase-ordered, s_nop 1 + DPP pairs (29)", d, s);
  runb<7>("block: cell by cell WITHOUT s_nop (28)", d, s);
  runb<8>("block: two cells interleaved, one s_nop 1 per two cells (30)", d, s);
  runb<9>("block: cell by cell, ghost fma instead of the s_nop (28)", d, s);
  runb<10>("block: cell by cell, ghost fma + s_nop 0 (32)", d, s);
  runb<4>("only the 8 DPP adds", d, s);
  runb<5>("only the 20 plain instructions", d, s);
  runs<0>("sweep pieces: 16 cells only", d, s);
  runs<1>("sweep pieces: 16 cells + s_barrier", d, s);
  runs<2>("sweep pieces: cells + 3 LDS stores + wait + barrier + 3 LDS loads (covered)", d, s);
  runs<3>("sweep pieces: + wave reduction (6 DPP steps behind s_nop 1)", d, s);
  runs<4>("sweep pieces: + loads waited for right behind the barrier", d, s);
  runm<0>("sweep pieces, edges mid-sweep: reduction + its store at the end", d, s);
  runm<1>("sweep pieces, edges mid-sweep: every store mid-sweep (lagged)", d, s);
  runm<2>("sweep pieces, edges mid-sweep: 3 loads only, no stores, no reduction", d, s);
  runm<3>("sweep pieces, edges mid-sweep: 3 stores only, no loads, no reduction", d, s);
  runm<4>("sweep pieces, edges mid-sweep: loads + stores, no reduction", d, s);
  runm<6>("sweep pieces, edges mid-sweep: loads + stores, no reduction, no mid-sweep wait", d, s);
  runm<4, true>("sweep pieces, edges mid-sweep: loads + stores, no reduction, NO barrier (racy, timing only)", d, s);
  runm<1, true>("sweep pieces, edges mid-sweep: every store mid-sweep + reduction, NO barrier (racy, timing only)", d, s);
  runm<5>("sweep pieces, edges mid-sweep: b64 halo store/load + b128 partials, no reduction", d, s);
  runm<7>("sweep pieces, edges mid-sweep: ONE b32 load only", d, s);
  runm<8>("sweep pieces, edges mid-sweep: ONE b32 store only", d, s);
  runm<9>("sweep pieces, edges mid-sweep: one b32 load + one b32 store", d, s);
  runx<0, 0>("x: 16 cells (wave-wide DPP) + barrier", d, s);
  runx<0, 1>("x: 16 cells (wave-wide DPP) + ONE ds_read_b32 issued first + barrier", d, s);
  runx<0, 2>("x: 16 cells (wave-wide DPP) + ONE global_load_dword issued first + barrier", d, s);
  runx<0, 3>("x: 16 cells + ONE ds_read_b32 issued first and waited for at once", d, s);
  runx<0, 4>("x: 16 cells + ONE ds_read_b32 issued after 8 cells", d, s);
  runx<0, 5>("x: 16 cells + ONE ds_read_b32 issued after 16 cells (latency exposed)", d, s);
  runx<0, 6>("x: 16 cells + ONE ds_read_b32 issued first, waited for after 8 cells", d, s);
  runx<0, 7>("x: 16 cells + ONE ds_read_b32 issued first by wave 0 only", d, s);
  runx<0, 10>("x: 16 cells one by one + ds_read_b32 issued after 0 cells", d, s);
  runx<0, 11>("x: 16 cells one by one + ds_read_b32 issued after 1 cell", d, s);
  runx<0, 12>("x: 16 cells one by one + ds_read_b32 issued after 2 cells", d, s);
  runx<0, 14>("x: 16 cells one by one + ds_read_b32 issued after 4 cells", d, s);
  runx<0, 18>("x: 16 cells one by one + ds_read_b32 issued after 8 cells", d, s);
  runx<0, 22>("x: 16 cells one by one + ds_read_b32 after 2 cells (waves 0-3) / 10 cells (waves 4-7)", d, s);
  runx<0, 24>("x: 16 cells one by one + ds_read_b32 after 4 cells (waves 0-3) / 12 cells (waves 4-7)", d, s);
  runx<0, 0, true>("x: 16 cells, NO barrier", d, s);
  runx<0, 1, true>("x: 16 cells + ONE ds_read_b32 issued first + wait, NO barrier", d, s);
  runx<0, 5, true>("x: 16 cells + ONE ds_read_b32 issued last + wait (exposed), NO barrier", d, s);
  runx<0, 4, true>("x: 16 cells + ONE ds_read_b32 after 8 cells + wait at the end, NO barrier", d, s);
  runx<1, 0>("x: 16 cells (row DPP) + barrier", d, s);
  runx<1, 1>("x: 16 cells (row DPP) + ONE ds_read_b32 + barrier", d, s);
  runx<2, 0>("x: 16 cells (no DPP, plain adds) + barrier", d, s);
  runx<2, 1>("x: 16 cells (no DPP, plain adds) + ONE ds_read_b32 + barrier", d, s);
  return 0;
}
