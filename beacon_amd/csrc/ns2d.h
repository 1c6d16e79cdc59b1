// ns2d.h -- argument block shared by the rayleigh / mixing kernels (2D incompressible
// Navier-Stokes fractional step on a MAC grid).
//
// HBM layout: every field is [B][ny+2][nx+2], x fastest (the transpose, per replica, of the
// reference's [nx+2][ny+2] arrays): F(i,j) = f[b*ncell + j*sx + i], ghosts at 0 and n+1.
#pragma once
#include "bcn_common.h"

// What only the launchers read and write, kept OUT of the kernel argument block: a field more in NS2DArgs shifts the kernarg
// offsets, and hipcc's schedule of the register-saturated kernels reacts to that (round 5: 8 bytes more = 18 761 changed lines
// of ISA in ns2d_fast.hip).
struct NS2DHost {
  const char* launched = nullptr;   // name of the kernel the last step dispatched
  int sched_tail = 0;               // bcn_set_option "sched_tail": short chunks that end a step of the ticket scheduler (0 = default 6)
  int generic_nt = 0;               // bcn_set_option "generic_threads": threads per workgroup of the generic kernel, 256 / 1024 (0 = by grid size)
};

template <typename real>
struct NS2DArgs {
  int nx, ny, sx, ncell;
  int ndt_act, n_act, itmax;
  int kind;  // 0 rayleigh, 1 mixing
  int n_sgts, nx_sgts;
  int nxo, nyo, nx_obs, ny_obs, n_obs_steps, n_obs;
  int i_min, i_max, j_min, j_max;  // mixing patch
  real dt, rdx, rdy, rdx2, rdy2;
  real kmom;  // sqrt(pr/ra)   | 1/re
  real ksc;   // 1/sqrt(pr*ra) | 1/pe
  real cx, cy, cb;  // Jacobi: phi = cx (E+W) + cy (N+S) - cb * div(us,vs)
  real tol;
  real Tc, Th, C, u_max, ref_c, C0;
  real rwd_scale;  // rayleigh: 1 / (0.5 dy nx)
  // persistent device state
  real *u, *v, *p, *S, *us, *vs;
  real *g0, *g1, *g2;  // per-replica global work arrays (used when they do not fit LDS)
  real* obs_hist;      // [B][n_obs]
  real* a_last;        // rayleigh [B][n_sgts]
  int32_t* ia_last;    // mixing [B]
  int32_t* stp;
  // per-call I/O
  const real* actions;
  const int32_t* iactions;
  real* actions_norm;
  real* obs_out;
  real* rwd_out;
  uint8_t* done;
  uint8_t* trunc;
  int32_t* status;
  int32_t* sweeps;
  const real* init_fields;  // reset only
  const uint8_t* mask;      // per-replica enable (NULL = all)
  int work_in_lds;
  // fast path only: timestep range of this launch and replica order (LPT scheduling, see ns2d_fast.hip)
  int it_begin, it_end;
  int first_chunk, last_chunk;
  const int32_t* order;     // blockIdx -> replica, or NULL for identity
  int32_t* order_out;       // rank kernel output
  int32_t* sweeps_int;      // handle-owned [B][ndt_act] when the caller passes no sweeps buffer
  void* sched_ctl;          // handle-owned control block of the ticketed chunk scheduler (64 + 4B bytes), followed by
  unsigned long long* cyc;  // [B][4] of the last step: shader-clock cycles inside the Jacobi loop / in the whole replica, late stops, repeated timesteps (bcn_get_counters)
  size_t sched_bytes;       // bytes of sched_ctl + cyc: zeroed by one memset in front of every step launch
  int sched_q;              // timesteps per chunk
  int sched_nbig = 0;       // the first sched_nbig chunks of a step are 2 sched_q timesteps long (ns2d_sched.h)
  int conv_plan = 1;        // which Jacobi sweeps evaluate the residual: 0 all, 1 proven skips only, 2 + extrapolated, 3 = 2 with unverified stops repeated under 1 (ns2d_fast_impl.h)
  int plan_overshoot = 0;   // TEST HOOK: sweeps added to every skip of the extrapolating plan (provokes late stops: tests of conv_plan 3)
  int spec_start = 0;       // first evaluation of a solve at spec_start/8 of the previous timestep's sweep count (0: at sweep 1): ns2d_fast_impl.h
  int verify_conv = 0;      // 1: evaluate the Jacobi residual after every sweep and flag BCN_ST_PLAN if the evaluation plan
                            //    of the register-resident kernels would have skipped a sweep that passes the test
  int sched_mode = -1;      // scheduling of the register-resident kernels (bcn_set_sched): -1 / 0 = default
  int sched_grid = 0, sched_q_user = 0, lpt_min_batch = 0;
  NS2DHost* host;           // host side only (never read by a kernel; may be NULL): see NS2DHost
  real* fscr;               // per-workgroup field scratch of the register-resident kernels whose u, v, T do not
  size_t fscr_stride;       //   fit LDS (float64 128x64): [slots][fscr_stride] elements, slot = workgroup index
  // slow-mode landing guard of conv_plan 3 (ns2d_fast_impl.h, scripts/weighted_norm_bound.py): log2 of two eigenvalue cutoffs of the
  // Jacobi matrix and, per cutoff, the bound C_L >= 1 on the growth of the reference norm over any number of sweeps within the span
  // of the modes above it -- properties of the grid (nx, ny, kind, cx), set by the host (bcn_set_slow_mode_bound); +inf: none
  int transport_iter = 0;   // mixing, float32, two-rows-per-lane kernel: the ordered part of the scalar transport as at most this many
                            // parallel passes (ns2d_fast2_impl.h; bcn_set_option "transport_iter"); 0 = the reference's ordered sweep
  float slow_l2lc[2] = {0.f, 0.f};
  float slow_cl[2] = {__builtin_inff(), __builtin_inff()};
};

// launchers (one per translation unit)
template <typename real> int ns2d_launch_generic(const NS2DArgs<real>& a, int batch, hipStream_t s);
template <typename real> int ns2d_launch_reset(const NS2DArgs<real>& a, int batch, hipStream_t s);
size_t ns2d_generic_lds_bytes(int ncell, size_t esz);
// register-resident CDNA4 path (ns2d_fast.hip); returns BCN_ERR_UNSUPPORTED when the grid has none
template <typename real> bool ns2d_fast_supported(const NS2DArgs<real>& a);
template <typename real> int ns2d_launch_fast(const NS2DArgs<real>& a, int batch, hipStream_t s);
// two-rows-per-lane variant for 64 < ny <= 128 (ns2d_fast2.hip), reached through ns2d_launch_fast
// elements of field scratch one workgroup of the fast path needs for this configuration (0: fields live in LDS)
template <typename real> size_t ns2d_fast_scratch_elems(const NS2DArgs<real>& a);
template <typename real> bool ns2d_fast2_supported(const NS2DArgs<real>& a);
template <typename real> size_t ns2d_fast2_scratch_elems(const NS2DArgs<real>& a);
template <typename real> int ns2d_launch_fast2(const NS2DArgs<real>& a, int batch, hipStream_t s);
