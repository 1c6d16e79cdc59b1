// env1d.h -- argument block of the 1D solver kernels (burgers, shkadov, sloshing;
// SURVEY.md 8a rows B1-B3, S1-S3, L1-L2).  Fields are [B][n] contiguous per field in HBM.
#pragma once
#include "bcn_common.h"

template <typename real>
struct Env1DArgs {
  int n;        // array length per replica (burgers/shkadov: nx, sloshing: nx+2)
  int nx;       // interior / nominal nx
  int ndt_act, n_act, n_obs;
  // burgers
  int ctrl_pos, n_obs_pts;
  real u_target, amp;
  // shkadov
  int n_jets, jet_pos, jet_hw, jet_space, l_obs, l_rwd, n_obs_jet, obs_stride, n_interp;
  real delta_p;  // 1/(5 delta)
  real jet_amp, eps, h_blow, blowup_rwd;
  // sloshing
  real g, alpha;
  // common numerics
  real dx, rdx, dt;
  // persistent state: f0..f3 (burgers: u,up,upp,-; shkadov/sloshing: h,q,rhsh,rhsq)
  real *f0, *f1, *f2, *f3;
  real* a_last;   // [B][n_actions] current action (self.a / self.u)
  real* a_prev;   // [B][n_actions] previous action (self.up) -- shkadov, sloshing
  int32_t* stp;
  // per-call I/O
  const real* actions;
  const real* noise;
  // inlet noise drawn on the device when `noise` is NULL (bcn_set_noise): uniform(-nsigma, nsigma) from Philox4x32-10 keyed by
  // the seed, counter = (global replica index, this replica's step counter nctr[b], timestep, 0); nsigma = 0: no noise
  real nsigma;
  uint32_t nseed_lo, nseed_hi;
  int noff;                 // global index of replica 0 (sharded batches)
  uint32_t* nctr;           // [B] steps taken with device noise (advanced by the kernel: a captured graph replays fresh noise)
  const real* init_fields;
  const uint8_t* mask;      // per-replica enable (NULL = all)
  real* obs_out;
  real* rwd_out;
  uint8_t* done;
  uint8_t* trunc;
  int32_t* status;
  // host side only (launchers), by bcn_set_option:
  int force_k = 0;          // "cells_per_thread": 1, 2, 4 or 8 cells per thread (0 = chosen from grid and batch: pick_k)
  int one_wave = 1;         // "one_wave": grids up to 512 cells as ONE wave per replica with DPP halos (0: the LDS-halo kernels)
};

// Philox4x32-10 (Salmon et al., SC'11): four 32-bit words per (counter, key)
__device__ __forceinline__ void bcn_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                               uint32_t (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// uniform(-sigma, sigma) of replica b (local index), draw counter ctr, timestep k
template <typename real>
__device__ __forceinline__ real bcn_device_noise(const Env1DArgs<real>& A, int b, uint32_t ctr, int k) {
  uint32_t o[4];
  bcn_philox4x32((uint32_t)(b + A.noff), ctr, (uint32_t)k, 0u, A.nseed_lo, A.nseed_hi, o);
  const real r = sizeof(real) == 8 ? (real)((((unsigned long long)o[0] << 21) ^ (unsigned long long)(o[1] >> 11)) & ((1ull << 53) - 1)) *
                                         (real)(1.0 / 9007199254740992.0)
                                   : (real)(o[0] >> 8) * (real)(1.0 / 16777216.0);
  return (real(2) * r - real(1)) * A.nsigma;
}

// name of the step kernel the last *_launch_step of this thread dispatched when it is not the env's general one (else nullptr)
extern thread_local const char* bcn_env1d_launched;
template <typename real> int burgers_launch_step(const Env1DArgs<real>& a, int batch, hipStream_t s);
template <typename real> int burgers_launch_reset(const Env1DArgs<real>& a, int batch, hipStream_t s);
template <typename real> int shkadov_launch_step(const Env1DArgs<real>& a, int batch, hipStream_t s);
template <typename real> int shkadov_launch_reset(const Env1DArgs<real>& a, int batch, hipStream_t s);
template <typename real> int sloshing_launch_step(const Env1DArgs<real>& a, int batch, hipStream_t s);
template <typename real> int sloshing_launch_reset(const Env1DArgs<real>& a, int batch, hipStream_t s);
