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
  const real* init_fields;
  const uint8_t* mask;      // per-replica enable (NULL = all)
  real* obs_out;
  real* rwd_out;
  uint8_t* done;
  uint8_t* trunc;
  int32_t* status;
};

template <typename real> int burgers_launch_step(const Env1DArgs<real>& a, int batch, hipStream_t s);
template <typename real> int burgers_launch_reset(const Env1DArgs<real>& a, int batch, hipStream_t s);
template <typename real> int shkadov_launch_step(const Env1DArgs<real>& a, int batch, hipStream_t s);
template <typename real> int shkadov_launch_reset(const Env1DArgs<real>& a, int batch, hipStream_t s);
template <typename real> int sloshing_launch_step(const Env1DArgs<real>& a, int batch, hipStream_t s);
template <typename real> int sloshing_launch_reset(const Env1DArgs<real>& a, int batch, hipStream_t s);
