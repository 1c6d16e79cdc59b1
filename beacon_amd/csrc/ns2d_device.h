// ns2d_device.h -- device code shared by the generic and the register-resident rayleigh /
// mixing kernels: observation history, reward, episode bookkeeping (global-memory state).
#pragma once
#include "ns2d.h"

// observation history shift + sample + copy out (rayleigh.py:243-262 / mixing.py:237-258).
// hist: [n_obs_steps][3][nxo][nyo]; the reference's probe (x,y) are ARRAY indices incl. ghosts.
template <typename real, int NT>
__device__ void ns2d_obs(const NS2DArgs<real>& A, int b, const real* u, const real* v, const real* S) {
  real* hist = A.obs_hist + (size_t)b * A.n_obs;
  const int per = 3 * A.nxo * A.nyo;
  for (int s = 0; s + 1 < A.n_obs_steps; s++) {
    for (int k = threadIdx.x; k < per; k += NT) hist[s * per + k] = hist[(s + 1) * per + k];
    __syncthreads();
  }
  real* last = hist + (A.n_obs_steps - 1) * per;
  for (int k = threadIdx.x; k < per; k += NT) {
    int f = k / (A.nxo * A.nyo);
    int r = k - f * (A.nxo * A.nyo);
    int io = r / A.nyo, jo = r - io * A.nyo;
    int x = A.nx_obs / 2 + io * A.nx_obs, y = A.ny_obs / 2 + jo * A.ny_obs;
    const real* src = (f == 0) ? S : (f == 1) ? u : v;
    last[k] = src[y * A.sx + x];
  }
  __syncthreads();
  if (A.obs_out)
    for (int k = threadIdx.x; k < A.n_obs; k += NT) A.obs_out[(size_t)b * A.n_obs + k] = hist[k];
}

// reward (rayleigh.py:265-275 / mixing.py:261-264), done/trunc (rayleigh.py:148-155), status.
// Called by all threads after the final state is in global memory and a barrier.
template <typename real, int NT>
__device__ void ns2d_finish(const NS2DArgs<real>& A, int b, const real* u, const real* v, const real* S,
                            int status, real* red) {
  const int tid = threadIdx.x;
  ns2d_obs<real, NT>(A, b, u, v, S);
  real loc = 0;
  if (A.kind == 0) {
    // returns -nu = (1/nx) sum_i (T[i,1]-Th)/(0.5 dy)
    for (int i = 1 + tid; i <= A.nx; i += NT) loc += (S[1 * A.sx + i] - A.Th);
    loc *= A.rwd_scale;
  } else {
    // -mean |C - ref| over the whole array incl. ghosts
    for (int c = tid; c < A.ncell; c += NT) loc += bcn_abs(S[c] - A.ref_c);
    loc = -loc / (real)A.ncell;
  }
  __syncthreads();
  real rwd = block_sum<real, NT>(loc, red);
  if (tid == 0) {
    const int stp = A.stp[b];
    const uint8_t dn = (stp == A.n_act - 1) ? 1 : 0;
    if (A.rwd_out) A.rwd_out[b] = rwd;
    if (A.done) A.done[b] = dn;
    if (A.trunc) A.trunc[b] = dn;
    if (A.status) A.status[b] = status;
    A.stp[b] = stp + 1;
  }
}
