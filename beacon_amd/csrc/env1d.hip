// env1d.hip -- burgers / shkadov / sloshing action steps, one workgroup per replica.
//
// Each thread owns K consecutive cells of every state array in registers for the whole
// action step (ndt_act timesteps); per timestep the only traffic is one LDS halo exchange
// (double-buffered: ONE barrier per timestep) -- HBM is touched once on entry and once on
// exit.  Inlet noise is an explicit input (the reference draws it from numpy's global
// stream: burgers.py:127, shkadov.py:204).  Citations: file:line into
// /root/reference/beacon/.
#include <stdlib.h>

#include <type_traits>

#include "env1d.h"

namespace {

template <typename real>
__device__ __forceinline__ real np_clip01(real r) {
  // np.maximum(0, np.minimum(r, 1)): NaN propagates (shkadov.py:500)
  real t = (r > real(1)) ? real(1) : r;
  return (t < real(0)) ? real(0) : t;
}

template <typename real>
__device__ __forceinline__ real np_max(real a, real b) {
  return (a > b || a != a) ? a : b;
}

// float32 kernels: divisions use a * v_rcp_f32(b) (~1.5 ulp, operands are O(1): no denormal scaling
// needed) and 1/dx is a multiplication; IEEE division costs ~10 VALU instructions and made up 20-40 % of
// the step time (scripts/exp_1ddiv.sh).  One exception: burgers' BDF2 update keeps the true division by 3
// (a reciprocal constant is a systematic bias there).  float64 keeps the reference's divisions.
template <typename real> __device__ __forceinline__ real fdiv(real a, real b) { return a / b; }
template <> __device__ __forceinline__ float fdiv<float>(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
template <typename real> __device__ __forceinline__ real divc(real a, real c, real rc) { (void)rc; return a / c; }
template <> __device__ __forceinline__ float divc<float>(float a, float c, float rc) { (void)c; return a * rc; }
template <typename real> __device__ __forceinline__ real fsqrt(real a) { return sqrt(a); }
template <> __device__ __forceinline__ float fsqrt<float>(float a) { return __builtin_amdgcn_sqrtf(a); }

template <typename real, int NT>
__device__ __forceinline__ void finish(const Env1DArgs<real>& A, int b, real rwd, bool blow, real blow_rwd,
                                       bool blow_overrides_rwd) {
  if (threadIdx.x == 0) {
    const int stp = A.stp[b];
    uint8_t dn = (stp == A.n_act - 1) ? 1 : 0, tr = dn;
    if (blow) { dn = 1; tr = 0; if (blow_overrides_rwd) rwd = blow_rwd; }
    if (A.rwd_out) A.rwd_out[b] = rwd;
    if (A.done) A.done[b] = dn;
    if (A.trunc) A.trunc[b] = tr;
    if (A.status) A.status[b] = blow ? BCN_ST_BLOWUP : BCN_ST_OK;
    A.stp[b] = stp + 1;
  }
}

// =========================================================================================
// burgers (burgers.py:119-166, kernels :230-255)
// =========================================================================================
template <typename real, int NT>
__device__ real burgers_obs_rwd(const Env1DArgs<real>& A, int b, real* red) {
  const real* gu = A.f0 + (size_t)b * A.n;
  if (A.obs_out && threadIdx.x < A.n_obs_pts)
    A.obs_out[(size_t)b * A.n_obs + threadIdx.x] = gu[A.ctrl_pos - A.n_obs_pts + threadIdx.x];
  real loc = 0;
  for (int i = A.ctrl_pos + threadIdx.x; i < A.n; i += NT) loc += bcn_abs(gu[i] - A.u_target);
  return -block_sum<real, NT>(loc, red) * A.dx;
}

template <typename real, int K, int NT>
__global__ __launch_bounds__(NT) void burgers_step_k(Env1DArgs<real> A) {
  if (A.mask && !A.mask[blockIdx.x]) return;
  constexpr int NB = (2 * NT * K * sizeof(real) <= 65536) ? 2 : 1;
  __shared__ real lds[NB][NT * K + 1];   // + 1: a slot holding 0 for reads outside the array
  __shared__ real red[NT / BCN_WAVE];
  const int b = blockIdx.x, tid = threadIdx.x, i0 = tid * K, n = A.n;
  real* gu = A.f0 + (size_t)b * n;
  real* gup = A.f1 + (size_t)b * n;
  real* gupp = A.f2 + (size_t)b * n;
  real u[K], up[K], upp[K];
#pragma unroll
  for (int k = 0; k < K; k++) {
    const int c = i0 + k;
    u[k] = c < n ? gu[c] : real(0);
    up[k] = c < n ? gup[c] : real(0);
    upp[k] = c < n ? gupp[c] : real(0);
  }
  const real act = A.actions ? A.actions[b] : A.a_last[b];
  const real noise = A.noise ? A.noise[b] : real(0);
  if (tid == 0) A.a_last[b] = act;
  const real force = act * A.amp;
  // static per thread: halo indices with the outflow copy u[nx-1] = u[nx-2] (:138) resolved on read
  // (cells outside the array read a slot holding 0: they only feed limiter entries that are masked)
  constexpr int ZERO = NT * K;
  auto hidx = [&](int c) -> int { return (c < 0 || c >= n) ? ZERO : (c == n - 1 ? n - 2 : c); };
  const int xm2 = hidx(i0 - 2), xm1 = hidx(i0 - 1), xp0 = hidx(i0 + K);
  const bool has_last = (i0 <= n - 1) && (n - 1 < i0 + K);
  if (tid == 0) { lds[0][ZERO] = 0; lds[NB - 1][ZERO] = 0; }

  for (int it = 0; it < A.ndt_act; it++) {
    real* L = lds[it & (NB - 1)];
#pragma unroll
    for (int k = 0; k < K; k++) { upp[k] = up[k]; up[k] = u[k]; }
    if (tid == 0) u[0] = A.u_target + noise;          // burgers.py:137
#pragma unroll
    for (int k = 0; k < K; k++) L[i0 + k] = u[k];
    __syncthreads();
    real e[K + 3];  // cells i0-2 .. i0+K
    e[0] = L[xm2];
    e[1] = L[xm1];
    if (has_last) {
#pragma unroll
      for (int k = 0; k < K; k++)
        if (i0 + k == n - 1) u[k] = L[n - 2];
    }
#pragma unroll
    for (int k = 0; k < K; k++) e[k + 2] = u[k];
    e[K + 2] = L[xp0];
    real ph[K + 1];  // limiter at cells i0-1 .. i0+K-1 (zero at both ends, :232-236)
#pragma unroll
    for (int k = 0; k <= K; k++) {
      const int c = i0 - 1 + k;
      real r = fdiv<real>(e[k + 1] - e[k], e[k + 2] - e[k + 1] + real(1.0e-8));
      real f = fdiv<real>(r + bcn_abs(r), real(1) + r);
      ph[k] = (c <= 0 || c >= n - 1) ? real(0) : f;
    }
#pragma unroll
    for (int k = 0; k < K; k++) {
      const int c = i0 + k;
      real fp = e[k + 2] + real(0.5) * ph[k + 1] * (e[k + 3] - e[k + 2]);
      real fm = e[k + 1] + real(0.5) * ph[k] * (e[k + 2] - e[k + 1]);
      real du = divc<real>(fp - fm, A.dx, A.rdx);
      real rhs = e[k + 2] * du;
      if (c == A.ctrl_pos) rhs += force;              // :143
      // true division: float(1/3) is 3e-8 too large, a bias that compounds to 2e-4 over 12400 timesteps
      real un = (real(4) * up[k] - upp[k] - real(2) * A.dt * rhs) / real(3);  // :246-249
      if (c >= 1 && c <= n - 2) u[k] = un;
    }
    if (NB == 1) __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < K; k++) {
    const int c = i0 + k;
    if (c < n) { gu[c] = u[k]; gup[c] = up[k]; gupp[c] = upp[k]; }
  }
  __syncthreads();
  real rwd = burgers_obs_rwd<real, NT>(A, b, red);
  finish<real, NT>(A, b, rwd, false, real(0), false);
}

template <typename real, int NT>
__global__ __launch_bounds__(NT) void burgers_reset_k(Env1DArgs<real> A) {
  if (A.mask && !A.mask[blockIdx.x]) return;
  __shared__ real red[NT / BCN_WAVE];
  const int b = blockIdx.x, n = A.n;
  for (int c = threadIdx.x; c < n; c += NT) {
    A.f0[(size_t)b * n + c] = A.u_target;
    A.f1[(size_t)b * n + c] = A.u_target;
    A.f2[(size_t)b * n + c] = A.u_target;
  }
  if (threadIdx.x == 0) { A.a_last[b] = 0; A.stp[b] = 0; }
  __syncthreads();
  (void)burgers_obs_rwd<real, NT>(A, b, red);
}

// =========================================================================================
// shkadov (shkadov.py:188-264, kernels :484-518)
// =========================================================================================
template <typename real, int NT>
__device__ real shkadov_obs_rwd(const Env1DArgs<real>& A, int b, real* red, bool* blow) {
  const real* gh = A.f0 + (size_t)b * A.n;
  const real* gq = A.f1 + (size_t)b * A.n;
  if (A.obs_out)
    for (int k = threadIdx.x; k < A.n_obs; k += NT) {
      const int j = k / A.n_obs_jet, m = k - j * A.n_obs_jet;
      const int s = A.jet_pos + j * A.jet_space - A.l_obs;
      A.obs_out[(size_t)b * A.n_obs + k] = gq[s + m * A.obs_stride];
    }
  real loc = 0;
  for (int k = threadIdx.x; k < A.n_jets * A.l_rwd; k += NT) {
    const int j = k / A.l_rwd, m = k - j * A.l_rwd;
    const real d = gh[A.jet_pos + j * A.jet_space + m] - real(1);
    loc += d * d;
  }
  real bl = 0;
  for (int c = threadIdx.x; c < A.n; c += NT) {
    const real hv = gh[c];
    if (hv < -A.h_blow || hv > A.h_blow) bl = 1;
  }
  real tot = block_sum<real, NT>(loc, red);
  __syncthreads();
  *blow = block_sum<real, NT>(bl, red) > real(0);
  return -(tot * A.dx) / (real)(A.n_jets * A.l_rwd);
}

template <typename real, int K, int NT>
__global__ __launch_bounds__(NT) void shkadov_step_k(Env1DArgs<real> A) {
  if (A.mask && !A.mask[blockIdx.x]) return;
  constexpr int NB = (4 * NT * K * sizeof(real) <= 65536) ? 2 : 1;
  __shared__ real lh[NB][NT * K + 1];   // + 1: a slot holding 1 for reads outside the array
  __shared__ real lq[NB][NT * K + 1];
  __shared__ real red[NT / BCN_WAVE];
  __shared__ real s_u[64], s_up[64];
  const int b = blockIdx.x, tid = threadIdx.x, i0 = tid * K, n = A.n;
  real* gh = A.f0 + (size_t)b * n;
  real* gq = A.f1 + (size_t)b * n;
  real* grh = A.f2 + (size_t)b * n;
  real* grq = A.f3 + (size_t)b * n;
  real h[K], q[K], rh[K], rq[K];
#pragma unroll
  for (int k = 0; k < K; k++) {
    const int c = i0 + k;
    h[k] = c < n ? gh[c] : real(1);
    q[k] = c < n ? gq[c] : real(1);
    rh[k] = c < n ? grh[c] : real(0);
    rq[k] = c < n ? grq[c] : real(0);
  }
  // action shift: up <- u, u <- new (shkadov.py:193-194)
  if (tid < A.n_jets) {
    const real uo = A.a_last[(size_t)b * A.n_jets + tid];
    const real un = A.actions ? A.actions[(size_t)b * A.n_jets + tid] : uo;
    s_u[tid] = un;
    s_up[tid] = uo;
    A.a_last[(size_t)b * A.n_jets + tid] = un;
    A.a_prev[(size_t)b * A.n_jets + tid] = uo;
  }
  const real* nz = A.noise ? A.noise + (size_t)b * A.ndt_act : nullptr;
  const real rdx3 = real(1) / (A.dx * A.dx * A.dx);
  __syncthreads();   // s_u / s_up
  // ---- everything that does not change over the action step is resolved once ----------------
  // halo cells i0-2, i0-1, i0+K, i0+K+1, i0+K+2: LDS index with the outflow copy BC
  // h[nx-1] = h[nx-2], q[nx-1] = q[nx-2] (shkadov.py:206-207) resolved on read; cells outside the
  // array read a slot that holds 1 (never used by a cell that is updated)
  constexpr int ONE = NT * K;
  auto hidx = [&](int c) -> int { return (c < 0 || c >= n) ? ONE : (c == n - 1 ? n - 2 : c); };
  const int xm2 = hidx(i0 - 2), xm1 = hidx(i0 - 1), xp0 = hidx(i0 + K), xp1 = hidx(i0 + K + 1), xp2 = hidx(i0 + K + 2);
  const bool has_last = (i0 <= n - 1) && (n - 1 < i0 + K);   // this thread owns cell nx-1
  const bool has_tail = (i0 + K > n - 3) && (i0 <= n - 2);   // ... cells nx-3 / nx-2 (one-sided d3o2u)
  // jets (:223-232): parabolic profile on [s, e], s = jet_pos + j*space - hw, e = s + 2 hw
  real jvv[K], ju0[K], ju1[K];
  bool jon[K];
#pragma unroll
  for (int k = 0; k < K; k++) {
    const int rel = i0 + k - (A.jet_pos - A.jet_hw);
    const int j = rel >= 0 ? rel / A.jet_space : 0;
    const int ks = rel - j * A.jet_space;              // k - s
    jon[k] = rel >= 0 && j < A.n_jets && ks <= 2 * A.jet_hw;
    jvv[k] = (real)(ks * (2 * A.jet_hw - ks)) / (real(0.25) * (real)(4 * A.jet_hw * A.jet_hw));
    ju0[k] = jon[k] ? s_up[j] : real(0);
    ju1[k] = jon[k] ? s_u[j] : real(0);
  }
  if (tid == 0) { lh[0][ONE] = 1; lq[0][ONE] = 1; lh[NB - 1][ONE] = 1; lq[NB - 1][ONE] = 1; }

  // Waves whose cells are all away from both ends of the array (all but the first and the last one on the
  // reference grids) run a body without edge cases; waves without jet cells skip the forcing.  Both flags are
  // wave-uniform, every variant executes the one barrier of the timestep.
  const bool t_int = (i0 >= 2) && (i0 + K + 2 <= n - 2);
  bool t_jet = false;
#pragma unroll
  for (int k = 0; k < K; k++) t_jet = t_jet || jon[k];
  const bool w_int = __builtin_amdgcn_ballot_w64(!t_int) == 0;
  const bool w_jet = __builtin_amdgcn_ballot_w64(t_jet) != 0;
  auto step = [&](const int it, auto int_tag, auto jet_tag) {
    constexpr bool INT = decltype(int_tag)::value, JET = decltype(jet_tag)::value;
    real* Lh = lh[it & (NB - 1)];
    real* Lq = lq[it & (NB - 1)];
    if (!INT && tid == 0) {                           // inlet BC (:204-205)
      h[0] = real(1) + (nz ? nz[it] : real(0));
      q[0] = real(1);
    }
#pragma unroll
    for (int k = 0; k < K; k++) { Lh[i0 + k] = h[k]; Lq[i0 + k] = q[k]; }
    __syncthreads();
    real eh[K + 5];  // cells i0-2 .. i0+K+2
    real eq[K + 3];  // cells i0-2 .. i0+K
    eh[0] = Lh[xm2]; eh[1] = Lh[xm1];
    eq[0] = Lq[xm2]; eq[1] = Lq[xm1];
    if (!INT && has_last) {
#pragma unroll
      for (int k = 0; k < K; k++)
        if (i0 + k == n - 1) { h[k] = Lh[n - 2]; q[k] = Lq[n - 2]; }
    }
#pragma unroll
    for (int k = 0; k < K; k++) {
      eh[k + 2] = h[k];
      eq[k + 2] = q[k];
    }
    eh[K + 2] = Lh[xp0]; eh[K + 3] = Lh[xp1]; eh[K + 4] = Lh[xp2];
    eq[K + 2] = Lq[xp0];
    real e2[K + 3];  // q^2/(h+eps) at cells i0-2 .. i0+K (:213)
#pragma unroll
    for (int k = 0; k < K + 3; k++) e2[k] = fdiv<real>(eq[k] * eq[k], eh[k] + A.eps);
    // minmod limiter at cells i0-1 .. i0+K-1 (zero at both array ends, :497-500)
    real pq[K + 1], p2[K + 1];
#pragma unroll
    for (int k = 0; k <= K; k++) {
      const int c = i0 - 1 + k;
      const bool edge = !INT && (c <= 0 || c >= n - 1);
      real r1 = fdiv<real>(eq[k + 1] - eq[k], eq[k + 2] - eq[k + 1] + real(1.0e-8));
      real r2 = fdiv<real>(e2[k + 1] - e2[k], e2[k + 2] - e2[k + 1] + real(1.0e-8));
      pq[k] = edge ? real(0) : np_clip01(r1);
      p2[k] = edge ? real(0) : np_clip01(r2);
    }
    // d3o2u(h) (:485-491); eh index of cell c is k+2; the last two interior cells are one-sided
    real d3[K];
#pragma unroll
    for (int k = 0; k < K; k++)
      d3[k] = (-eh[k + 5] + real(6) * eh[k + 4] - real(12) * eh[k + 3] + real(10) * eh[k + 2] -
               real(3) * eh[k + 1]) * (real(0.5) * rdx3);
    if (!INT && has_tail) {
#pragma unroll
      for (int k = 0; k < K; k++) {
        const int c = i0 + k;
        if (c == n - 3) d3[k] = (eh[k + 4] - real(3) * eh[k + 3] + real(3) * eh[k + 2] - eh[k + 1]) * rdx3;
        if (c == n - 2) d3[k] = (-eh[k] + real(3) * eh[k + 1] - real(3) * eh[k + 2] + eh[k + 3]) * rdx3;
      }
    }
    const real alpha = fmin((real)it / (real)A.n_interp, real(1));
#pragma unroll
    for (int k = 0; k < K; k++) {
      const int c = i0 + k;
      const real rhp = rh[k], rqp = rq[k];            // rhs of the previous timestep (:200-201)
      // d1tvd(q) -> rhsh, d1tvd(q2h) -> dq2h (:494-504)
      real dq = eq[k + 2] + real(0.5) * pq[k + 1] * (eq[k + 3] - eq[k + 2]);
      dq -= eq[k + 1] + real(0.5) * pq[k] * (eq[k + 2] - eq[k + 1]);
      dq = divc<real>(dq, A.dx, A.rdx);
      real d2 = e2[k + 2] + real(0.5) * p2[k + 1] * (e2[k + 3] - e2[k + 2]);
      d2 -= e2[k + 1] + real(0.5) * p2[k] * (e2[k + 2] - e2[k + 1]);
      d2 = divc<real>(d2, A.dx, A.rdx);
      // rhsq (:507-512)
      real rqn = real(1.2) * d2 - A.delta_p * (h[k] * (d3[k] + real(1)) - fdiv<real>(q[k], h[k] * h[k] + A.eps));
      if (JET) {
        const real uj = (real(1) - alpha) * ju0[k] + alpha * ju1[k];
        const real rqj = rqn + A.jet_amp * uj * jvv[k];
        rqn = jon[k] ? rqj : rqn;
      }
      if (INT || (c >= 1 && c <= n - 2)) {
        rh[k] = dq;
        rq[k] = rqn;
        h[k] += real(0.5) * A.dt * (real(-3) * dq + rhp);    // adams (:515-518)
        q[k] += real(0.5) * A.dt * (real(-3) * rqn + rqp);
      }
    }
    if (NB == 1) __syncthreads();
  };
  using T_ = std::true_type;
  using F_ = std::false_type;
  for (int it = 0; it < A.ndt_act; it++) {
    if (w_int) { if (w_jet) step(it, T_{}, T_{}); else step(it, T_{}, F_{}); }
    else { if (w_jet) step(it, F_{}, T_{}); else step(it, F_{}, F_{}); }
  }
#pragma unroll
  for (int k = 0; k < K; k++) {
    const int c = i0 + k;
    if (c < n) { gh[c] = h[k]; gq[c] = q[k]; grh[c] = rh[k]; grq[c] = rq[k]; }
  }
  __syncthreads();
  bool blow;
  real rwd = shkadov_obs_rwd<real, NT>(A, b, red, &blow);
  finish<real, NT>(A, b, rwd, blow, A.blowup_rwd, true);   // shkadov.py:176-180
}

template <typename real, int NT>
__global__ __launch_bounds__(NT) void shkadov_reset_k(Env1DArgs<real> A) {
  if (A.mask && !A.mask[blockIdx.x]) return;
  __shared__ real red[NT / BCN_WAVE];
  const int b = blockIdx.x, n = A.n;
  for (int c = threadIdx.x; c < n; c += NT) {
    A.f0[(size_t)b * n + c] = A.init_fields ? A.init_fields[c] : real(1);
    A.f1[(size_t)b * n + c] = A.init_fields ? A.init_fields[n + c] : real(1);
    A.f2[(size_t)b * n + c] = 0;
    A.f3[(size_t)b * n + c] = 0;
  }
  if (threadIdx.x < A.n_jets) {
    A.a_last[(size_t)b * A.n_jets + threadIdx.x] = 0;
    A.a_prev[(size_t)b * A.n_jets + threadIdx.x] = 0;
  }
  if (threadIdx.x == 0) A.stp[b] = 0;
  __syncthreads();
  bool blow;
  (void)shkadov_obs_rwd<real, NT>(A, b, red, &blow);
}

// =========================================================================================
// sloshing (sloshing.py:168-244, kernels :322-331); arrays have n = nx+2 entries
// =========================================================================================
template <typename real, int NT>
__device__ real sloshing_obs_rwd(const Env1DArgs<real>& A, int b, real ua, real* red, bool* blow) {
  const real* gh = A.f0 + (size_t)b * A.n;
  const real* gq = A.f1 + (size_t)b * A.n;
  if (A.obs_out)
    for (int k = threadIdx.x; k < A.n_obs; k += NT) A.obs_out[(size_t)b * A.n_obs + k] = gq[1 + 2 * k];
  real loc = 0, bl = 0;
  for (int c = threadIdx.x; c < A.n; c += NT) {
    const real hv = gh[c];
    if (c >= 1 && c <= A.nx) { const real d = hv - real(1); loc += d * d; }
    if (hv < real(-5) || hv > real(2)) bl = 1;     // sloshing.py:156
  }
  real tot = block_sum<real, NT>(loc, red);
  __syncthreads();
  *blow = block_sum<real, NT>(bl, red) > real(0);
  return -(sqrt(tot) * A.dx) - A.alpha * bcn_abs(A.amp * ua);
}

template <typename real, int K, int NT>
__global__ __launch_bounds__(NT) void sloshing_step_k(Env1DArgs<real> A) {
  if (A.mask && !A.mask[blockIdx.x]) return;
  constexpr int NB = (4 * NT * K * sizeof(real) <= 65536) ? 2 : 1;
  __shared__ real lh[NB][NT * K + 1];   // + 1: a slot holding 1 (h) / 0 (q) for reads outside the array
  __shared__ real lq[NB][NT * K + 1];
  __shared__ real red[NT / BCN_WAVE];
  const int b = blockIdx.x, tid = threadIdx.x, i0 = tid * K, n = A.n, nx = A.nx;
  real* gh = A.f0 + (size_t)b * n;
  real* gq = A.f1 + (size_t)b * n;
  real* grh = A.f2 + (size_t)b * n;
  real* grq = A.f3 + (size_t)b * n;
  real h[K], q[K], rh[K], rq[K];
#pragma unroll
  for (int k = 0; k < K; k++) {
    const int c = i0 + k;
    h[k] = c < n ? gh[c] : real(1);
    q[k] = c < n ? gq[c] : real(0);
    rh[k] = c < n ? grh[c] : real(0);
    rq[k] = c < n ? grq[c] : real(0);
  }
  const real uo = A.a_last[b];
  const real un = A.actions ? A.actions[b] : uo;
  __syncthreads();
  if (tid == 0) { A.a_last[b] = un; A.a_prev[b] = uo; }
  const real g = A.g;
  // static per thread: halo indices with the wall BCs h[0]=h[1], q[0]=0, h[nx+1]=h[nx], q[nx+1]=0 (:184-187)
  // resolved on read
  constexpr int SLOT = NT * K;
  auto hh = [&](int c) -> int { return (c < 0 || c >= n) ? SLOT : (c == 0 ? 1 : (c == n - 1 ? n - 2 : c)); };
  auto hq = [&](int c) -> int { return (c <= 0 || c >= n - 1) ? SLOT : c; };
  const int xhm = hh(i0 - 1), xhp = hh(i0 + K), xqm = hq(i0 - 1), xqp = hq(i0 + K);
  const bool has_wall = (i0 == 0) || ((i0 <= n - 1) && (n - 1 < i0 + K));
  if (tid == 0) { lh[0][SLOT] = 1; lh[NB - 1][SLOT] = 1; lq[0][SLOT] = 0; lq[NB - 1][SLOT] = 0; }

  for (int it = 0; it < A.ndt_act; it++) {
    real* Lh = lh[it & (NB - 1)];
    real* Lq = lq[it & (NB - 1)];
#pragma unroll
    for (int k = 0; k < K; k++) { Lh[i0 + k] = h[k]; Lq[i0 + k] = q[k]; }
    __syncthreads();
    real eh[K + 2], eq[K + 2];  // cells i0-1 .. i0+K
    eh[0] = Lh[xhm]; eq[0] = Lq[xqm];
    if (has_wall) {
#pragma unroll
      for (int k = 0; k < K; k++) {
        const int c = i0 + k;
        if (c == 0 || c == n - 1) { h[k] = Lh[c == 0 ? 1 : n - 2]; q[k] = real(0); }
      }
    }
#pragma unroll
    for (int k = 0; k < K; k++) {
      eh[k + 1] = h[k];
      eq[k + 1] = q[k];
    }
    eh[K + 1] = Lh[xhp]; eq[K + 1] = Lq[xqp];
    real ev[K + 2], eg[K + 2], es[K + 2];  // v, q^2/h + g h^2/2, |v| + sqrt(g h) (:193-199)
#pragma unroll
    for (int k = 0; k < K + 2; k++) {
      ev[k] = fdiv<real>(eq[k], eh[k]);
      eg[k] = fdiv<real>(eq[k] * eq[k], eh[k]) + real(0.5) * g * (eh[k] * eh[k]);
      es[k] = bcn_abs(ev[k]) + fsqrt<real>(g * eh[k]);
    }
    const real alpha = fmin((real)it / (real)A.n_interp, real(1));
    const real uu = (real(1) - alpha) * uo + alpha * un;
#pragma unroll
    for (int k = 0; k < K; k++) {
      const int c = i0 + k;
      const real cl = np_max(es[k], es[k + 1]);        // c[c-1]
      const real cr = np_max(es[k + 1], es[k + 2]);    // c[c]
      // rusanov (:322-325)
      real fhg = real(0.5) * (eq[k] + eq[k + 1]) - real(0.5) * cl * (eh[k + 1] - eh[k]);
      real fhd = real(0.5) * (eq[k + 1] + eq[k + 2]) - real(0.5) * cr * (eh[k + 2] - eh[k + 1]);
      real fqg = real(0.5) * (eg[k] + eg[k + 1]) - real(0.5) * cl * (eq[k + 1] - eq[k]);
      real fqd = real(0.5) * (eg[k + 1] + eg[k + 2]) - real(0.5) * cr * (eq[k + 2] - eq[k + 1]);
      real rhn = divc<real>(fhd - fhg, A.dx, A.rdx);
      real rqn = divc<real>(fqd - fqg, A.dx, A.rdx) + uu * A.amp;      // :213-218
      if (c >= 1 && c <= nx) {
        const real rhp = rh[k], rqp = rq[k];
        rh[k] = rhn;
        rq[k] = rqn;
        h[k] += real(0.5) * A.dt * (real(-3) * rhn + rhp);  // adams (:328-331)
        q[k] += real(0.5) * A.dt * (real(-3) * rqn + rqp);
      }
    }
    if (NB == 1) __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < K; k++) {
    const int c = i0 + k;
    if (c < n) { gh[c] = h[k]; gq[c] = q[k]; grh[c] = rh[k]; grq[c] = rq[k]; }
  }
  __syncthreads();
  bool blow;
  real rwd = sloshing_obs_rwd<real, NT>(A, b, un, red, &blow);
  finish<real, NT>(A, b, rwd, blow, real(0), false);  // the -10 blow-up reward is dead code (:156-160)
}

template <typename real, int NT>
__global__ __launch_bounds__(NT) void sloshing_reset_k(Env1DArgs<real> A) {
  if (A.mask && !A.mask[blockIdx.x]) return;
  __shared__ real red[NT / BCN_WAVE];
  const int b = blockIdx.x, n = A.n;
  for (int c = threadIdx.x; c < n; c += NT) {
    A.f0[(size_t)b * n + c] = A.init_fields ? A.init_fields[c] : real(1);
    A.f1[(size_t)b * n + c] = A.init_fields ? A.init_fields[n + c] : real(0);
    A.f2[(size_t)b * n + c] = 0;
    A.f3[(size_t)b * n + c] = 0;
  }
  if (threadIdx.x == 0) { A.a_last[b] = 0; A.a_prev[b] = 0; A.stp[b] = 0; }
  __syncthreads();
  bool blow;
  (void)sloshing_obs_rwd<real, NT>(A, b, real(0), red, &blow);
}

// ---- (K, NT) selection: NT*K >= n ------------------------------------------------------------
// K cells per thread, NT threads per replica.  Fewer cells per thread = more waves per replica: the
// per-timestep work of a thread is a dependent chain (halo exchange -> stencil -> update), so small
// batches of short grids want K small (more waves per SIMD to hide it), long grids want K = 8 (less
// halo traffic per cell).  pick_k(): 4 cells per thread up to 4096 cells (8 beyond), 2 when that leaves
// the chip with fewer than ~8 waves per CU; BCN_1D_K overrides (tests / tuning).
inline int pick_k(int n, int batch) {
  static int force = -1, ncu = 256;
  if (force < 0) {
    const char* e = getenv("BCN_1D_K");
    force = e ? atoi(e) : 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  }
  if (force == 1 || force == 2 || force == 4 || force == 8) return (n <= force * 1024) ? force : 8;
  // measured (scripts/exp_1dk.sh, B=1024): burgers N=512 K=8/4/2/1 -> 0.121/0.099/0.096/0.103 ms,
  // sloshing N=202 0.148/0.081/0.079/0.090 ms, shkadov N=4096 K=8/4 -> 1.146/0.937 ms
  int k = (n > 4096) ? 8 : 4;
  if (k == 4 && n <= 2048 && (long)batch * ((n + 255) / 256) < 8L * ncu) k = 2;
  return k;
}

#define BCN_LAUNCH_NT(KERNEL, K_, A, BATCH, STREAM)                                                             \
  do {                                                                                                          \
    const int n__ = (A).n;                                                                                      \
    if (n__ <= K_ * 64) hipLaunchKernelGGL((KERNEL<real, K_, 64>), dim3(BATCH), dim3(64), 0, STREAM, A);        \
    else if (n__ <= K_ * 128) hipLaunchKernelGGL((KERNEL<real, K_, 128>), dim3(BATCH), dim3(128), 0, STREAM, A); \
    else if (n__ <= K_ * 256) hipLaunchKernelGGL((KERNEL<real, K_, 256>), dim3(BATCH), dim3(256), 0, STREAM, A); \
    else if (n__ <= K_ * 512) hipLaunchKernelGGL((KERNEL<real, K_, 512>), dim3(BATCH), dim3(512), 0, STREAM, A); \
    else hipLaunchKernelGGL((KERNEL<real, K_, 1024>), dim3(BATCH), dim3(1024), 0, STREAM, A);                   \
  } while (0)

#define BCN_DISPATCH_1D(KERNEL, A, BATCH, STREAM)                                              \
  do {                                                                                         \
    if ((A).n > 8 * 1024) {                                                                    \
      bcn_set_error("1D grid of %d cells exceeds the 8192-cell kernel limit", (A).n);          \
      return BCN_ERR_UNSUPPORTED;                                                              \
    }                                                                                          \
    int k__ = pick_k((A).n, BATCH);                                                            \
    while ((A).n > k__ * 1024) k__ *= 2;                                                       \
    switch (k__) {                                                                             \
      case 1: BCN_LAUNCH_NT(KERNEL, 1, A, BATCH, STREAM); break;                               \
      case 2: BCN_LAUNCH_NT(KERNEL, 2, A, BATCH, STREAM); break;                               \
      case 4: BCN_LAUNCH_NT(KERNEL, 4, A, BATCH, STREAM); break;                               \
      default: BCN_LAUNCH_NT(KERNEL, 8, A, BATCH, STREAM); break;                              \
    }                                                                                          \
    BCN_HIP(hipGetLastError());                                                                \
  } while (0)

}  // namespace

template <typename real> int burgers_launch_step(const Env1DArgs<real>& a, int batch, hipStream_t s) {
  BCN_DISPATCH_1D(burgers_step_k, a, batch, s);
  return BCN_OK;
}
template <typename real> int burgers_launch_reset(const Env1DArgs<real>& a, int batch, hipStream_t s) {
  hipLaunchKernelGGL((burgers_reset_k<real, 64>), dim3(batch), dim3(64), 0, s, a);
  BCN_HIP(hipGetLastError());
  return BCN_OK;
}
template <typename real> int shkadov_launch_step(const Env1DArgs<real>& a, int batch, hipStream_t s) {
  if (a.n_jets > 64) { bcn_set_error("n_jets > 64 unsupported"); return BCN_ERR_UNSUPPORTED; }
  BCN_DISPATCH_1D(shkadov_step_k, a, batch, s);
  return BCN_OK;
}
template <typename real> int shkadov_launch_reset(const Env1DArgs<real>& a, int batch, hipStream_t s) {
  hipLaunchKernelGGL((shkadov_reset_k<real, 256>), dim3(batch), dim3(256), 0, s, a);
  BCN_HIP(hipGetLastError());
  return BCN_OK;
}
template <typename real> int sloshing_launch_step(const Env1DArgs<real>& a, int batch, hipStream_t s) {
  BCN_DISPATCH_1D(sloshing_step_k, a, batch, s);
  return BCN_OK;
}
template <typename real> int sloshing_launch_reset(const Env1DArgs<real>& a, int batch, hipStream_t s) {
  hipLaunchKernelGGL((sloshing_reset_k<real, 64>), dim3(batch), dim3(64), 0, s, a);
  BCN_HIP(hipGetLastError());
  return BCN_OK;
}

#define BCN_INST(T)                                                                   \
  template int burgers_launch_step<T>(const Env1DArgs<T>&, int, hipStream_t);         \
  template int burgers_launch_reset<T>(const Env1DArgs<T>&, int, hipStream_t);        \
  template int shkadov_launch_step<T>(const Env1DArgs<T>&, int, hipStream_t);         \
  template int shkadov_launch_reset<T>(const Env1DArgs<T>&, int, hipStream_t);        \
  template int sloshing_launch_step<T>(const Env1DArgs<T>&, int, hipStream_t);        \
  template int sloshing_launch_reset<T>(const Env1DArgs<T>&, int, hipStream_t);
BCN_INST(float)
BCN_INST(double)
