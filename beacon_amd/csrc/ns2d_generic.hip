// ns2d_generic.hip -- grid-size-agnostic rayleigh / mixing action step (variant 0).
//
// One workgroup per replica runs the WHOLE action step (ndt_act timesteps, then obs + reward)
// in one launch: BC -> predictor -> Jacobi pressure-Poisson -> p += phi, corrector -> scalar
// transport.  u,v,p,S,us,vs stay in HBM/L2 (x-fastest rows, lanes along x: coalesced); the
// three work arrays (phi ping-pong + Poisson rhs, later the transport coefficients) live in
// LDS when 3*(nx+2)(ny+2) reals fit 160 KB, else in per-replica global scratch.
//
// Reference semantics kept (file:line into /root/reference/beacon/):
//  * Jacobi stop test err = sum over the WHOLE array incl. ghosts of (phi-phin)^2, checked
//    every sweep (rayleigh.py:448-449): ghosts copy their interior neighbour, so an edge
//    cell's increment is counted once more per Neumann side -> weight w(i,j).
//  * ghosts of phin are never materialised: neighbour indices are clamped (Neumann) or
//    read as 0 (mixing's top Dirichlet, mixing.py:450-451).
//  * transport is the reference's IN-PLACE sweep (rayleigh.py:468-487): cell (i,j) sees the
//    NEW values of (i-1,j) and (i,j-1).  It is linear in those two, so it is evaluated as
//    S' = A + aW*S'(i-1,j) + aS*S'(i,j-1) along anti-diagonals d = i+j (cells of one diagonal
//    are independent); A, aW, aS are computed for all cells in parallel first.
#include <stdlib.h>

#include "ns2d.h"
#include "ns2d_device.h"

namespace {


template <typename real, int NT, bool LDSW>
__global__ __launch_bounds__(NT) void ns2d_generic_step(NS2DArgs<real> A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NW = NT / BCN_WAVE;
  const int b = blockIdx.x;
  if (A.mask && !A.mask[b]) return;
  const int tid = threadIdx.x;
  const int tx = tid & (BCN_WAVE - 1), ty = tid >> 6;
  const int nx = A.nx, ny = A.ny, sx = A.sx;
  const size_t off = (size_t)b * A.ncell;

  real* red = reinterpret_cast<real*>(smem);  // [2][NW]
  real* sact = red + 2 * NW;                  // [64] conditioned actions
  real* lds_work = sact + 64;
  real *W0, *W1, *W2;
  if constexpr (LDSW) {
    W0 = lds_work; W1 = W0 + A.ncell; W2 = W1 + A.ncell;
  } else {
    W0 = A.g0 + off; W1 = A.g1 + off; W2 = A.g2 + off;
  }
  real* __restrict__ u = A.u + off;
  real* __restrict__ v = A.v + off;
  real* __restrict__ p = A.p + off;
  real* __restrict__ S = A.S + off;
  real* __restrict__ us = A.us + off;
  real* __restrict__ vs = A.vs + off;

  // ---- action conditioning (rayleigh.py:162-171) / wall speeds (mixing.py:212-234) ----
  real u_t = 0, u_b = 0, v_l = 0, v_r = 0;
  if (A.kind == 0) {
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
    int act = A.iactions ? A.iactions[b] : A.ia_last[b];
    __syncthreads();
    if (tid == 0) A.ia_last[b] = act;
    if (act == 0) { u_b = A.u_max; u_t = -A.u_max; }
    if (act == 1) { u_b = -A.u_max; u_t = A.u_max; }
    if (act == 2) { v_r = A.u_max; v_l = -A.u_max; }
    if (act == 3) { v_r = -A.u_max; v_l = A.u_max; }
  }
  __syncthreads();

  int status = 0;
  for (int it = 0; it < A.ndt_act && status == 0; it++) {
    // ---- boundary conditions (rayleigh.py:180-202 / mixing.py:153-171) ----
    for (int j = 1 + tid; j <= ny; j += NT) {
      u[j * sx + 1] = 0;
      u[j * sx + nx + 1] = 0;
      if (j >= 2) {
        v[j * sx + 0] = 2 * v_l - v[j * sx + 1];
        v[j * sx + nx + 1] = 2 * v_r - v[j * sx + nx];
      }
      S[j * sx + 0] = S[j * sx + 1];
      S[j * sx + nx + 1] = S[j * sx + nx];
    }
    for (int i = 1 + tid; i <= nx + 1; i += NT) {
      // u[1,.] and u[nx+1,.] are zeroed by the loop above in the same phase: use the value
      // they will have (the reference sets the side walls first)
      const bool wall = (i == 1) || (i == nx + 1);
      real utop = wall ? real(0) : u[ny * sx + i];
      real ubot = wall ? real(0) : u[1 * sx + i];
      u[(ny + 1) * sx + i] = 2 * u_t - utop;
      u[0 * sx + i] = 2 * u_b - ubot;
      if (i <= nx) {
        v[(ny + 1) * sx + i] = 0;
        v[1 * sx + i] = 0;
        if (A.kind == 0) {
          S[(ny + 1) * sx + i] = 2 * A.Tc - S[ny * sx + i];
          int k = (i - 1) / A.nx_sgts;
          if (k < A.n_sgts) S[0 * sx + i] = 2 * (A.Th + sact[k]) - S[1 * sx + i];
        } else {
          S[(ny + 1) * sx + i] = S[ny * sx + i];
          S[0 * sx + i] = S[1 * sx + i];
        }
      }
    }
    __syncthreads();

    // ---- predictor (rayleigh.py:370-407 / mixing.py:381-416) ----
    for (int j = 1 + ty; j <= ny; j += NW)
      for (int i = 1 + tx; i <= nx; i += BCN_WAVE) {
        const int c = j * sx + i;
        const real uc = u[c], uE_ = u[c + 1], uW_ = u[c - 1], uN_ = u[c + sx], uS_ = u[c - sx];
        const real vc = v[c], vE_ = v[c + 1], vW_ = v[c - 1], vN_ = v[c + sx], vS_ = v[c - sx];
        const real pc = p[c];
        if (i >= 2) {
          real uE = real(0.5) * (uE_ + uc), uW = real(0.5) * (uc + uW_);
          real uN = real(0.5) * (uN_ + uc), uS = real(0.5) * (uc + uS_);
          real vN = real(0.5) * (vN_ + v[c + sx - 1]), vS = real(0.5) * (vc + vW_);
          real conv = (uE * uE - uW * uW) * A.rdx + (uN * vN - uS * vS) * A.rdy;
          real diff = ((uE_ - 2 * uc + uW_) * A.rdx2 + (uN_ - 2 * uc + uS_) * A.rdy2) * A.kmom;
          real pres = (pc - p[c - 1]) * A.rdx;
          us[c] = uc + A.dt * (diff - conv - pres);
        }
        if (j >= 2) {
          real vE = real(0.5) * (vE_ + vc), vW = real(0.5) * (vc + vW_);
          real uE = real(0.5) * (uE_ + u[c + 1 - sx]), uW = real(0.5) * (uc + uS_);
          real vN = real(0.5) * (vN_ + vc), vS = real(0.5) * (vc + vS_);
          real conv = (uE * vE - uW * vW) * A.rdx + (vN * vN - vS * vS) * A.rdy;
          real diff = ((vE_ - 2 * vc + vW_) * A.rdx2 + (vN_ - 2 * vc + vS_) * A.rdy2) * A.kmom;
          real pres = (pc - p[c - sx]) * A.rdy;
          real buoy = (A.kind == 0) ? S[c] : real(0);
          vs[c] = vc + A.dt * (diff - conv - pres + buoy);
        }
      }
    __syncthreads();

    // ---- Poisson rhs (recomputed every sweep in the reference, rayleigh.py:424-426) ----
    real* src = W0;
    real* dst = W1;
    int itp = 0;
    real err;
    // Every thread owns at most NA x NB cells (i = 1+tx+64a, j = 1+ty+NW b).  When the grid fits
    // these static slots, a cell's centre value, rhs and error weight stay in registers and the
    // ghost cells of the destination buffer are written by the owners of the edge cells (the
    // reference copies them after each sweep), so the inner loop is 4 LDS reads + 1 write with
    // no index clamping.  Larger grids take the clamped loop below.
    constexpr int NA = 2, NB = 8;
    constexpr bool REGS = NT <= 256;      // 1024 threads leave 128 VGPRs: slots in registers would spill
    if (LDSW && nx <= NA * BCN_WAVE && ny <= NB * NW) {
      real pc[REGS ? NB : 1][NA], rh[REGS ? NB : 1][NA];
      for (int c = tid; c < A.ncell; c += NT) { W0[c] = 0; W1[c] = 0; }
#pragma unroll
      for (int bb = 0; bb < NB; bb++)
#pragma unroll
        for (int aa = 0; aa < NA; aa++) {
          const int jj = 1 + ty + NW * bb, ii = 1 + tx + BCN_WAVE * aa;
          const bool valid = (jj <= ny) && (ii <= nx);
          const int c = valid ? jj * sx + ii : sx + 1;
          const real r = valid ? A.cb * ((us[c + 1] - us[c]) * A.rdx + (vs[c + sx] - vs[c]) * A.rdy) : real(0);
          if constexpr (REGS) { rh[bb][aa] = r; pc[bb][aa] = 0; }
          else if (valid) W2[c] = r;
        }
      // error weights (ghosts copy their interior neighbour): 1 + one per Neumann side the cell touches
      const int aE = (nx - 1) / BCN_WAVE;                              // slot column that holds i == nx
      const real wW = (tx == 0) ? real(1) : real(0);                   // i == 1  (slot column 0)
      const real wE = (tx == (nx - 1) % BCN_WAVE) ? real(1) : real(0); // i == nx (slot column aE)
      __syncthreads();
#ifdef BCN_STAMP
      unsigned long long gs0 = 0, gs1 = 0, gt = __builtin_amdgcn_s_memtime();
#endif
      do {
        real loc = 0;
#pragma unroll
        for (int bb = 0; bb < NB; bb++) {
          const int jj = 1 + ty + NW * bb;          // wave-uniform
          if (jj <= ny) {
            const real wrow = real(1) + (jj == 1 ? 1 : 0) + ((jj == ny && A.kind == 0) ? 1 : 0);
#pragma unroll
            for (int aa = 0; aa < NA; aa++) {
              const int ii = 1 + tx + BCN_WAVE * aa;
              if (ii <= nx) {
                const int c = jj * sx + ii;
                const real rhs = REGS ? rh[REGS ? bb : 0][aa] : W2[c];
                const real cen = REGS ? pc[REGS ? bb : 0][aa] : src[c];
                const real ph = A.cx * (src[c + 1] + src[c - 1]) + A.cy * (src[c + sx] + src[c - sx]) - rhs;
                const real d = ph - cen;
                const real w = wrow + (aa == 0 ? wW : real(0)) + (aa == aE ? wE : real(0));
                loc += w * d * d;
                if constexpr (REGS) pc[bb][aa] = ph;
                dst[c] = ph;
              }
            }
          }
        }
        // ghost cells of dst = copies of the adjacent new interior values (0 on mixing's top wall),
        // written by the threads that own the edge cells, outside the hot loop
        if (tx == 0 || tx == (nx - 1) % BCN_WAVE) {
#pragma unroll
          for (int bb = 0; bb < NB; bb++) {
            const int jj = 1 + ty + NW * bb;
            if (jj <= ny) {
              if (tx == 0) dst[jj * sx + 0] = REGS ? pc[REGS ? bb : 0][0] : dst[jj * sx + 1];
              if (tx == (nx - 1) % BCN_WAVE)
                dst[jj * sx + nx + 1] = REGS ? pc[REGS ? bb : 0][aE < NA ? aE : 0] : dst[jj * sx + nx];
            }
          }
        }
        if (ty == 0) {
#pragma unroll
          for (int aa = 0; aa < NA; aa++) {
            const int ii = 1 + tx + BCN_WAVE * aa;
            if (ii <= nx) dst[0 * sx + ii] = REGS ? pc[0][aa] : dst[1 * sx + ii];
          }
        }
        if (ty == (ny - 1) % NW) {
          const int bT = (ny - 1) / NW;                 // slot row that holds j == ny
#pragma unroll
          for (int aa = 0; aa < NA; aa++) {
            const int ii = 1 + tx + BCN_WAVE * aa;
            if (ii <= nx) {
              real top = 0;
              if (A.kind == 0) {
                if constexpr (REGS) {
#pragma unroll
                  for (int bb = 0; bb < NB; bb++) if (bb == bT) top = pc[bb][aa];
                } else {
                  top = dst[ny * sx + ii];
                }
              }
              dst[(ny + 1) * sx + ii] = top;
            }
          }
        }
#ifdef BCN_STAMP
        { __builtin_amdgcn_sched_barrier(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); gs0 += t_ - gt; gt = t_; __builtin_amdgcn_sched_barrier(0); }
#endif
        err = block_sum<real, NT>(loc, red + (itp & 1) * NW);
#ifdef BCN_STAMP
        { __builtin_amdgcn_sched_barrier(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); gs1 += t_ - gt; gt = t_; __builtin_amdgcn_sched_barrier(0); }
#endif
        itp++;
        real* t = src; src = dst; dst = t;
        if (itp > A.itmax) { status |= BCN_ST_ITMAX; break; }
      } while (err > A.tol);
#ifdef BCN_STAMP
      if (tid == 0 && A.sweeps) { A.sweeps[(size_t)b * A.ndt_act + it] = itp | ((int)(gs0 / itp / 16) << 12) | ((int)(gs1 / itp / 16) << 22); }
#endif
    } else {
      // rolled loops; ghosts of the destination buffer are materialised by the edge-cell owners
      for (int c = tid; c < A.ncell; c += NT) { W0[c] = 0; W1[c] = 0; }
      for (int j = 1 + ty; j <= ny; j += NW)
        for (int i = 1 + tx; i <= nx; i += BCN_WAVE) {
          const int c = j * sx + i;
          W2[c] = A.cb * ((us[c + 1] - us[c]) * A.rdx + (vs[c + sx] - vs[c]) * A.rdy);
        }
      __syncthreads();

      // ---- Jacobi sweeps (rayleigh.py:419-454 / mixing.py:428-463) ----
      do {
        real loc = 0;
        for (int j = 1 + ty; j <= ny; j += NW) {
          const bool top = (j == ny), bot = (j == 1);
          const real wrow = real(1) + (bot ? 1 : 0) + ((top && A.kind == 0) ? 1 : 0);
          for (int i = 1 + tx; i <= nx; i += BCN_WAVE) {
            const int c = j * sx + i;
            const real ph = A.cx * (src[c + 1] + src[c - 1]) + A.cy * (src[c + sx] + src[c - sx]) - W2[c];
            const real d = ph - src[c];
            const real w = wrow + (i == 1 ? 1 : 0) + (i == nx ? 1 : 0);
            loc += w * d * d;
            dst[c] = ph;
            if (i == 1) dst[c - 1] = ph;
            if (i == nx) dst[c + 1] = ph;
            if (bot) dst[c - sx] = ph;
            if (top) dst[c + sx] = (A.kind == 0) ? ph : real(0);
          }
        }
        err = block_sum<real, NT>(loc, red + (itp & 1) * NW);
        itp++;
        real* t = src; src = dst; dst = t;
        if (itp > A.itmax) { status |= BCN_ST_ITMAX; break; }
      } while (err > A.tol);
    }
    real* phi = src;  // converged field (interior); `dst` is free
#ifndef BCN_STAMP
    if (A.sweeps && tid == 0) A.sweeps[(size_t)b * A.ndt_act + it] = itp;
#endif

    // ---- p += phi incl. ghosts (rayleigh.py:219), corrector (rayleigh.py:460-464) ----
    for (int j = 1 + ty; j <= ny; j += NW)
      for (int i = 1 + tx; i <= nx; i += BCN_WAVE) {
        const int c = j * sx + i;
        const real ph = phi[c];
        p[c] += ph;
        if (i == 1) p[c - 1] += ph;
        if (i == nx) p[c + 1] += ph;
        if (j == 1) p[c - sx] += ph;
        if (j == ny && A.kind == 0) p[c + sx] += ph;
        if (i >= 2) u[c] = us[c] - A.dt * (ph - phi[c - 1]) * A.rdx;
        if (j >= 2) v[c] = vs[c] - A.dt * (ph - phi[c - sx]) * A.rdy;
      }
    __syncthreads();

    // ---- transport, parallel part: A -> X, aW -> Y, aS -> Z ----
    real* X = dst;
    real* Y = W2;
    real* Z = phi;
    for (int j = 1 + ty; j <= ny; j += NW)
      for (int i = 1 + tx; i <= nx; i += BCN_WAVE) {
        const int c = j * sx + i;
        const real uE = u[c + 1], uW = u[c], vN = v[c + sx], vS = v[c];
        const real Tc_ = S[c], TE = S[c + 1], TN = S[c + sx];
        real expl = A.ksc * ((TE - 2 * Tc_) * A.rdx2 + (TN - 2 * Tc_) * A.rdy2) -
                    (uE * real(0.5) * (TE + Tc_) - uW * real(0.5) * Tc_) * A.rdx -
                    (vN * real(0.5) * (TN + Tc_) - vS * real(0.5) * Tc_) * A.rdy;
        real aw = A.dt * (A.ksc * A.rdx2 + real(0.5) * uW * A.rdx);
        real as = A.dt * (A.ksc * A.rdy2 + real(0.5) * vS * A.rdy);
        // the explicit part is written after the barrier below (Z aliases phi, read above only)
        X[c] = Tc_ + A.dt * expl;
        Y[c] = aw;
        Z[c] = as;
        if (i == 1) X[c - 1] = S[c - 1];   // west ghost (old BC value)
        if (j == 1) X[c - sx] = S[c - sx]; // south ghost
      }
    __syncthreads();

    // ---- transport, ordered part: anti-diagonal wavefront ----
    {
      const int dmax = nx + ny;
      const int maxlen = nx < ny ? nx : ny;
      if (LDSW && maxlen <= 4 * BCN_WAVE) {
        // one wave walks all diagonals (up to 4 cells per lane); LDS operations of one wave complete
        // in issue order, so no workgroup barrier is needed between diagonals
        if (ty == 0) {
          for (int d = 2; d <= dmax; d++) {
            const int i0 = (d - ny > 1) ? d - ny : 1;
            const int i1 = (d - 1 < nx) ? d - 1 : nx;
            for (int i = i0 + tx; i <= i1; i += BCN_WAVE) {
              const int c = (d - i) * sx + i;
              X[c] = X[c] + Y[c] * X[c - 1] + Z[c] * X[c - sx];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
          }
        }
        __syncthreads();
      } else {
        for (int d = 2; d <= dmax; d++) {
          const int i0 = (d - ny > 1) ? d - ny : 1;
          const int i1 = (d - 1 < nx) ? d - 1 : nx;
          for (int i = i0 + tid; i <= i1; i += NT) {
            const int c = (d - i) * sx + i;
            X[c] = X[c] + Y[c] * X[c - 1] + Z[c] * X[c - sx];
          }
          __syncthreads();
        }
      }
    }
    for (int j = 1 + ty; j <= ny; j += NW)
      for (int i = 1 + tx; i <= nx; i += BCN_WAVE) S[j * sx + i] = X[j * sx + i];
    __syncthreads();
  }

  ns2d_finish<real, NT>(A, b, u, v, S, status, red);
}

// reset (rayleigh.py:89-128 / mixing.py:73-111)
template <typename real, int NT>
__global__ __launch_bounds__(NT) void ns2d_reset_kernel(NS2DArgs<real> A) {
  const int b = blockIdx.x;
  if (A.mask && !A.mask[b]) return;
  const size_t off = (size_t)b * A.ncell;
  for (int c = threadIdx.x; c < A.ncell; c += NT) {
    real u0 = 0, v0 = 0, p0 = 0, s0 = 0;
    if (A.kind == 0) {
      if (A.init_fields) {
        u0 = A.init_fields[c];
        v0 = A.init_fields[A.ncell + c];
        p0 = A.init_fields[2 * A.ncell + c];
        s0 = A.init_fields[3 * A.ncell + c];
      }
    } else {
      int j = c / A.sx, i = c - j * A.sx;
      if (i >= A.i_min && i < A.i_max && j >= A.j_min && j < A.j_max) s0 = A.C0;
    }
    A.u[off + c] = u0; A.v[off + c] = v0; A.p[off + c] = p0; A.S[off + c] = s0;
    A.us[off + c] = 0; A.vs[off + c] = 0;
  }
  for (int k = threadIdx.x; k < A.n_obs; k += NT) A.obs_hist[(size_t)b * A.n_obs + k] = 0;
  if (A.kind == 0) {
    for (int k = threadIdx.x; k < A.n_sgts; k += NT) A.a_last[(size_t)b * A.n_sgts + k] = 0;
  } else if (threadIdx.x == 0) {
    A.ia_last[b] = 1;  // mixing.py:101
  }
  if (threadIdx.x == 0) A.stp[b] = 0;
  __syncthreads();
  ns2d_obs<real, NT>(A, b, A.u + off, A.v + off, A.S + off);
}

constexpr size_t kLdsBudget = 160 * 1024;

template <typename real, int NT>
size_t fixed_lds() { return (2 * (NT / BCN_WAVE) + 64) * sizeof(real); }

template <typename real, int NT>
int launch_step(const NS2DArgs<real>& a_in, int batch, hipStream_t s) {
  NS2DArgs<real> a = a_in;
  size_t lds = fixed_lds<real, NT>();
  const size_t work = 3 * (size_t)a.ncell * sizeof(real);
  a.work_in_lds = (lds + work <= kLdsBudget) ? 1 : 0;
  if (a.work_in_lds) {
    lds += work;
    auto k = ns2d_generic_step<real, NT, true>;
    BCN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3(batch), dim3(NT), lds, s, a);
  } else {
    auto k = ns2d_generic_step<real, NT, false>;
    hipLaunchKernelGGL(k, dim3(batch), dim3(NT), lds, s, a);
  }
  BCN_HIP(hipGetLastError());
  if (a.host) a.host->launched = "ns2d_generic_step";
  return BCN_OK;
}

}  // namespace

size_t ns2d_generic_lds_bytes(int ncell, size_t esz) {
  size_t fixed = (2 * 16 + 64) * esz;
  size_t work = 3 * (size_t)ncell * esz;
  return (fixed + work <= kLdsBudget) ? fixed + work : fixed;
}

template <typename real>
int ns2d_launch_generic(const NS2DArgs<real>& a, int batch, hipStream_t s) {
  // 16 waves when there is enough work per replica to feed them, else 4 (bcn_set_option "generic_threads": 256 / 1024)
  const int force_nt = a.host ? a.host->generic_nt : 0;
  if (force_nt == 1024 || (force_nt == 0 && a.nx * a.ny >= 4096)) return launch_step<real, 1024>(a, batch, s);
  return launch_step<real, 256>(a, batch, s);
}

template <typename real>
int ns2d_launch_reset(const NS2DArgs<real>& a, int batch, hipStream_t s) {
  hipLaunchKernelGGL((ns2d_reset_kernel<real, 256>), dim3(batch), dim3(256), 0, s, a);
  BCN_HIP(hipGetLastError());
  return BCN_OK;
}

template int ns2d_launch_generic<float>(const NS2DArgs<float>&, int, hipStream_t);
template int ns2d_launch_generic<double>(const NS2DArgs<double>&, int, hipStream_t);
template int ns2d_launch_reset<float>(const NS2DArgs<float>&, int, hipStream_t);
template int ns2d_launch_reset<double>(const NS2DArgs<double>&, int, hipStream_t);
