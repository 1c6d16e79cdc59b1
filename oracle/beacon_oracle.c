/*
 * beacon_oracle.c -- CPU restatement (float64, scalar C) of the solver hot path of
 * jviquerat/beacon.  TEST INFRASTRUCTURE ONLY: it is the checker the HIP path is
 * compared against (tests/, __graft_entry__.smoke()) and the "port" CPU baseline
 * leg of bench.py.  Nothing in the product package beacon_amd/ may call it.
 *
 * Parity status: PINNED.  Every function below is checked in tests/test_oracle.py
 * against golden vectors captured from the unmodified reference source
 * (oracle/capture/capture.py -> tests/golden/ .npz files).
 *
 * Arrays keep the reference's own layout here: 2D fields are [nx+2][ny+2] C-order
 * (y fastest), f[i][j] == f[i*(ny+2)+j]; loops keep the reference's order so that
 * the float64 results are the reference's, operation for operation.  Citations are
 * file:line into /root/reference/beacon/.
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* 2D incompressible Navier-Stokes fractional step: rayleigh and mixing       */
/* ------------------------------------------------------------------------- */
typedef struct {
  int32_t kind;      /* 0 = rayleigh (rayleigh/rayleigh.py), 1 = mixing (mixing/mixing.py) */
  int32_t nx, ny;
  int32_t ndt_act;
  int32_t n_sgts, nx_sgts;             /* rayleigh bottom-plate segments */
  int32_t nx_obs_pts, ny_obs_pts, nx_obs, ny_obs, n_obs_steps;
  int32_t itmax;                       /* 300000 */
  double dx, dy, dt;
  double pr, ra, Tc, Th, C;            /* rayleigh */
  double re, pe, u_max, ref_c;         /* mixing */
  double tol;                          /* 1e-8 rayleigh, 1e-4 mixing */
} orc_ns2d_cfg;

#define F(a, i, j) (a)[(size_t)(i) * (size_t)(ny + 2) + (size_t)(j)]

/* rayleigh.py:180-202 / mixing.py:153-171 */
ORC_API void orc_ns2d_bc(const orc_ns2d_cfg* c, double* u, double* v, double* S, const double* a,
                         double u_t, double u_b, double v_l, double v_r) {
  const int nx = c->nx, ny = c->ny;
  int i, j, k;
  if (c->kind == 0) {
    for (j = 1; j <= ny; j++) F(u, 1, j) = 0.0;
    for (j = 2; j <= ny; j++) F(v, 0, j) = -F(v, 1, j);
    for (j = 1; j <= ny; j++) F(S, 0, j) = F(S, 1, j);
    for (j = 1; j <= ny; j++) F(u, nx + 1, j) = 0.0;
    for (j = 2; j <= ny; j++) F(v, nx + 1, j) = -F(v, nx, j);
    for (j = 1; j <= ny; j++) F(S, nx + 1, j) = F(S, nx, j);
    for (i = 1; i <= nx + 1; i++) F(u, i, ny + 1) = -F(u, i, ny);
    for (i = 1; i <= nx; i++) F(v, i, ny + 1) = 0.0;
    for (i = 1; i <= nx; i++) F(S, i, ny + 1) = 2.0 * c->Tc - F(S, i, ny);
    for (i = 1; i <= nx + 1; i++) F(u, i, 0) = -F(u, i, 1);
    for (i = 1; i <= nx; i++) F(v, i, 1) = 0.0;
    for (k = 0; k < c->n_sgts; k++) {
      int s = 1 + k * c->nx_sgts, e = 1 + (k + 1) * c->nx_sgts;
      for (i = s; i < e; i++) F(S, i, 0) = 2.0 * (c->Th + a[k]) - F(S, i, 1);
    }
  } else {
    for (j = 1; j <= ny; j++) F(u, 1, j) = 0.0;
    for (j = 2; j <= ny; j++) F(v, 0, j) = 2.0 * v_l - F(v, 1, j);
    for (j = 1; j <= ny; j++) F(S, 0, j) = F(S, 1, j);
    for (j = 1; j <= ny; j++) F(u, nx + 1, j) = 0.0;
    for (j = 2; j <= ny; j++) F(v, nx + 1, j) = 2.0 * v_r - F(v, nx, j);
    for (j = 1; j <= ny; j++) F(S, nx + 1, j) = F(S, nx, j);
    for (i = 1; i <= nx + 1; i++) F(u, i, ny + 1) = 2.0 * u_t - F(u, i, ny);
    for (i = 1; i <= nx; i++) F(v, i, ny + 1) = 0.0;
    for (i = 1; i <= nx; i++) F(S, i, ny + 1) = F(S, i, ny);
    for (i = 1; i <= nx + 1; i++) F(u, i, 0) = 2.0 * u_b - F(u, i, 1);
    for (i = 1; i <= nx; i++) F(v, i, 1) = 0.0;
    for (i = 1; i <= nx; i++) F(S, i, 0) = F(S, i, 1);
  }
}

/* rayleigh.py:370-407 / mixing.py:381-416 */
ORC_API void orc_ns2d_predictor(const orc_ns2d_cfg* c, const double* u, const double* v, double* us,
                                double* vs, const double* p, const double* T) {
  const int nx = c->nx, ny = c->ny;
  const double dx = c->dx, dy = c->dy, dt = c->dt;
  const double sq = (c->kind == 0) ? sqrt(c->pr / c->ra) : 0.0;
  int i, j;
  for (i = 2; i <= nx; i++)
    for (j = 1; j <= ny; j++) {
      double uE = 0.5 * (F(u, i + 1, j) + F(u, i, j));
      double uW = 0.5 * (F(u, i, j) + F(u, i - 1, j));
      double uN = 0.5 * (F(u, i, j + 1) + F(u, i, j));
      double uS = 0.5 * (F(u, i, j) + F(u, i, j - 1));
      double vN = 0.5 * (F(v, i, j + 1) + F(v, i - 1, j + 1));
      double vS = 0.5 * (F(v, i, j) + F(v, i - 1, j));
      double conv = (uE * uE - uW * uW) / dx + (uN * vN - uS * vS) / dy;
      double diff = ((F(u, i + 1, j) - 2.0 * F(u, i, j) + F(u, i - 1, j)) / (dx * dx) +
                     (F(u, i, j + 1) - 2.0 * F(u, i, j) + F(u, i, j - 1)) / (dy * dy));
      if (c->kind == 0) diff *= sq; else diff = diff / c->re;
      double pres = (F(p, i, j) - F(p, i - 1, j)) / dx;
      F(us, i, j) = F(u, i, j) + dt * (diff - conv - pres);
    }
  for (i = 1; i <= nx; i++)
    for (j = 2; j <= ny; j++) {
      double vE = 0.5 * (F(v, i + 1, j) + F(v, i, j));
      double vW = 0.5 * (F(v, i, j) + F(v, i - 1, j));
      double uE = 0.5 * (F(u, i + 1, j) + F(u, i + 1, j - 1));
      double uW = 0.5 * (F(u, i, j) + F(u, i, j - 1));
      double vN = 0.5 * (F(v, i, j + 1) + F(v, i, j));
      double vS = 0.5 * (F(v, i, j) + F(v, i, j - 1));
      double conv = (uE * vE - uW * vW) / dx + (vN * vN - vS * vS) / dy;
      double diff = ((F(v, i + 1, j) - 2.0 * F(v, i, j) + F(v, i - 1, j)) / (dx * dx) +
                     (F(v, i, j + 1) - 2.0 * F(v, i, j) + F(v, i, j - 1)) / (dy * dy));
      if (c->kind == 0) diff *= sq; else diff = diff / c->re;
      double pres = (F(p, i, j) - F(p, i, j - 1)) / dy;
      if (c->kind == 0)
        F(vs, i, j) = F(v, i, j) + dt * (diff - conv - pres + F(T, i, j));
      else
        F(vs, i, j) = F(v, i, j) + dt * (diff - conv - pres);
    }
}

/* rayleigh.py:411-456 / mixing.py:420-465.  phin is caller scratch [(nx+2)*(ny+2)].
 * Returns itp; *ovf set when itp > itmax.  The reference's err = np.dot(dphi,dphi)
 * over the WHOLE array incl. ghosts; here a plain left-to-right sum (BLAS order is
 * not reproducible; differs in the last bits only). */
ORC_API int orc_ns2d_poisson(const orc_ns2d_cfg* c, const double* us, const double* vs, double* phi,
                             double* phin, int* ovf) {
  const int nx = c->nx, ny = c->ny;
  const double dx = c->dx, dy = c->dy, dt = c->dt;
  const size_t n = (size_t)(nx + 2) * (size_t)(ny + 2);
  double err = 1.0e10;
  int itp = 0, i, j;
  size_t k;
  *ovf = 0;
  memset(phi, 0, n * sizeof(double));
  memset(phin, 0, n * sizeof(double));
  while (err > c->tol) {
    memcpy(phin, phi, n * sizeof(double));
    for (i = 1; i <= nx; i++)
      for (j = 1; j <= ny; j++) {
        double b = ((F(us, i + 1, j) - F(us, i, j)) / dx + (F(vs, i, j + 1) - F(vs, i, j)) / dy) / dt;
        F(phi, i, j) = 0.5 * ((F(phin, i + 1, j) + F(phin, i - 1, j)) * dy * dy +
                              (F(phin, i, j + 1) + F(phin, i, j - 1)) * dx * dx - b * dx * dx * dy * dy) /
                       (dx * dx + dy * dy);
      }
    for (j = 1; j <= ny; j++) F(phi, 0, j) = F(phi, 1, j);
    for (j = 1; j <= ny; j++) F(phi, nx + 1, j) = F(phi, nx, j);
    if (c->kind == 0) {
      for (i = 1; i <= nx; i++) F(phi, i, ny + 1) = F(phi, i, ny);   /* rayleigh.py:441-442 */
    } else {
      for (i = 1; i <= nx; i++) F(phi, i, ny + 1) = 0.0;            /* mixing.py:450-451 */
    }
    for (i = 1; i <= nx; i++) F(phi, i, 0) = F(phi, i, 1);
    err = 0.0;
    for (k = 0; k < n; k++) {
      double d = phi[k] - phin[k];
      err += d * d;
    }
    itp += 1;
    if (itp > c->itmax) {
      *ovf = 1;
      break;
    }
  }
  return itp;
}

/* rayleigh.py:460-464 / mixing.py:469-473 */
ORC_API void orc_ns2d_corrector(const orc_ns2d_cfg* c, double* u, double* v, const double* us,
                                const double* vs, const double* phi) {
  const int nx = c->nx, ny = c->ny;
  int i, j;
  for (i = 2; i <= nx; i++)
    for (j = 1; j <= ny; j++)
      F(u, i, j) = F(us, i, j) - c->dt * (F(phi, i, j) - F(phi, i - 1, j)) / c->dx;
  for (i = 1; i <= nx; i++)
    for (j = 2; j <= ny; j++)
      F(v, i, j) = F(vs, i, j) - c->dt * (F(phi, i, j) - F(phi, i, j - 1)) / c->dy;
}

/* rayleigh.py:468-487 / mixing.py:477-495: IN-PLACE sequential sweep, i outer, j inner */
ORC_API void orc_ns2d_transport(const orc_ns2d_cfg* c, const double* u, const double* v, double* T) {
  const int nx = c->nx, ny = c->ny;
  const double dx = c->dx, dy = c->dy;
  const double sq = (c->kind == 0) ? sqrt(c->pr * c->ra) : 0.0;
  int i, j;
  for (i = 1; i <= nx; i++)
    for (j = 1; j <= ny; j++) {
      double uE = F(u, i + 1, j), uW = F(u, i, j), vN = F(v, i, j + 1), vS = F(v, i, j);
      double TE = 0.5 * (F(T, i + 1, j) + F(T, i, j));
      double TW = 0.5 * (F(T, i - 1, j) + F(T, i, j));
      double TN = 0.5 * (F(T, i, j + 1) + F(T, i, j));
      double TS = 0.5 * (F(T, i, j - 1) + F(T, i, j));
      double conv = (uE * TE - uW * TW) / dx + (vN * TN - vS * TS) / dy;
      double diff = ((F(T, i + 1, j) - 2.0 * F(T, i, j) + F(T, i - 1, j)) / (dx * dx) +
                     (F(T, i, j + 1) - 2.0 * F(T, i, j) + F(T, i, j - 1)) / (dy * dy));
      if (c->kind == 0) diff /= sq; else diff = diff / c->pe;
      F(T, i, j) += c->dt * (diff - conv);
    }
}

/* mixing.py:212-234 */
static void mixing_control(const orc_ns2d_cfg* c, int a, double* u_t, double* u_b, double* v_l, double* v_r) {
  *u_t = *u_b = *v_l = *v_r = 0.0;
  if (a == 0) { *u_b = c->u_max; *u_t = -c->u_max; }
  if (a == 1) { *u_b = -c->u_max; *u_t = c->u_max; }
  if (a == 2) { *v_r = c->u_max; *v_l = -c->u_max; }
  if (a == 3) { *v_r = -c->u_max; *v_l = c->u_max; }
}

/* rayleigh.py:162-171: zero-mean then scale into [-C, C]; np.mean = pairwise sum for n<8 is a plain sum */
ORC_API void orc_rayleigh_condition_action(const orc_ns2d_cfg* c, double* a) {
  int k, n = c->n_sgts;
  double s = 0.0, m = 1.0;
  /* numpy's add.reduce on a contiguous float64 vector of <= 8 elements is sequential; beyond
     that it is 8-way unrolled pairwise -- restated for n up to 128 */
  if (n < 8) {
    for (k = 0; k < n; k++) s += a[k];
  } else {
    double r[8];
    int i;
    for (k = 0; k < 8; k++) r[k] = a[k];
    for (i = 8; i < n - (n % 8); i += 8)
      for (k = 0; k < 8; k++) r[k] += a[i + k];
    s = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) s += a[i];
  }
  s = s / (double)n;
  for (k = 0; k < n; k++) a[k] = a[k] - s;
  for (k = 0; k < n; k++) {
    double t = fabs(a[k]) / c->C;
    if (t > m) m = t;
  }
  for (k = 0; k < n; k++) a[k] = a[k] / m;
}

/* rayleigh.py:243-262 / mixing.py:237-258: obs is [n_obs_steps][3][nx_obs_pts][ny_obs_pts] */
ORC_API void orc_ns2d_obs(const orc_ns2d_cfg* c, const double* u, const double* v, const double* S, double* obs) {
  const int ny = c->ny;
  const int per = 3 * c->nx_obs_pts * c->ny_obs_pts;
  int i, j, x, y;
  double* last = obs + (size_t)(c->n_obs_steps - 1) * per;
  memmove(obs, obs + per, (size_t)(c->n_obs_steps - 1) * per * sizeof(double));
  x = c->nx_obs / 2;
  for (i = 0; i < c->nx_obs_pts; i++) {
    y = c->ny_obs / 2;
    for (j = 0; j < c->ny_obs_pts; j++) {
      last[(0 * c->nx_obs_pts + i) * c->ny_obs_pts + j] = F(S, x, y);
      last[(1 * c->nx_obs_pts + i) * c->ny_obs_pts + j] = F(u, x, y);
      last[(2 * c->nx_obs_pts + i) * c->ny_obs_pts + j] = F(v, x, y);
      y += c->ny_obs;
    }
    x += c->nx_obs;
  }
}

/* rayleigh.py:265-275 (returns -nu) / mixing.py:261-264 (-mean|C-ref| over the whole array) */
ORC_API double orc_ns2d_rwd(const orc_ns2d_cfg* c, const double* S) {
  const int nx = c->nx, ny = c->ny;
  int i;
  if (c->kind == 0) {
    double nu = 0.0;
    for (i = 1; i <= nx; i++) {
      double dT = (F(S, i, 1) - c->Th) / (0.5 * c->dy);
      nu -= dT;
    }
    nu /= (double)nx;
    return -nu;
  } else {
    size_t n = (size_t)(nx + 2) * (size_t)(ny + 2), k;
    double s = 0.0;
    for (k = 0; k < n; k++) s += fabs(S[k] - c->ref_c);
    return -(s / (double)n);
  }
}

/* One action step = ndt_act timesteps (rayleigh.py:174-240 / mixing.py:147-209).
 * fields: u,v,p,S,us,vs,phi,phin each [(nx+2)(ny+2)] contiguous in `st` (8 arrays).
 * a: rayleigh -> normalised action vector (already conditioned); mixing -> a[0] = action id.
 * itp_out[ndt_act] receives the Jacobi sweep count of each timestep.  Returns 0, or 1 on overflow. */
ORC_API int orc_ns2d_solve(const orc_ns2d_cfg* c, double* st, const double* a, int32_t* itp_out) {
  const size_t n = (size_t)(c->nx + 2) * (size_t)(c->ny + 2);
  double *u = st, *v = st + n, *p = st + 2 * n, *S = st + 3 * n, *us = st + 4 * n, *vs = st + 5 * n,
         *phi = st + 6 * n, *phin = st + 7 * n;
  double u_t = 0, u_b = 0, v_l = 0, v_r = 0;
  int it, ovf = 0;
  size_t k;
  if (c->kind == 1) mixing_control(c, (int)a[0], &u_t, &u_b, &v_l, &v_r);
  for (it = 0; it < c->ndt_act; it++) {
    orc_ns2d_bc(c, u, v, S, a, u_t, u_b, v_l, v_r);
    orc_ns2d_predictor(c, u, v, us, vs, p, S);
    int itp = orc_ns2d_poisson(c, us, vs, phi, phin, &ovf);
    if (itp_out) itp_out[it] = itp;
    for (k = 0; k < n; k++) p[k] += phi[k];
    if (ovf) return 1;
    orc_ns2d_corrector(c, u, v, us, vs, phi);
    orc_ns2d_transport(c, u, v, S);
  }
  return 0;
}

/* Batched action step for the CPU baseline: B independent envs, OpenMP over envs.
 * st: [B][8][n]; a: [B][na]; obs: [B][n_obs_tot] history buffers; rwd: [B]; sweeps: [B] total sweeps */
ORC_API int orc_ns2d_step_batch(const orc_ns2d_cfg* c, int B, double* st, double* a, int na, double* obs,
                                double* rwd, int64_t* sweeps, int nthreads) {
  const size_t n = (size_t)(c->nx + 2) * (size_t)(c->ny + 2);
  const size_t nobs = (size_t)c->n_obs_steps * 3 * c->nx_obs_pts * c->ny_obs_pts;
  int b, bad = 0;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1) reduction(| : bad)
  for (b = 0; b < B; b++) {
    double* s = st + (size_t)b * 8 * n;
    int32_t* itp = (int32_t*)malloc(sizeof(int32_t) * (size_t)c->ndt_act);
    int t;
    int64_t tot = 0;
    if (c->kind == 0) orc_rayleigh_condition_action(c, a + (size_t)b * na);
    bad |= orc_ns2d_solve(c, s, a + (size_t)b * na, itp);
    for (t = 0; t < c->ndt_act; t++) tot += itp[t];
    if (sweeps) sweeps[b] = tot;
    orc_ns2d_obs(c, s, s + n, s + 3 * n, obs + (size_t)b * nobs);
    rwd[b] = orc_ns2d_rwd(c, s + 3 * n);
    free(itp);
  }
  return bad;
}

/* ------------------------------------------------------------------------- */
/* burgers (burgers/burgers.py)                                               */
/* ------------------------------------------------------------------------- */
typedef struct {
  int32_t nx, ndt_act, ctrl_pos, n_obs_pts;
  double dx, dt, amp, u_target;
} orc_burgers_cfg;

/* burgers.py:119-151 with kernels :230-255.  u,up,upp,du,rhs: [nx]; noise: the one
 * uniform(-sigma,sigma) draw of this action step; a: action (pre-amp). */
ORC_API void orc_burgers_solve(const orc_burgers_cfg* c, double* u, double* up, double* upp, double* du,
                               double* rhs, double a, double noise) {
  const int nx = c->nx;
  const double dx = c->dx, dt = c->dt;
  double* phi = (double*)calloc((size_t)nx, sizeof(double));
  int it, i;
  for (it = 0; it < c->ndt_act; it++) {
    memcpy(upp, up, sizeof(double) * (size_t)nx);
    memcpy(up, u, sizeof(double) * (size_t)nx);
    u[0] = c->u_target + noise;
    u[nx - 1] = u[nx - 2];
    /* derx :230-243 (phi[0] = phi[nx-1] = 0) */
    for (i = 1; i < nx - 1; i++) {
      double r = (u[i] - u[i - 1]) / (u[i + 1] - u[i] + 1.0e-8);
      phi[i] = (r + fabs(r)) / (1.0 + r);
    }
    for (i = 1; i < nx - 1; i++) {
      double fp = u[i] + 0.5 * phi[i] * (u[i + 1] - u[i]);
      double fm = u[i - 1] + 0.5 * phi[i - 1] * (u[i] - u[i - 1]);
      du[i] = (fp - fm) / dx;
    }
    for (i = 1; i < nx - 1; i++) rhs[i] = u[i] * du[i];               /* :252-255 */
    rhs[c->ctrl_pos] += a * c->amp;                                   /* :143 */
    for (i = 1; i < nx - 1; i++) u[i] = (4.0 * up[i] - upp[i] - 2.0 * dt * rhs[i]) / 3.0; /* :246-249 */
  }
  free(phi);
}

/* burgers.py:154-166 */
ORC_API double orc_burgers_obs_rwd(const orc_burgers_cfg* c, const double* u, double* obs) {
  int i;
  double s = 0.0;
  for (i = 0; i < c->n_obs_pts; i++) obs[i] = u[c->ctrl_pos - c->n_obs_pts + i];
  for (i = c->ctrl_pos; i < c->nx; i++) s += fabs(u[i] - c->u_target);
  return -s * c->dx;
}

/* ------------------------------------------------------------------------- */
/* shkadov (shkadov/shkadov.py)                                               */
/* ------------------------------------------------------------------------- */
typedef struct {
  int32_t nx, ndt_act, n_jets, jet_pos, jet_hw, jet_space, l_obs, l_rwd, n_obs, n_interp, obs_stride;
  double dx, dt, delta, jet_amp, eps;
} orc_shkadov_cfg;

/* shkadov.py:494-504 */
static void shk_d1tvd(const double* u, double* du, double* phi, int nx, double dx) {
  int i;
  phi[0] = 0.0;
  phi[nx - 1] = 0.0;
  for (i = 1; i < nx - 1; i++) {
    double r = (u[i] - u[i - 1]) / (u[i + 1] - u[i] + 1.0e-8);
    phi[i] = fmax(0.0, fmin(r, 1.0));
    /* np.minimum/np.maximum propagate NaN; fmin/fmax do not -- keep NaN like numpy */
    if (r != r) phi[i] = r;
  }
  for (i = 1; i < nx - 1; i++) {
    double d = u[i] + 0.5 * phi[i] * (u[i + 1] - u[i]);
    d -= u[i - 1] + 0.5 * phi[i - 1] * (u[i] - u[i - 1]);
    du[i] = d / dx;
  }
}

/* shkadov.py:188-236.  h,q,rhsh,rhsq: [nx] state; ua: target action [n_jets] (self.u after the
 * shift), upa: previous action (self.up); noise[ndt_act]: inlet draws. */
ORC_API void orc_shkadov_solve(const orc_shkadov_cfg* c, double* h, double* q, double* rhsh, double* rhsq,
                               const double* ua, const double* upa, const double* noise) {
  const int nx = c->nx;
  const double dx = c->dx, dt = c->dt;
  double* w = (double*)calloc((size_t)nx * 6, sizeof(double));
  double *q2h = w, *dq2h = w + nx, *dddh = w + 2 * nx, *rhshp = w + 3 * nx, *rhsqp = w + 4 * nx,
         *phi = w + 5 * nx;
  const double pp = 1.0 / (5.0 * c->delta);
  int it, i, j, k;
  for (it = 0; it < c->ndt_act; it++) {
    memcpy(rhshp, rhsh, sizeof(double) * (size_t)nx);
    memcpy(rhsqp, rhsq, sizeof(double) * (size_t)nx);
    h[0] = 1.0 + noise[it];
    q[0] = 1.0;
    h[nx - 1] = h[nx - 2];
    q[nx - 1] = q[nx - 2];
    shk_d1tvd(q, rhsh, phi, nx, dx);
    for (i = 0; i < nx; i++) q2h[i] = q[i] * q[i] / (h[i] + c->eps);
    shk_d1tvd(q2h, dq2h, phi, nx, dx);
    /* d3o2u :485-491 */
    for (i = 1; i < nx - 3; i++)
      dddh[i] = (-h[i + 3] + 6.0 * h[i + 2] - 12.0 * h[i + 1] + 10.0 * h[i] - 3.0 * h[i - 1]) /
                (2.0 * dx * dx * dx);
    dddh[nx - 3] = (h[nx - 1] - 3.0 * h[nx - 2] + 3.0 * h[nx - 3] - h[nx - 4]) / (dx * dx * dx);
    dddh[nx - 2] = (-h[nx - 4] + 3.0 * h[nx - 3] - 3.0 * h[nx - 2] + h[nx - 1]) / (dx * dx * dx);
    /* rhsq :507-512 */
    for (i = 1; i < nx - 1; i++)
      rhsq[i] = 1.2 * dq2h[i] - pp * (h[i] * (dddh[i] + 1.0) - q[i] / (h[i] * h[i] + c->eps));
    /* jets :223-232 */
    {
      double alpha = fmin((double)it / (double)c->n_interp, 1.0);
      for (j = 0; j < c->n_jets; j++) {
        double uj = (1.0 - alpha) * upa[j] + alpha * ua[j];
        int s = c->jet_pos + j * c->jet_space - c->jet_hw;
        int e = s + 2 * c->jet_hw;
        for (k = s; k <= e; k++) {
          double vv = (double)((k - s) * (e - k)) / (0.25 * (double)((e - s) * (e - s)));
          rhsq[k] += c->jet_amp * uj * vv;
        }
      }
    }
    /* adams :515-518 */
    for (i = 1; i < nx - 1; i++) h[i] += 0.5 * dt * (-3.0 * rhsh[i] + rhshp[i]);
    for (i = 1; i < nx - 1; i++) q[i] += 0.5 * dt * (-3.0 * rhsq[i] + rhsqp[i]);
  }
  free(w);
}

/* shkadov.py:239-264 and blow-up test :176; returns rwd, *blowup set */
ORC_API double orc_shkadov_obs_rwd(const orc_shkadov_cfg* c, const double* h, const double* q, double* obs,
                                   int* blowup) {
  int i, k;
  double rwd = 0.0;
  for (i = 0; i < c->n_jets; i++) {
    int s = c->jet_pos + i * c->jet_space - c->l_obs;
    for (k = 0; k < c->n_obs; k++) obs[i * c->n_obs + k] = q[s + k * c->obs_stride];
  }
  for (i = 0; i < c->n_jets; i++) {
    int s = c->jet_pos + i * c->jet_space;
    double acc = 0.0;
    for (k = 0; k < c->l_rwd; k++) {
      double d = h[s + k] - 1.0;
      acc += d * d;
    }
    rwd -= acc * c->dx;
  }
  rwd /= (double)(c->n_jets * c->l_rwd);
  *blowup = 0;
  for (i = 0; i < c->nx; i++)
    if (h[i] < -25.0 || h[i] > 25.0) *blowup = 1;
  return rwd;
}

/* ------------------------------------------------------------------------- */
/* sloshing (sloshing/sloshing.py)                                            */
/* ------------------------------------------------------------------------- */
typedef struct {
  int32_t nx, ndt_act, n_interp;
  double dx, dt, g, amp, alpha;
} orc_sloshing_cfg;

/* sloshing.py:168-224.  h,q,rhsh,rhsq: [nx+2]; ua / upa: current / previous action */
ORC_API void orc_sloshing_solve(const orc_sloshing_cfg* c, double* h, double* q, double* rhsh, double* rhsq,
                                double ua, double upa) {
  const int nx = c->nx;
  const double dx = c->dx, dt = c->dt, g = c->g;
  double* w = (double*)calloc((size_t)(nx + 2) * 5, sizeof(double));
  double *v = w, *qgh = w + (nx + 2), *cc = w + 2 * (nx + 2), *rhshp = w + 3 * (nx + 2),
         *rhsqp = w + 4 * (nx + 2);
  int it, i;
  for (it = 0; it < c->ndt_act; it++) {
    h[0] = h[1];
    q[0] = 0.0;
    h[nx + 1] = h[nx];
    q[nx + 1] = 0.0;
    for (i = 1; i <= nx; i++) { rhshp[i] = rhsh[i]; rhsqp[i] = rhsq[i]; }
    for (i = 0; i < nx + 2; i++) {
      v[i] = q[i] / h[i];
      qgh[i] = q[i] * q[i] / h[i] + 0.5 * g * (h[i] * h[i]);
    }
    for (i = 0; i <= nx; i++) {
      double cl = fabs(v[i]) + sqrt(g * h[i]), cr = fabs(v[i + 1]) + sqrt(g * h[i + 1]);
      cc[i] = (cl > cr || cl != cl) ? cl : cr;
    }
    for (i = 1; i <= nx; i++) {
      /* rusanov :322-325 */
      double fhg = 0.5 * (q[i - 1] + q[i]) - 0.5 * cc[i - 1] * (h[i] - h[i - 1]);
      double fhd = 0.5 * (q[i] + q[i + 1]) - 0.5 * cc[i] * (h[i + 1] - h[i]);
      double fqg = 0.5 * (qgh[i - 1] + qgh[i]) - 0.5 * cc[i - 1] * (q[i] - q[i - 1]);
      double fqd = 0.5 * (qgh[i] + qgh[i + 1]) - 0.5 * cc[i] * (q[i + 1] - q[i]);
      rhsh[i] = (fhd - fhg) / dx;
      rhsq[i] = (fqd - fqg) / dx;
    }
    {
      double alpha = fmin((double)it / (double)c->n_interp, 1.0);
      double uu = (1.0 - alpha) * upa + alpha * ua;
      for (i = 1; i <= nx; i++) rhsq[i] += uu * c->amp;
    }
    for (i = 1; i <= nx; i++) h[i] += 0.5 * dt * (-3.0 * rhsh[i] + rhshp[i]);
    for (i = 1; i <= nx; i++) q[i] += 0.5 * dt * (-3.0 * rhsq[i] + rhsqp[i]);
  }
  free(w);
}

/* sloshing.py:227-244 and blow-up test :156; obs = q[1:-1][::2] */
ORC_API double orc_sloshing_obs_rwd(const orc_sloshing_cfg* c, const double* h, const double* q, double ua,
                                    double* obs, int* blowup) {
  int i, k = 0;
  double s = 0.0;
  for (i = 1; i <= c->nx; i += 2) obs[k++] = q[i];
  for (i = 1; i <= c->nx; i++) {
    double d = h[i] - 1.0;
    s += d * d;
  }
  *blowup = 0;
  for (i = 0; i < c->nx + 2; i++)
    if (h[i] < -5.0 || h[i] > 2.0) *blowup = 1;
  return -(sqrt(s) * c->dx) - c->alpha * fabs(c->amp * ua);
}

/* ------------------------------------------------------------------------- */
/* lorenz (lorenz/lorenz.py) -- plumbing config, CPU only                     */
/* ------------------------------------------------------------------------- */
static const double LSRK_A[5] = {0.000000000000000, -0.417890474499852, -1.192151694642677,
                                 -1.697784692471528, -1.514183444257156};
static const double LSRK_B[5] = {0.149659021999229, 0.379210312999627, 0.822955029386982,
                                 0.699450455949122, 0.153057247968152};

/* lorenz.py:120-153 with lsrk4.update :293-297.  x,xk,fx: [3]; force = actions[u] in {-1,0,1} */
ORC_API void orc_lorenz_solve(double* x, double* xk, double* fx, double sigma, double rho, double beta,
                              double dt, int ndt_act, double force) {
  int it, j, i;
  for (it = 0; it < ndt_act; it++) {
    for (i = 0; i < 3; i++) xk[i] = x[i];
    for (j = 0; j < 5; j++) {
      fx[0] = sigma * (xk[1] - xk[0]);
      fx[1] = xk[0] * (rho - xk[2]) - xk[1];
      fx[2] = xk[0] * xk[1] - beta * xk[2];
      fx[1] += force;
      for (i = 0; i < 3; i++) {
        x[i] = LSRK_A[j] * x[i] + dt * fx[i];
        xk[i] += LSRK_B[j] * x[i];
      }
    }
    for (i = 0; i < 3; i++) x[i] = xk[i];
  }
}
