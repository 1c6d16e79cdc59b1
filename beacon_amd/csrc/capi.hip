// capi.hip -- extern "C" surface of libbeacon_hip.so (include/beacon_hip.h): handle
// management, argument-block construction (derived constants are computed here in double and
// narrowed once), state copies.  All device work is enqueued on the caller's stream.
#include <math.h>
#include <stdarg.h>

#include <new>
#include <vector>

#include "env1d.h"
#include "ns2d.h"

thread_local const char* bcn_env1d_launched = nullptr;   // env1d.h: set by the 1D launchers

static thread_local char g_err[512] = "";

// default visibility: the on-demand kernel plugins (csrc/jit/ns2d_jit.hip) report through the library's error buffer
__attribute__((visibility("default"))) void bcn_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

namespace {

struct DeviceGuard {
  int prev = 0;
  explicit DeviceGuard(int dev) { (void)hipGetDevice(&prev); (void)hipSetDevice(dev); }
  ~DeviceGuard() { (void)hipSetDevice(prev); }
};

// ------------------------------------------------------------------------------------------
// rayleigh / mixing
// ------------------------------------------------------------------------------------------
template <typename real>
struct NS2DEnv : bcn_env_s {
  NS2DArgs<real> a{};
  DevBuf fields;    // u,v,p,S,us,vs   [6][B][ncell]
  DevBuf work;      // g0,g1,g2        [3][B][ncell]  (only when the work arrays do not fit LDS)
  DevBuf obs_hist, a_last, ia_last, stpbuf, sweepbuf, orderbuf, schedbuf, fscrbuf, statusbuf;
  int32_t* status_int = nullptr;   // per-replica status words when the caller passes no status_dev
  bool fast_ok = false;
  bool guard_holds() const { return a.nx >= 48 && a.ny >= 48; }

  int init() {
    const size_t per = (size_t)batch * a.ncell * sizeof(real);
    int rc;
    if ((rc = fields.alloc(6 * per))) return rc;
    BCN_HIP(hipMemset(fields.p, 0, 6 * per));
    real* f = static_cast<real*>(fields.p);
    const size_t n = (size_t)batch * a.ncell;
    a.u = f; a.v = f + n; a.p = f + 2 * n; a.S = f + 3 * n; a.us = f + 4 * n; a.vs = f + 5 * n;
    const bool in_lds = ns2d_generic_lds_bytes(a.ncell, sizeof(real)) > (2 * 16 + 64) * sizeof(real);
    if (!in_lds) {
      if ((rc = work.alloc(3 * per))) return rc;
      real* w = static_cast<real*>(work.p);
      a.g0 = w; a.g1 = w + n; a.g2 = w + 2 * n;
    }
    if ((rc = obs_hist.alloc((size_t)batch * a.n_obs * sizeof(real)))) return rc;
    BCN_HIP(hipMemset(obs_hist.p, 0, obs_hist.bytes));
    a.obs_hist = static_cast<real*>(obs_hist.p);
    if ((rc = a_last.alloc((size_t)batch * (a.n_sgts > 0 ? a.n_sgts : 1) * sizeof(real)))) return rc;
    BCN_HIP(hipMemset(a_last.p, 0, a_last.bytes));
    a.a_last = static_cast<real*>(a_last.p);
    if ((rc = ia_last.alloc((size_t)batch * sizeof(int32_t)))) return rc;
    BCN_HIP(hipMemset(ia_last.p, 0, ia_last.bytes));
    a.ia_last = static_cast<int32_t*>(ia_last.p);
    if ((rc = stpbuf.alloc((size_t)batch * sizeof(int32_t)))) return rc;
    BCN_HIP(hipMemset(stpbuf.p, 0, stpbuf.bytes));
    stp = a.stp = static_cast<int32_t*>(stpbuf.p);
    if ((rc = sweepbuf.alloc((size_t)batch * (a.ndt_act > 0 ? a.ndt_act : 1) * sizeof(int32_t)))) return rc;
    BCN_HIP(hipMemset(sweepbuf.p, 0, sweepbuf.bytes));
    a.sweeps_int = static_cast<int32_t*>(sweepbuf.p);
    if ((rc = statusbuf.alloc((size_t)batch * sizeof(int32_t)))) return rc;
    BCN_HIP(hipMemset(statusbuf.p, 0, statusbuf.bytes));
    status_int = static_cast<int32_t*>(statusbuf.p);
    if ((rc = orderbuf.alloc((size_t)batch * sizeof(int32_t)))) return rc;
    a.order_out = static_cast<int32_t*>(orderbuf.p);
    {   // [ SchedCtl + progress[B] | pad to 16 | cyc[B][4] ]: zeroed by one memset per step
      const size_t ctl = (128 + (size_t)batch * sizeof(uint32_t) + 15) / 16 * 16;
      a.sched_bytes = ctl + (size_t)batch * 4 * sizeof(unsigned long long);
      if ((rc = schedbuf.alloc(a.sched_bytes))) return rc;
      BCN_HIP(hipMemset(schedbuf.p, 0, a.sched_bytes));
      a.sched_ctl = schedbuf.p;
      a.cyc = reinterpret_cast<unsigned long long*>(static_cast<char*>(schedbuf.p) + ctl);
    }
    fast_ok = ns2d_fast_supported<real>(a);
    if (fast_ok && (a.fscr_stride = ns2d_fast_scratch_elems<real>(a)) > 0) {
      if ((rc = fscrbuf.alloc((size_t)batch * a.fscr_stride * sizeof(real)))) return rc;
      BCN_HIP(hipMemset(fscrbuf.p, 0, fscrbuf.bytes));
      a.fscr = static_cast<real*>(fscrbuf.p);
    }
    variant = fast_ok ? 1 : 0;
    // both precisions: the extrapolating plan with PROVEN landings (plan 3: every stop sweep is the reference's, ns2d_fast_impl.h).
    // (Until round 5 float64 ran the lower-bound plan 1: plan 3's landings were not yet verified by a bound.)
    a.conv_plan = 3;
    // the landing test of plans 2 / 3 rests on a bound (BCN_CONV_GUARD, bcn_common.h) that holds for grids with no side below 48
    // cells -- every grid the reference can construct (nx = 50 L, ny = 50 H, L, H >= 1); smaller ones: the proven plan
    if (!guard_holds()) a.conv_plan = 1;
    // rayleigh float32: a solve opens with unevaluated double sweeps up to 15/16 of the previous timestep's count minus the stretch
    // in front of the stop where the residual is already below the landing guard (spec_start 17: ns2d_fast_impl.h); the landing
    // must find the residual above the guard or the solve is repeated without the guess.  Measured on the bench workload, repeats
    // per step of 102 400 solves / cycles per sweep -- with the global guard alone and a fixed fraction: 7/8 27 252 / 916,
    // 6/8 2 021 / 818, 5/8 205 / 827, 4/8 5 / 837, off 0 / 888; with the slow-mode guard: 6/8 789, the zone-aware opening 219 / 778
    // (the unverified rule of round 3 at 7/8: 750).  mixing's counts drop by up to 9x from one timestep to the next: off; the
    // float64 kernels are built without the opening
    a.spec_start = (a.kind == 0 && sizeof(real) == 4) ? 17 : 0;
    // mixing float32: the ordered part of the scalar transport as parallel passes while their count stays within 24 (12 at the
    // reference's u_max; ns2d_fast2_impl.h); float64 keeps the reference's ordered sweep
    a.transport_iter = (a.kind == 1 && sizeof(real) == 4) ? 24 : 0;
    // slow-mode landing guard: the constants of the grids the reference's defaults construct are built in (computed by
    // beacon_amd/stoprule.py, checked by tests/test_oracle.py); any other grid: bcn_set_slow_mode_bound, else BCN_CONV_GUARD alone
    static const struct { int nx, ny, kind; double cx, cut[2], cl[2]; } kBuiltin[] = {
        {128, 64, 0, 0.25, {0.9, 0.8}, {1.00020, 1.00568}},    // rayleigh L = 2.56, H = 1.28 (the bench workload)
        {50, 50, 0, 0.25, {0.9, 0.8}, {1.00020, 1.00020}},     // rayleigh.py:20-27 defaults L = H = 1
        {100, 100, 1, 0.25, {0.9, 0.8}, {1.00167, 1.00594}},   // mixing.py:20-28 defaults L = H = 1
    };
    for (const auto& e : kBuiltin)
      if (e.nx == a.nx && e.ny == a.ny && e.kind == a.kind && fabs((double)a.cx - e.cx) < 1e-6) set_slow_mode_bound(2, e.cut, e.cl);
    return BCN_OK;
  }
  int set_slow_mode_bound(int n, const double* cutoff, const double* bound) override {
    for (int k = 0; k < n; k++)
      if (!(cutoff[k] > 0.0 && cutoff[k] < 1.0) || !(bound[k] >= 1.0)) {
        bcn_set_error("bcn_set_slow_mode_bound: cutoff %g must lie in (0, 1), bound %g must be >= 1", cutoff[k], bound[k]);
        return BCN_ERR_ARG;
      }
    for (int k = 0; k < 2; k++) {
      a.slow_l2lc[k] = k < n ? (float)log2(cutoff[k]) : 0.f;
      // rounded UP to float: the kernels compare against it
      a.slow_cl[k] = k < n ? nextafterf((float)bound[k], INFINITY) : INFINITY;
      slow_cut[k] = k < n ? cutoff[k] : 0.0;
    }
    return BCN_OK;
  }
  int get_slow_mode_bound(double* cutoff, double* bound) const override {
    int n = 0;
    for (int k = 0; k < 2; k++)
      if (slow_cut[k] > 0.0) { cutoff[n] = slow_cut[k]; bound[n] = (double)a.slow_cl[k]; n++; }
    return n;
  }
  double slow_cut[2] = {0.0, 0.0};
  ~NS2DEnv() override {
    DeviceGuard g(device);
    fields.release(); work.release(); fscrbuf.release(); obs_hist.release(); a_last.release(); ia_last.release();
    stpbuf.release(); sweepbuf.release(); orderbuf.release(); schedbuf.release(); statusbuf.release();
  }
  size_t state_elems() const override { return 4 * (size_t)a.ncell; }
  // state buffer layout: [B][4][ncell]; device layout: [4][B][ncell]
  int copy_state(void* buf, int is_device, hipStream_t s, bool out) {
    const size_t row = (size_t)a.ncell * sizeof(real);
    real* f = static_cast<real*>(fields.p);
    for (int k = 0; k < 4; k++) {
      char* ext = static_cast<char*>(buf) + (size_t)k * row;
      real* dev = f + (size_t)k * batch * a.ncell;
      hipMemcpyKind kind = is_device ? hipMemcpyDeviceToDevice : (out ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice);
      if (out) BCN_HIP(hipMemcpy2DAsync(ext, 4 * row, dev, row, row, batch, kind, s));
      else BCN_HIP(hipMemcpy2DAsync(dev, row, ext, 4 * row, row, batch, kind, s));
    }
    if (!is_device) BCN_HIP(hipStreamSynchronize(s));
    return BCN_OK;
  }
  int get_state(void* buf, int is_device, hipStream_t s) override { return copy_state(buf, is_device, s, true); }
  int set_state(const void* buf, int is_device, hipStream_t s) override {
    return copy_state(const_cast<void*>(buf), is_device, s, false);
  }
  int set_variant(int v) override { variant = (v == 1 && fast_ok) ? 1 : 0; host.launched = nullptr; return variant; }
  void set_mask(const uint8_t* m) override { a.mask = m; }
  int set_sched(int mode, int grid, int q, int lpt_min_batch) override {
    a.sched_mode = mode; a.sched_grid = grid; a.sched_q_user = q; a.lpt_min_batch = lpt_min_batch;
    return BCN_OK;
  }
  int set_option(const char* name, int value) override {
    if (!strcmp(name, "conv_plan") && value >= 0 && value <= 3) {
      if (value >= 2 && !guard_holds()) { bcn_set_error("conv_plan %d needs a grid with no side below 48 cells (%dx%d): plans 0, 1 only", value, a.nx, a.ny); return BCN_ERR_ARG; }
      a.conv_plan = value;
      return BCN_OK;
    }
    if (!strcmp(name, "plan_overshoot") && value >= 0 && value <= 64) { a.plan_overshoot = value; return BCN_OK; }
    if (!strcmp(name, "verify_conv")) { a.verify_conv = value ? 1 : 0; return BCN_OK; }
    if (!strcmp(name, "spec_start") && value >= 0 && value <= 17) { a.spec_start = value; return BCN_OK; }
    if (!strcmp(name, "transport_iter") && value >= 0 && value <= 64) { a.transport_iter = value; return BCN_OK; }
    if (!strcmp(name, "sched_tail") && value >= 0 && value <= 1024) { host.sched_tail = value; return BCN_OK; }
    if (!strcmp(name, "generic_threads") && (value == 0 || value == 256 || value == 1024)) { host.generic_nt = value; return BCN_OK; }
    return bcn_env_s::set_option(name, value);
  }
  int get_counters(uint64_t* host, hipStream_t s) override {
    BCN_HIP(hipMemcpyAsync(host, a.cyc, (size_t)batch * 4 * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    BCN_HIP(hipStreamSynchronize(s));
    return BCN_OK;
  }
  typedef int (*plugin_fn)(const void*, int, void*);
  plugin_fn plugin = nullptr;        // register-resident kernel of a grid that is not built in (bcn_set_fast_plugin)
  int set_fast_plugin(void* fn, size_t scratch_elems) override {
    if (!fn) { plugin = nullptr; fast_ok = ns2d_fast_supported<real>(a); variant = fast_ok ? 1 : 0; return BCN_OK; }
    if (scratch_elems > 0) {
      DeviceGuard g(device);
      fscrbuf.release();
      int rc = fscrbuf.alloc((size_t)batch * scratch_elems * sizeof(real));
      if (rc) return rc;
      BCN_HIP(hipMemset(fscrbuf.p, 0, fscrbuf.bytes));
      a.fscr = static_cast<real*>(fscrbuf.p);
      a.fscr_stride = scratch_elems;
    }
    plugin = reinterpret_cast<plugin_fn>(fn);
    fast_ok = true; variant = 1; host.launched = nullptr;
    return BCN_OK;
  }
  NS2DHost host;   // launcher-side state: kernel name of the last step, options that no kernel reads
  const char* kernel_name() const override {
    return host.launched ? host.launched : (variant == 1 ? "ns2d_fast_step" : "ns2d_generic_step");
  }
  int launch(hipStream_t s) {
    a.host = &host;
    if (variant == 1 && plugin) {
      // BCN_ERR_UNSUPPORTED: the batch does not fit the plugin's addressing (ns2d_fast4_impl.h: 32-bit offsets from a
      // replica's u): the generic kernel takes the step
      const int rc = plugin(&a, batch, s);
      if (rc != BCN_ERR_UNSUPPORTED) return rc;
      return ns2d_launch_generic<real>(a, batch, s);
    }
    if (variant == 1) return ns2d_launch_fast<real>(a, batch, s);
    return ns2d_launch_generic<real>(a, batch, s);
  }
};

template <typename real>
void ns2d_common(NS2DArgs<real>& a, int nx, int ny, double dx, double dy, double dt, double tol) {
  a.nx = nx; a.ny = ny; a.sx = nx + 2; a.ncell = (nx + 2) * (ny + 2);
  a.dt = (real)dt; a.rdx = (real)(1.0 / dx); a.rdy = (real)(1.0 / dy);
  a.rdx2 = (real)(1.0 / (dx * dx)); a.rdy2 = (real)(1.0 / (dy * dy));
  const double den = dx * dx + dy * dy;
  a.cx = (real)(0.5 * dy * dy / den);
  a.cy = (real)(0.5 * dx * dx / den);
  a.cb = (real)(0.5 * dx * dx * dy * dy / den / dt);
  a.tol = (real)tol;
}

template <typename real>
int make_rayleigh(const bcn_rayleigh_cfg* c, int batch, int dtype, int device, bcn_env_t* out) {
  auto* e = new (std::nothrow) NS2DEnv<real>();
  if (!e) { bcn_set_error("out of host memory"); return BCN_ERR_ARG; }
  e->kind = BCN_RAYLEIGH; e->batch = batch; e->dtype = dtype; e->device = device; e->esz = sizeof(real);
  NS2DArgs<real>& a = e->a;
  ns2d_common(a, c->nx, c->ny, c->dx, c->dy, c->dt, c->tol);
  a.kind = 0; a.ndt_act = c->ndt_act; a.n_act = c->n_act; a.itmax = c->itmax;
  a.n_sgts = c->n_sgts; a.nx_sgts = c->nx_sgts;
  a.nxo = c->nx_obs_pts; a.nyo = c->ny_obs_pts; a.nx_obs = c->nx_obs; a.ny_obs = c->ny_obs;
  a.n_obs_steps = c->n_obs_steps; a.n_obs = 3 * c->n_obs_steps * c->nx_obs_pts * c->ny_obs_pts;
  a.kmom = (real)sqrt(c->pr / c->ra);
  a.ksc = (real)(1.0 / sqrt(c->pr * c->ra));
  a.Tc = (real)c->Tc; a.Th = (real)c->Th; a.C = (real)c->C;
  a.rwd_scale = (real)(1.0 / (0.5 * c->dy * c->nx));
  e->n_obs = a.n_obs; e->n_act = c->n_sgts; e->ndt_act = a.ndt_act;
  int rc = e->init();
  if (rc) { delete e; return rc; }
  *out = e;
  return BCN_OK;
}

template <typename real>
int make_mixing(const bcn_mixing_cfg* c, int batch, int dtype, int device, bcn_env_t* out) {
  auto* e = new (std::nothrow) NS2DEnv<real>();
  if (!e) { bcn_set_error("out of host memory"); return BCN_ERR_ARG; }
  e->kind = BCN_MIXING; e->batch = batch; e->dtype = dtype; e->device = device; e->esz = sizeof(real);
  NS2DArgs<real>& a = e->a;
  ns2d_common(a, c->nx, c->ny, c->dx, c->dy, c->dt, c->tol);
  a.kind = 1; a.ndt_act = c->ndt_act; a.n_act = c->n_act; a.itmax = c->itmax;
  a.n_sgts = 0; a.nx_sgts = 1;
  a.nxo = c->nx_obs_pts; a.nyo = c->ny_obs_pts; a.nx_obs = c->nx_obs; a.ny_obs = c->ny_obs;
  a.n_obs_steps = c->n_obs_steps; a.n_obs = 3 * c->n_obs_steps * c->nx_obs_pts * c->ny_obs_pts;
  a.i_min = c->i_min; a.i_max = c->i_max; a.j_min = c->j_min; a.j_max = c->j_max;
  a.kmom = (real)(1.0 / c->re);
  a.ksc = (real)(1.0 / c->pe);
  a.u_max = (real)c->u_max; a.ref_c = (real)c->ref_c; a.C0 = (real)c->C0;
  e->n_obs = a.n_obs; e->n_act = 1; e->ndt_act = a.ndt_act;
  int rc = e->init();
  if (rc) { delete e; return rc; }
  *out = e;
  return BCN_OK;
}

// ------------------------------------------------------------------------------------------
// 1D envs
// ------------------------------------------------------------------------------------------
template <typename real>
struct Env1D : bcn_env_s {
  Env1DArgs<real> a{};
  int nfields = 4;
  DevBuf fields, a_last, a_prev, stpbuf, nctrbuf;
  const char* kname = "";

  int init(int n_actions) {
    int rc;
    const size_t per = (size_t)batch * a.n * sizeof(real);
    if ((rc = fields.alloc(4 * per))) return rc;
    BCN_HIP(hipMemset(fields.p, 0, 4 * per));
    real* f = static_cast<real*>(fields.p);
    const size_t n = (size_t)batch * a.n;
    a.f0 = f; a.f1 = f + n; a.f2 = f + 2 * n; a.f3 = f + 3 * n;
    if ((rc = a_last.alloc((size_t)batch * n_actions * sizeof(real)))) return rc;
    if ((rc = a_prev.alloc((size_t)batch * n_actions * sizeof(real)))) return rc;
    BCN_HIP(hipMemset(a_last.p, 0, a_last.bytes));
    BCN_HIP(hipMemset(a_prev.p, 0, a_prev.bytes));
    a.a_last = static_cast<real*>(a_last.p);
    a.a_prev = static_cast<real*>(a_prev.p);
    if ((rc = stpbuf.alloc((size_t)batch * sizeof(int32_t)))) return rc;
    BCN_HIP(hipMemset(stpbuf.p, 0, stpbuf.bytes));
    stp = a.stp = static_cast<int32_t*>(stpbuf.p);
    if ((rc = nctrbuf.alloc((size_t)batch * sizeof(uint32_t)))) return rc;
    BCN_HIP(hipMemset(nctrbuf.p, 0, nctrbuf.bytes));
    a.nctr = static_cast<uint32_t*>(nctrbuf.p);
    a.nsigma = 0; a.nseed_lo = 0; a.nseed_hi = 0; a.noff = 0;
    return BCN_OK;
  }
  ~Env1D() override {
    DeviceGuard g(device);
    fields.release(); a_last.release(); a_prev.release(); stpbuf.release(); nctrbuf.release();
  }
  int set_noise(double sigma, uint64_t seed, int64_t replica_offset) override {
    if (!(sigma >= 0) || replica_offset < 0) { bcn_set_error("bcn_set_noise: sigma >= 0 and replica_offset >= 0"); return BCN_ERR_ARG; }
    a.nsigma = (real)sigma; a.nseed_lo = (uint32_t)seed; a.nseed_hi = (uint32_t)(seed >> 32); a.noff = (int)replica_offset;
    DeviceGuard g(device);
    BCN_HIP(hipMemset(nctrbuf.p, 0, nctrbuf.bytes));
    return BCN_OK;
  }
  size_t state_elems() const override { return (size_t)nfields * a.n; }
  int copy_state(void* buf, int is_device, hipStream_t s, bool out) {
    const size_t row = (size_t)a.n * sizeof(real);
    real* f = static_cast<real*>(fields.p);
    for (int k = 0; k < nfields; k++) {
      char* ext = static_cast<char*>(buf) + (size_t)k * row;
      real* dev = f + (size_t)k * batch * a.n;
      hipMemcpyKind kind = is_device ? hipMemcpyDeviceToDevice : (out ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice);
      if (out) BCN_HIP(hipMemcpy2DAsync(ext, nfields * row, dev, row, row, batch, kind, s));
      else BCN_HIP(hipMemcpy2DAsync(dev, row, ext, nfields * row, row, batch, kind, s));
    }
    if (!is_device) BCN_HIP(hipStreamSynchronize(s));
    return BCN_OK;
  }
  int get_state(void* buf, int is_device, hipStream_t s) override { return copy_state(buf, is_device, s, true); }
  int set_state(const void* buf, int is_device, hipStream_t s) override {
    return copy_state(const_cast<void*>(buf), is_device, s, false);
  }
  const char* kernel_name() const override { return kname; }
  void note_kernel(const char* n) override { kname = n; }
  void set_mask(const uint8_t* m) override { a.mask = m; }
  int set_option(const char* name, int value) override {
    if (!strcmp(name, "cells_per_thread") && (value == 0 || value == 1 || value == 2 || value == 4 || value == 8)) { a.force_k = value; return BCN_OK; }
    if (!strcmp(name, "one_wave") && value >= 0 && value <= 2) { a.one_wave = value; return BCN_OK; }   // 2: without the packed float32 kernel
    return bcn_env_s::set_option(name, value);
  }
};

template <typename real>
int make_burgers(const bcn_burgers_cfg* c, int batch, int dtype, int device, bcn_env_t* out) {
  auto* e = new (std::nothrow) Env1D<real>();
  if (!e) { bcn_set_error("out of host memory"); return BCN_ERR_ARG; }
  e->kind = BCN_BURGERS; e->batch = batch; e->dtype = dtype; e->device = device; e->esz = sizeof(real);
  e->nfields = 3; e->kname = "burgers_step_k";
  Env1DArgs<real>& a = e->a;
  a.n = a.nx = c->nx; a.ndt_act = c->ndt_act; a.n_act = c->n_act; a.n_obs = c->n_obs_pts;
  a.ctrl_pos = c->ctrl_pos; a.n_obs_pts = c->n_obs_pts;
  a.u_target = (real)c->u_target; a.amp = (real)c->amp;
  a.dx = (real)c->dx; a.rdx = (real)(1.0 / c->dx); a.dt = (real)c->dt;
  e->n_obs = c->n_obs_pts; e->n_act = 1; e->ndt_act = a.ndt_act;
  int rc = e->init(1);
  if (rc) { delete e; return rc; }
  *out = e;
  return BCN_OK;
}

template <typename real>
int make_shkadov(const bcn_shkadov_cfg* c, int batch, int dtype, int device, bcn_env_t* out) {
  auto* e = new (std::nothrow) Env1D<real>();
  if (!e) { bcn_set_error("out of host memory"); return BCN_ERR_ARG; }
  e->kind = BCN_SHKADOV; e->batch = batch; e->dtype = dtype; e->device = device; e->esz = sizeof(real);
  e->nfields = 4; e->kname = "shkadov_step_k";
  Env1DArgs<real>& a = e->a;
  a.n = a.nx = c->nx; a.ndt_act = c->ndt_act; a.n_act = c->n_act; a.n_obs = c->n_obs * c->n_jets;
  a.n_jets = c->n_jets; a.jet_pos = c->jet_pos; a.jet_hw = c->jet_hw; a.jet_space = c->jet_space;
  a.l_obs = c->l_obs; a.l_rwd = c->l_rwd; a.n_obs_jet = c->n_obs; a.obs_stride = c->obs_stride;
  a.n_interp = c->n_interp;
  a.delta_p = (real)(1.0 / (5.0 * c->delta));
  a.jet_amp = (real)c->jet_amp; a.eps = (real)c->eps; a.h_blow = (real)c->h_blow;
  a.blowup_rwd = (real)c->blowup_rwd;
  a.dx = (real)c->dx; a.rdx = (real)(1.0 / c->dx); a.dt = (real)c->dt;
  e->n_obs = a.n_obs; e->n_act = c->n_jets; e->ndt_act = a.ndt_act;
  int rc = e->init(c->n_jets);
  if (rc) { delete e; return rc; }
  *out = e;
  return BCN_OK;
}

template <typename real>
int make_sloshing(const bcn_sloshing_cfg* c, int batch, int dtype, int device, bcn_env_t* out) {
  auto* e = new (std::nothrow) Env1D<real>();
  if (!e) { bcn_set_error("out of host memory"); return BCN_ERR_ARG; }
  e->kind = BCN_SLOSHING; e->batch = batch; e->dtype = dtype; e->device = device; e->esz = sizeof(real);
  e->nfields = 4; e->kname = "sloshing_step_k";
  Env1DArgs<real>& a = e->a;
  a.nx = c->nx; a.n = c->nx + 2; a.ndt_act = c->ndt_act; a.n_act = c->n_act;
  a.n_obs = c->nx / 2 + (c->nx % 2 ? 1 : 0);
  a.n_interp = c->n_interp;
  a.g = (real)c->g; a.amp = (real)c->amp; a.alpha = (real)c->alpha;
  a.dx = (real)c->dx; a.rdx = (real)(1.0 / c->dx); a.dt = (real)c->dt;
  e->n_obs = a.n_obs; e->n_act = 1; e->ndt_act = a.ndt_act;
  int rc = e->init(1);
  if (rc) { delete e; return rc; }
  *out = e;
  return BCN_OK;
}

int check_create(const void* cfg, int batch, int dtype, int device, bcn_env_t* out) {
  if (!cfg || !out) { bcn_set_error("null cfg/out"); return BCN_ERR_ARG; }
  if (batch <= 0) { bcn_set_error("batch must be > 0"); return BCN_ERR_ARG; }
  if (dtype != BCN_F32 && dtype != BCN_F64) { bcn_set_error("dtype must be BCN_F32 or BCN_F64"); return BCN_ERR_ARG; }
  int ndev = 0;
  BCN_HIP(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) { bcn_set_error("device %d out of range (%d visible)", device, ndev); return BCN_ERR_ARG; }
  return BCN_OK;
}

#define BCN_CHECK_KIND(h, K)                                                        \
  if (!(h) || (h)->kind != (K)) { bcn_set_error("handle is not a " #K " env"); return BCN_ERR_ARG; }

template <typename real>
static int ns2d_reset_t(bcn_env_t h, const void* init, void* obs, void* stream) {
  auto* e = static_cast<NS2DEnv<real>*>(h);
  DeviceGuard g(e->device);
  NS2DArgs<real> a = e->a;
  a.init_fields = static_cast<const real*>(init);
  a.obs_out = static_cast<real*>(obs);
  return ns2d_launch_reset<real>(a, e->batch, static_cast<hipStream_t>(stream));
}

template <typename real>
static int ns2d_step_t(bcn_env_t h, const void* actions, const int32_t* iactions, void* actions_norm, void* obs,
                       void* rwd, uint8_t* done, uint8_t* trunc, int32_t* status, int32_t* sweeps, void* stream) {
  auto* e = static_cast<NS2DEnv<real>*>(h);
  DeviceGuard g(e->device);
  NS2DArgs<real>& a = e->a;
  a.actions = static_cast<const real*>(actions);
  a.iactions = iactions;
  a.actions_norm = static_cast<real*>(actions_norm);
  a.obs_out = static_cast<real*>(obs);
  a.rwd_out = static_cast<real*>(rwd);
  a.done = done; a.trunc = trunc; a.sweeps = sweeps;
  a.status = status ? status : e->status_int;   // the chunk scheduler carries a replica's status across chunks
  return e->launch(static_cast<hipStream_t>(stream));
}

template <typename real>
static Env1DArgs<real>& env1d_io(bcn_env_t h, const void* actions, const void* noise, const void* init, void* obs,
                                 void* rwd, uint8_t* done, uint8_t* trunc, int32_t* status) {
  auto* e = static_cast<Env1D<real>*>(h);
  Env1DArgs<real>& a = e->a;
  a.actions = static_cast<const real*>(actions);
  a.noise = static_cast<const real*>(noise);
  a.init_fields = static_cast<const real*>(init);
  a.obs_out = static_cast<real*>(obs);
  a.rwd_out = static_cast<real*>(rwd);
  a.done = done; a.trunc = trunc; a.status = status;
  return a;
}

}  // namespace

extern "C" {

// ---- rayleigh --------------------------------------------------------------------------------
int bcn_rayleigh_create(const bcn_rayleigh_cfg* c, int batch, int dtype, int device, bcn_env_t* out) {
  int rc = check_create(c, batch, dtype, device, out);
  if (rc) return rc;
  if (c->nx < 2 || c->ny < 2 || c->n_sgts < 1 || c->n_sgts > 64 || c->nx_sgts < 1 || c->ndt_act < 0) {
    bcn_set_error("rayleigh cfg out of range (nx,ny >= 2; 1 <= n_sgts <= 64)");
    return BCN_ERR_ARG;
  }
  DeviceGuard g(device);
  return dtype == BCN_F32 ? make_rayleigh<float>(c, batch, dtype, device, out)
                          : make_rayleigh<double>(c, batch, dtype, device, out);
}

int bcn_rayleigh_reset(bcn_env_t h, const void* init_fields_dev, void* obs_dev, void* stream) {
  BCN_CHECK_KIND(h, BCN_RAYLEIGH);
  return h->dtype == BCN_F32 ? ns2d_reset_t<float>(h, init_fields_dev, obs_dev, stream)
                             : ns2d_reset_t<double>(h, init_fields_dev, obs_dev, stream);
}

int bcn_rayleigh_step(bcn_env_t h, const void* actions_dev, void* actions_norm_dev, void* obs_dev, void* rwd_dev,
                      uint8_t* done_dev, uint8_t* trunc_dev, int32_t* status_dev, int32_t* sweeps_dev,
                      void* stream) {
  BCN_CHECK_KIND(h, BCN_RAYLEIGH);
  return h->dtype == BCN_F32
             ? ns2d_step_t<float>(h, actions_dev, nullptr, actions_norm_dev, obs_dev, rwd_dev, done_dev, trunc_dev,
                                  status_dev, sweeps_dev, stream)
             : ns2d_step_t<double>(h, actions_dev, nullptr, actions_norm_dev, obs_dev, rwd_dev, done_dev, trunc_dev,
                                   status_dev, sweeps_dev, stream);
}

// ---- mixing ----------------------------------------------------------------------------------
int bcn_mixing_create(const bcn_mixing_cfg* c, int batch, int dtype, int device, bcn_env_t* out) {
  int rc = check_create(c, batch, dtype, device, out);
  if (rc) return rc;
  if (c->nx < 2 || c->ny < 2 || c->ndt_act < 0) { bcn_set_error("mixing cfg out of range"); return BCN_ERR_ARG; }
  DeviceGuard g(device);
  return dtype == BCN_F32 ? make_mixing<float>(c, batch, dtype, device, out)
                          : make_mixing<double>(c, batch, dtype, device, out);
}

int bcn_mixing_reset(bcn_env_t h, void* obs_dev, void* stream) {
  BCN_CHECK_KIND(h, BCN_MIXING);
  return h->dtype == BCN_F32 ? ns2d_reset_t<float>(h, nullptr, obs_dev, stream)
                             : ns2d_reset_t<double>(h, nullptr, obs_dev, stream);
}

int bcn_mixing_step(bcn_env_t h, const int32_t* actions_dev, void* obs_dev, void* rwd_dev, uint8_t* done_dev,
                    uint8_t* trunc_dev, int32_t* status_dev, int32_t* sweeps_dev, void* stream) {
  BCN_CHECK_KIND(h, BCN_MIXING);
  return h->dtype == BCN_F32
             ? ns2d_step_t<float>(h, nullptr, actions_dev, nullptr, obs_dev, rwd_dev, done_dev, trunc_dev,
                                  status_dev, sweeps_dev, stream)
             : ns2d_step_t<double>(h, nullptr, actions_dev, nullptr, obs_dev, rwd_dev, done_dev, trunc_dev,
                                   status_dev, sweeps_dev, stream);
}

// ---- 1D envs ---------------------------------------------------------------------------------
#define BCN_1D_CALL(h, FN, ...)                                                                       \
  (h->dtype == BCN_F32 ? FN<float>(env1d_io<float>(h, __VA_ARGS__), h->batch, static_cast<hipStream_t>(stream)) \
                       : FN<double>(env1d_io<double>(h, __VA_ARGS__), h->batch, static_cast<hipStream_t>(stream)))

int bcn_burgers_create(const bcn_burgers_cfg* c, int batch, int dtype, int device, bcn_env_t* out) {
  int rc = check_create(c, batch, dtype, device, out);
  if (rc) return rc;
  if (c->nx < 8 || c->nx > 8192 || c->ctrl_pos < c->n_obs_pts || c->ctrl_pos >= c->nx || c->n_obs_pts > 64) {
    bcn_set_error("burgers cfg out of range (8 <= nx <= 8192, n_obs_pts <= ctrl_pos < nx)");
    return BCN_ERR_ARG;
  }
  DeviceGuard g(device);
  return dtype == BCN_F32 ? make_burgers<float>(c, batch, dtype, device, out)
                          : make_burgers<double>(c, batch, dtype, device, out);
}
int bcn_burgers_reset(bcn_env_t h, void* obs_dev, void* stream) {
  BCN_CHECK_KIND(h, BCN_BURGERS);
  DeviceGuard g(h->device);
  return BCN_1D_CALL(h, burgers_launch_reset, nullptr, nullptr, nullptr, obs_dev, nullptr, nullptr, nullptr, nullptr);
}
int bcn_burgers_step(bcn_env_t h, const void* actions_dev, const void* noise_dev, void* obs_dev, void* rwd_dev,
                     uint8_t* done_dev, uint8_t* trunc_dev, int32_t* status_dev, void* stream) {
  BCN_CHECK_KIND(h, BCN_BURGERS);
  DeviceGuard g(h->device);
  bcn_env1d_launched = nullptr;
  const int rc_ = BCN_1D_CALL(h, burgers_launch_step, actions_dev, noise_dev, nullptr, obs_dev, rwd_dev, done_dev, trunc_dev,
                     status_dev);
  h->note_kernel(bcn_env1d_launched ? bcn_env1d_launched : "burgers_step_k");
  return rc_;
}

int bcn_shkadov_create(const bcn_shkadov_cfg* c, int batch, int dtype, int device, bcn_env_t* out) {
  int rc = check_create(c, batch, dtype, device, out);
  if (rc) return rc;
  const int last = c->jet_pos + (c->n_jets - 1) * c->jet_space;
  if (c->nx < 16 || c->nx > 8192 || c->n_jets < 1 || c->n_jets > 64 || 2 * c->jet_hw >= c->jet_space ||
      c->jet_pos - c->l_obs < 0 || c->jet_pos - c->jet_hw < 1 || last + c->l_rwd > c->nx ||
      last + c->jet_hw > c->nx - 2 || c->n_interp < 1) {
    bcn_set_error("shkadov cfg out of range (16 <= nx <= 8192, 1 <= n_jets <= 64, non-overlapping jets inside the domain)");
    return BCN_ERR_ARG;
  }
  DeviceGuard g(device);
  return dtype == BCN_F32 ? make_shkadov<float>(c, batch, dtype, device, out)
                          : make_shkadov<double>(c, batch, dtype, device, out);
}
int bcn_shkadov_reset(bcn_env_t h, const void* init_fields_dev, void* obs_dev, void* stream) {
  BCN_CHECK_KIND(h, BCN_SHKADOV);
  DeviceGuard g(h->device);
  return BCN_1D_CALL(h, shkadov_launch_reset, nullptr, nullptr, init_fields_dev, obs_dev, nullptr, nullptr, nullptr,
                     nullptr);
}
int bcn_shkadov_step(bcn_env_t h, const void* actions_dev, const void* noise_dev, void* obs_dev, void* rwd_dev,
                     uint8_t* done_dev, uint8_t* trunc_dev, int32_t* status_dev, void* stream) {
  BCN_CHECK_KIND(h, BCN_SHKADOV);
  DeviceGuard g(h->device);
  return BCN_1D_CALL(h, shkadov_launch_step, actions_dev, noise_dev, nullptr, obs_dev, rwd_dev, done_dev, trunc_dev,
                     status_dev);
}

int bcn_sloshing_create(const bcn_sloshing_cfg* c, int batch, int dtype, int device, bcn_env_t* out) {
  int rc = check_create(c, batch, dtype, device, out);
  if (rc) return rc;
  if (c->nx < 4 || c->nx + 2 > 8192 || c->n_interp < 1) { bcn_set_error("sloshing cfg out of range"); return BCN_ERR_ARG; }
  DeviceGuard g(device);
  return dtype == BCN_F32 ? make_sloshing<float>(c, batch, dtype, device, out)
                          : make_sloshing<double>(c, batch, dtype, device, out);
}
int bcn_sloshing_reset(bcn_env_t h, const void* init_fields_dev, void* obs_dev, void* stream) {
  BCN_CHECK_KIND(h, BCN_SLOSHING);
  DeviceGuard g(h->device);
  return BCN_1D_CALL(h, sloshing_launch_reset, nullptr, nullptr, init_fields_dev, obs_dev, nullptr, nullptr, nullptr,
                     nullptr);
}
int bcn_sloshing_step(bcn_env_t h, const void* actions_dev, void* obs_dev, void* rwd_dev, uint8_t* done_dev,
                      uint8_t* trunc_dev, int32_t* status_dev, void* stream) {
  BCN_CHECK_KIND(h, BCN_SLOSHING);
  DeviceGuard g(h->device);
  bcn_env1d_launched = nullptr;
  const int rc_ = BCN_1D_CALL(h, sloshing_launch_step, actions_dev, nullptr, nullptr, obs_dev, rwd_dev, done_dev, trunc_dev,
                     status_dev);
  h->note_kernel(bcn_env1d_launched ? bcn_env1d_launched : "sloshing_step_k");
  return rc_;
}

// ---- common ----------------------------------------------------------------------------------
int bcn_env_kind(bcn_env_t h) { return h ? h->kind : -1; }
int bcn_batch(bcn_env_t h) { return h ? h->batch : 0; }
int bcn_dtype(bcn_env_t h) { return h ? h->dtype : -1; }
int bcn_n_obs(bcn_env_t h) { return h ? h->n_obs : 0; }
int bcn_n_act(bcn_env_t h) { return h ? h->n_act : 0; }
int bcn_ndt_act(bcn_env_t h) { return h ? h->ndt_act : 0; }
int bcn_device(bcn_env_t h) { return h ? h->device : -1; }
size_t bcn_state_elems(bcn_env_t h) { return h ? h->state_elems() : 0; }

int bcn_get_state(bcn_env_t h, void* buf, int is_device, void* stream) {
  if (!h || !buf) { bcn_set_error("null handle/buffer"); return BCN_ERR_ARG; }
  DeviceGuard g(h->device);
  return h->get_state(buf, is_device, static_cast<hipStream_t>(stream));
}
int bcn_set_state(bcn_env_t h, const void* buf, int is_device, void* stream) {
  if (!h || !buf) { bcn_set_error("null handle/buffer"); return BCN_ERR_ARG; }
  DeviceGuard g(h->device);
  return h->set_state(buf, is_device, static_cast<hipStream_t>(stream));
}
int bcn_get_stp(bcn_env_t h, int32_t* buf_host, void* stream) {
  if (!h || !buf_host) { bcn_set_error("null handle/buffer"); return BCN_ERR_ARG; }
  DeviceGuard g(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  BCN_HIP(hipMemcpyAsync(buf_host, h->stp, (size_t)h->batch * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  BCN_HIP(hipStreamSynchronize(s));
  return BCN_OK;
}
int bcn_set_stp(bcn_env_t h, const int32_t* buf_host, void* stream) {
  if (!h || !buf_host) { bcn_set_error("null handle/buffer"); return BCN_ERR_ARG; }
  DeviceGuard g(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  BCN_HIP(hipMemcpyAsync(h->stp, buf_host, (size_t)h->batch * sizeof(int32_t), hipMemcpyHostToDevice, s));
  BCN_HIP(hipStreamSynchronize(s));
  return BCN_OK;
}
int bcn_set_mask(bcn_env_t h, const uint8_t* mask_dev) {
  if (!h) { bcn_set_error("null handle"); return BCN_ERR_ARG; }
  h->set_mask(mask_dev);
  return BCN_OK;
}
int bcn_set_variant(bcn_env_t h, int variant) { return h ? h->set_variant(variant) : 0; }
int bcn_get_counters(bcn_env_t h, uint64_t* buf_host, void* stream) {
  if (!h || !buf_host) { bcn_set_error("null handle/buffer"); return BCN_ERR_ARG; }
  DeviceGuard g(h->device);
  return h->get_counters(buf_host, static_cast<hipStream_t>(stream));
}
int bcn_get_counters_n(bcn_env_t h, uint64_t* buf_host, int words_per_replica, void* stream) {
  if (!h || !buf_host || words_per_replica < 1) { bcn_set_error("null handle/buffer or words_per_replica < 1"); return BCN_ERR_ARG; }
  if (words_per_replica == BCN_COUNTER_WORDS) return bcn_get_counters(h, buf_host, stream);
  std::vector<uint64_t> tmp((size_t)h->batch * BCN_COUNTER_WORDS);
  int rc = bcn_get_counters(h, tmp.data(), stream);
  if (rc) return rc;
  for (int b = 0; b < h->batch; b++)
    for (int k = 0; k < words_per_replica; k++)
      buf_host[(size_t)b * words_per_replica + k] = k < BCN_COUNTER_WORDS ? tmp[(size_t)b * BCN_COUNTER_WORDS + k] : 0;
  return BCN_OK;
}
int bcn_api_version(void) { return BCN_API_VERSION; }
int bcn_set_fast_plugin(bcn_env_t h, void* launch_fn, size_t scratch_elems) {
  if (!h) { bcn_set_error("null handle"); return BCN_ERR_ARG; }
  return h->set_fast_plugin(launch_fn, scratch_elems);
}
int bcn_set_noise(bcn_env_t h, double sigma, uint64_t seed, int64_t replica_offset) {
  if (!h) { bcn_set_error("null handle"); return BCN_ERR_ARG; }
  return h->set_noise(sigma, seed, replica_offset);
}
int bcn_set_option(bcn_env_t h, const char* name, int value) {
  if (!h || !name) { bcn_set_error("null handle/name"); return BCN_ERR_ARG; }
  return h->set_option(name, value);
}
int bcn_set_slow_mode_bound(bcn_env_t h, int n, const double* cutoff, const double* bound) {
  if (!h || n < 0 || n > 2 || (n > 0 && (!cutoff || !bound))) { bcn_set_error("bcn_set_slow_mode_bound: null handle/arrays or n outside 0..2"); return BCN_ERR_ARG; }
  return h->set_slow_mode_bound(n, cutoff, bound);
}
int bcn_get_slow_mode_bound(bcn_env_t h, double* cutoff, double* bound) {
  if (!h || !cutoff || !bound) { bcn_set_error("null handle/arrays"); return BCN_ERR_ARG; }
  return h->get_slow_mode_bound(cutoff, bound);
}
int bcn_set_sched(bcn_env_t h, int mode, int grid, int q, int lpt_min_batch) {
  if (!h) { bcn_set_error("null handle"); return BCN_ERR_ARG; }
  if (mode < -1 || mode > 2 || grid < 0 || q < 0 || lpt_min_batch < 0) { bcn_set_error("bcn_set_sched: argument out of range"); return BCN_ERR_ARG; }
  return h->set_sched(mode, grid, q, lpt_min_batch);
}
const char* bcn_kernel_name(bcn_env_t h) { return h ? h->kernel_name() : ""; }
int bcn_destroy(bcn_env_t h) {
  if (!h) return BCN_OK;
  {
    DeviceGuard g(h->device);
    (void)hipDeviceSynchronize();
  }
  delete h;
  return BCN_OK;
}
const char* bcn_last_error(void) { return g_err; }
const char* bcn_version(void) { return "beacon_hip 0.1 (gfx950)"; }

}  // extern "C"
