/*
 * beacon_hip.h -- C ABI of libbeacon_hip.so: the MI355X-native batched stepper that
 * replaces the per-env solver hot path of jviquerat/beacon.
 *
 * The reference has no FFI: its boundary is a duck-typed Gym env class, one instance
 * per env.  Each entry point below stands in for one method of that
 * class, batched over B independent replicas, and cites the reference method it
 * replaces (file:line into /root/reference/beacon/).  Plain C types only; every
 * pointer named *_dev is a DEVICE pointer in the handle's dtype (BCN_F32 -> float,
 * BCN_F64 -> double) unless a type is spelled out; `stream` is a hipStream_t passed as
 * void* (NULL = default stream).  Calls on one handle must be serialised by the caller
 * (the reference env is not thread-safe either).  No call blocks on the GPU except
 * bcn_get_state/bcn_set_state with host pointers, and *_destroy.
 *
 * Precision (measured on MI355X in round 4 against the float64 reference / its C restatement; the tests assert <= 10 x these
 * figures, tests/test_gpu_parity.py: table F32, EPISODE_TOL, shkadov_tol):
 *   BCN_F64: rayleigh / mixing fields and observations within 1e-9 with the reference's Jacobi sweep counts (measured 3e-15 ..
 *            6e-15 over 400 timesteps); burgers, shkadov, sloshing fields bit-identical (rewards, being reductions, 1e-13).
 *   BCN_F32, one action step: rayleigh 128x64 (200 timesteps, ~94 sweeps each) u, v 4e-7, T, p 1.3e-6, observations 1.2e-6,
 *            reward (Nusselt number) 4e-6, every sweep count equal; mixing 100x100 (250 timesteps) u, v, C 2.5e-6, p 1.1e-5,
 *            observations 1.2e-6, reward 2e-8; burgers observations 4e-6 on average (one step of a 200-step episode at 5e-5);
 *            sloshing observations 8e-6; shkadov observations 1e-6 .. 2e-6, growing 1.4 x per action step (the film is chaotic).
 *   BCN_F32, a whole 100-step episode against BCN_F64: rayleigh observations 2.3e-5 for the median replica, 2.5e-3 for the
 *            worst one in its most sensitive transient (the flow amplifies rounding-level differences ~3000 x there: two
 *            float64 kernels drift apart with the same profile), 7e-5 at the end of the episode; mixing 1.8e-4 (median) /
 *            8.5e-4 (worst probe), rewards 1e-5; shkadov: trajectories decorrelate after ~40 steps, episode-return statistics
 *            over 64 replicas agree to 3e-4 sigma.
 *
 * Return value: 0 = BCN_OK, otherwise a BCN_ERR_* code; bcn_last_error() gives text.
 * Solver failures never exit the process (the reference does: rayleigh.py:221-224);
 * they are reported per replica in status_dev (BCN_ST_* bits).
 */
#ifndef BEACON_HIP_H
#define BEACON_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BCN_API __attribute__((visibility("default")))

/* Bumped whenever the size of a caller-provided buffer or a struct layout changes; bcn_api_version() returns the value
 * the loaded library was built with, so a caller compiled against an older header can refuse to run.
 *   4: bcn_get_counters writes BCN_COUNTER_WORDS = 4 words per replica (2 before); bcn_get_counters_n takes the count. */
#define BCN_API_VERSION 4
#define BCN_COUNTER_WORDS 4

typedef struct bcn_env_s* bcn_env_t;

enum { BCN_F32 = 0, BCN_F64 = 1 };
enum { BCN_OK = 0, BCN_ERR_ARG = 1, BCN_ERR_HIP = 2, BCN_ERR_UNSUPPORTED = 3 };
/* per-replica status word written by *_step */
enum { BCN_ST_OK = 0, BCN_ST_ITMAX = 1, BCN_ST_BLOWUP = 2,
       BCN_ST_PLAN = 4 /* diagnostic (bcn_set_option "verify_conv"): a Jacobi sweep the residual-evaluation plan skips passed the test */ };
/* env kinds (bcn_env_kind) */
enum { BCN_RAYLEIGH = 0, BCN_MIXING = 1, BCN_BURGERS = 2, BCN_SHKADOV = 3, BCN_SLOSHING = 4 };

/* ---- rayleigh: rayleigh/rayleigh.py -------------------------------------------------- */
/* ctor kwargs + the derived quantities of rayleigh.__init__ (rayleigh.py:20-56) */
typedef struct {
  int32_t nx, ny;                 /* int(50 L), int(50 H) */
  int32_t ndt_act, n_act;         /* 200, 100 */
  int32_t n_sgts, nx_sgts;        /* 10, nx // n_sgts */
  int32_t nx_obs_pts, ny_obs_pts; /* 4 int(L), 4 int(H) */
  int32_t nx_obs, ny_obs;         /* nx // nx_obs_pts, ny // ny_obs_pts */
  int32_t n_obs_steps;            /* 4 */
  int32_t itmax;                  /* 300000 (rayleigh.py:417) */
  double dx, dy, dt;
  double pr, ra, Tc, Th, C;       /* 0.71, 1e4, -0.5, 0.5, 0.75 */
  double tol;                     /* 1e-8 (rayleigh.py:414) */
} bcn_rayleigh_cfg;

/* replaces rayleigh.__init__ (rayleigh.py:20-86); batch = number of replicas on this device */
BCN_API int bcn_rayleigh_create(const bcn_rayleigh_cfg* cfg, int batch, int dtype, int device, bcn_env_t* out);
/* replaces rayleigh.reset (rayleigh.py:89-128).  init_fields_dev: [4][ny+2][nx+2] (u,v,p,T; x fastest)
 * shared by all replicas, or NULL for all-zero fields (init=False).  obs_dev[B][n_obs] may be NULL. */
BCN_API int bcn_rayleigh_reset(bcn_env_t h, const void* init_fields_dev, void* obs_dev, void* stream);
/* replaces rayleigh.step (rayleigh.py:138-157) incl. solve/get_obs/get_rwd (:160-275).
 * actions_dev[B][n_sgts] raw actions, NULL = repeat last (a=None, :162); on return
 * actions_norm_dev[B][n_sgts] (may be NULL) holds the conditioned actions the reference writes
 * back into the caller's list (:165-168).  obs_dev[B][n_obs], rwd_dev[B], done_dev/trunc_dev
 * uint8[B], status_dev int32[B], sweeps_dev int32[B][ndt_act] Jacobi sweeps per timestep (NULL ok). */
BCN_API int bcn_rayleigh_step(bcn_env_t h, const void* actions_dev, void* actions_norm_dev, void* obs_dev,
                      void* rwd_dev, uint8_t* done_dev, uint8_t* trunc_dev, int32_t* status_dev,
                      int32_t* sweeps_dev, void* stream);

/* ---- mixing: mixing/mixing.py -------------------------------------------------------- */
typedef struct {
  int32_t nx, ny;                 /* int(100 L), int(100 H) */
  int32_t ndt_act, n_act;         /* 250, 100 */
  int32_t nx_obs_pts, ny_obs_pts, nx_obs, ny_obs, n_obs_steps;
  int32_t itmax;                  /* 300000 (mixing.py:426) */
  int32_t i_min, i_max, j_min, j_max; /* initial patch, ARRAY indices (mixing.py:90-94) */
  double dx, dy, dt;
  double re, pe, u_max, C0, ref_c; /* ref_c = side^2/(L H) C0 (mixing.py:261) */
  double tol;                     /* 1e-4 (mixing.py:423) */
} bcn_mixing_cfg;

/* replaces mixing.__init__ (mixing.py:20-70) */
BCN_API int bcn_mixing_create(const bcn_mixing_cfg* cfg, int batch, int dtype, int device, bcn_env_t* out);
/* replaces mixing.reset / reset_fields (mixing.py:73-111) */
BCN_API int bcn_mixing_reset(bcn_env_t h, void* obs_dev, void* stream);
/* replaces mixing.step (mixing.py:114-135) incl. solve/get_control/get_obs/get_rwd (:138-264).
 * actions_dev int32[B] in {0,1,2,3} (anything else = walls at rest), NULL = repeat last. */
BCN_API int bcn_mixing_step(bcn_env_t h, const int32_t* actions_dev, void* obs_dev, void* rwd_dev,
                    uint8_t* done_dev, uint8_t* trunc_dev, int32_t* status_dev, int32_t* sweeps_dev,
                    void* stream);

/* ---- burgers: burgers/burgers.py ----------------------------------------------------- */
typedef struct {
  int32_t nx;                     /* 500 in the reference (burgers.py:26) */
  int32_t ndt_act, n_act;         /* 62, 200 */
  int32_t ctrl_pos, n_obs_pts;    /* 250, 5 */
  double dx, dt, amp, u_target;
} bcn_burgers_cfg;

/* replaces burgers.__init__ (burgers.py:21-65) */
BCN_API int bcn_burgers_create(const bcn_burgers_cfg* cfg, int batch, int dtype, int device, bcn_env_t* out);
/* replaces burgers.reset (burgers.py:68-94) */
BCN_API int bcn_burgers_reset(bcn_env_t h, void* obs_dev, void* stream);
/* replaces burgers.step (burgers.py:97-166).  actions_dev[B] (NULL = repeat last);
 * noise_dev[B]: the uniform(-sigma,sigma) inlet draw of this step (burgers.py:127), explicit. */
BCN_API int bcn_burgers_step(bcn_env_t h, const void* actions_dev, const void* noise_dev, void* obs_dev,
                     void* rwd_dev, uint8_t* done_dev, uint8_t* trunc_dev, int32_t* status_dev,
                     void* stream);

/* ---- shkadov: shkadov/shkadov.py ----------------------------------------------------- */
typedef struct {
  int32_t nx;                     /* int(5 (L0 + jet_space (n_jets+2))) */
  int32_t ndt_act, n_act;         /* 50, 400 */
  int32_t n_jets, jet_pos, jet_hw, jet_space; /* lattice units (shkadov.py:64-68) */
  int32_t l_obs, l_rwd, n_obs, obs_stride, n_interp; /* 50, 50, 10, 5, 20 */
  double dx, dt, delta, jet_amp, eps;
  double h_blow;                  /* 5 h_max = 25 (shkadov.py:176) */
  double blowup_rwd;              /* -1 */
} bcn_shkadov_cfg;

/* replaces shkadov.__init__ (shkadov.py:20-110) */
BCN_API int bcn_shkadov_create(const bcn_shkadov_cfg* cfg, int batch, int dtype, int device, bcn_env_t* out);
/* replaces shkadov.reset without its rand_init loop (shkadov.py:113-146): init_fields_dev [2][nx]
 * (h_init, q_init) shared by all replicas, NULL = flat film h=q=1 (reset_fields only). */
BCN_API int bcn_shkadov_reset(bcn_env_t h, const void* init_fields_dev, void* obs_dev, void* stream);
/* replaces shkadov.step (shkadov.py:161-264).  actions_dev[B][n_jets] (NULL = repeat last);
 * noise_dev[B][ndt_act]: inlet draws, one per timestep (shkadov.py:204). */
BCN_API int bcn_shkadov_step(bcn_env_t h, const void* actions_dev, const void* noise_dev, void* obs_dev,
                     void* rwd_dev, uint8_t* done_dev, uint8_t* trunc_dev, int32_t* status_dev,
                     void* stream);

/* ---- sloshing: sloshing/sloshing.py -------------------------------------------------- */
typedef struct {
  int32_t nx;                     /* int(80 L) */
  int32_t ndt_act, n_act, n_interp; /* 50, 200, 10 */
  double dx, dt, g, amp, alpha;
} bcn_sloshing_cfg;

/* replaces sloshing.__init__ (sloshing.py:20-89) */
BCN_API int bcn_sloshing_create(const bcn_sloshing_cfg* cfg, int batch, int dtype, int device, bcn_env_t* out);
/* replaces sloshing.reset (sloshing.py:92-124): init_fields_dev [2][nx+2] (h_init, q_init) or NULL */
BCN_API int bcn_sloshing_reset(bcn_env_t h, const void* init_fields_dev, void* obs_dev, void* stream);
/* replaces sloshing.step (sloshing.py:141-244).  actions_dev[B] (NULL = repeat last) */
BCN_API int bcn_sloshing_step(bcn_env_t h, const void* actions_dev, void* obs_dev, void* rwd_dev,
                      uint8_t* done_dev, uint8_t* trunc_dev, int32_t* status_dev, void* stream);

/* ---- common -------------------------------------------------------------------------- */
BCN_API int bcn_env_kind(bcn_env_t h);
BCN_API int bcn_batch(bcn_env_t h);
BCN_API int bcn_dtype(bcn_env_t h);
BCN_API int bcn_n_obs(bcn_env_t h);       /* observation length per replica */
BCN_API int bcn_n_act(bcn_env_t h);       /* action length per replica */
BCN_API int bcn_ndt_act(bcn_env_t h);     /* timesteps per action step: rows of sweeps_dev [B][ndt_act] and of shkadov's noise_dev */
BCN_API int bcn_device(bcn_env_t h);      /* HIP device index the handle was created on (every *_dev pointer must live there) */
/* Solver state of all replicas, the equivalent of the env's field attributes (and of
 * dump()/load(), rayleigh.py:344-362): elements per replica, then copy out / in.  Layout per
 * replica: rayleigh/mixing [4][ny+2][nx+2] = u,v,p,S; burgers [3][nx] = u,up,upp;
 * shkadov [4][nx] = h,q,rhsh,rhsq; sloshing [4][nx+2] = h,q,rhsh,rhsq.  `buf` may be a host or a
 * device pointer (is_device); host copies synchronise the stream. */
BCN_API size_t bcn_state_elems(bcn_env_t h);
BCN_API int bcn_get_state(bcn_env_t h, void* buf, int is_device, void* stream);
BCN_API int bcn_set_state(bcn_env_t h, const void* buf, int is_device, void* stream);
/* Replica mask for the calls that follow: *_reset and *_step skip every replica b with mask_dev[b] == 0
 * (state, stp and that replica's rows of the output buffers stay untouched); NULL = all replicas.  The
 * pointer is read at launch time of each later call, so it must stay valid.  Used for per-replica
 * resets at episode end (the reference's trainer calls reset() on one env) and for the per-env random
 * number of uncontrolled warm-up steps of shkadov.reset (shkadov.py:119-123). */
BCN_API int bcn_set_mask(bcn_env_t h, const uint8_t* mask_dev);
/* episode counter `stp` of every replica (rayleigh.py:126,155): int32[B] */
BCN_API int bcn_get_stp(bcn_env_t h, int32_t* buf_host, void* stream);
BCN_API int bcn_set_stp(bcn_env_t h, const int32_t* buf_host, void* stream);
/* Which kernel variant *_step uses: 0 = generic (any grid, fields in HBM/L2, Jacobi in LDS),
 * 1 = register-resident CDNA4 path where the grid has one (default there; falls back to 0 otherwise).  Built into the
 * library: rayleigh 128x64 f32/f64, 50x50 f32/f64, 100x50 / 150x50 / 200x50 f32 (one row per lane: ns2d_fast_impl.h);
 * rayleigh and mixing 100x100 f32/f64 (two rows per lane: ns2d_fast2_impl.h).  Every other grid gets its kernel through
 * bcn_set_fast_plugin (beacon_amd/jit.py compiles it on demand): one row per lane for rayleigh with ny <= 64, two rows per
 * lane for 64 < ny <= 128, and for everything else up to ny = 256 (tall grids, grids wider than the strips, mixing below
 * ny = 64) the hybrid of ns2d_fast4_impl.h (Poisson solve in registers, fields in HBM/L2).  Beyond that: variant 0 only.
 * Results of the two variants agree to rounding (float64: 1e-9).  Returns the variant actually selected. */
BCN_API int bcn_set_variant(bcn_env_t h, int variant);
/* Measurement aid (no reference counterpart), uint64[B][4] on the host, per replica, of the last *_step (all chunks):
 *   [0] shader-clock cycles inside the Jacobi loop (rayleigh.py:419-454 / mixing.py:428-463), [1] in the whole replica,
 *   [2] solves whose LANDING -- the first evaluation of the residual behind sweeps the extrapolating plan (conv_plan 2 / 3) had
 *       skipped -- did not verify the skip: under plan 3 the residual was not above BCN_CONV_GUARD * tol there (so a skipped
 *       sweep may have passed the test); under plan 2 the landing itself passed,
 *   [3] timesteps (rayleigh) / solves (two-rows-per-lane and tall-grid kernels) that were repeated: a speculative opening whose
 *       landing did not verify it, or conv_plan 3 repeating an unverified solve under the proven plan.
 * Zeros for kernels that do not count (generic 2D kernel, 1D envs).
 * Under conv_plan 3 every stop sweep is the reference's BY PROOF (see "conv_plan"): zeros in [2] mean that no solve needed the
 * repeat, not that something went unnoticed.
 * buf_host must hold batch * BCN_COUNTER_WORDS words; bcn_get_counters_n writes `words_per_replica` words per replica instead
 * (the first min(words, BCN_COUNTER_WORDS) counters, zeros beyond) for callers built against another header. */
BCN_API int bcn_get_counters(bcn_env_t h, uint64_t* buf_host, void* stream);
BCN_API int bcn_get_counters_n(bcn_env_t h, uint64_t* buf_host, int words_per_replica, void* stream);
BCN_API int bcn_api_version(void);
/* Register-resident kernel for a grid that is not built into the library (up to ny = 256; above ny = 128 and for grids wider
 * than the all-in-registers kernels' strips only the Poisson solve is register-resident: csrc/ns2d_fast4_impl.h).  The reference takes any L, H
 * (rayleigh.py:20-27: nx = 50 L, ny = 50 H; mixing.py:20-28: 100 L, 100 H); csrc/jit/ns2d_jit.hip is compiled for ONE
 * grid into its own shared object (beacon_amd/jit.py does so on demand and caches it) whose
 *   int bcn_jit_launch(const void* step_args, int batch, void* stream)
 * is passed here together with bcn_jit_scratch_elems().  Selects variant 1; launch_fn = NULL restores the built-in
 * choice.  The plugin must outlive the handle. */
BCN_API int bcn_set_fast_plugin(bcn_env_t h, void* launch_fn, size_t scratch_elems);
/* Inlet noise on the device (burgers, shkadov).  The reference draws np.random.uniform(-sigma, sigma, 1) from numpy's global
 * stream -- once per action step (burgers.py:127), once per timestep (shkadov.py:204) -- and the *_step entry points take those
 * draws as noise_dev, so that a caller can reproduce the reference's stream.  A trainer that only needs noise of that law leaves
 * noise_dev NULL after this call: the step kernel then draws uniform(-sigma, sigma) itself (Philox4x32-10 keyed by `seed`, counter =
 * (replica_offset + replica index, the replica's own count of such steps, timestep)) -- no extra launch, no host work, fresh values
 * when a captured graph replays.  sigma = 0 (the default) restores "NULL = no noise".  replica_offset: global index of this handle's
 * replica 0 (sharded batches).  Resets the replicas' draw counters.  BCN_ERR_ARG for envs without inlet noise. */
BCN_API int bcn_set_noise(bcn_env_t h, double sigma, uint64_t seed, int64_t replica_offset);
/* Solver options of the 2D envs (no reference counterpart), by name:
 *   "conv_plan"   which Jacobi sweeps evaluate the residual err = sum((phi - phin)^2) of rayleigh.py:448-449 / mixing.py:457-458
 *                 in the register-resident kernels.  Every plan returns the reference's stop sweep -- the FIRST sweep with
 *                 err <= tol -- and differs only in how it knows that the sweeps it does not evaluate could not pass:
 *                 0 = every sweep evaluated, as the reference does;
 *                 1 = sweeps skipped while a PROVEN lower bound of err stays above tol: the increments obey d' = J d with J
 *                     symmetric, so the plain norm |d|^2 <= err is log-convex in the sweep count and its measured decay bounds
 *                     every later one (the default of BCN_F64 until round 5; what plan 3 repeats an unverified solve under);
 *                 3 = the decay of err itself extrapolated, landing where it is still expected above 1.035 tol, and every
 *                     landing -- the first evaluation behind skipped sweeps, the speculative opening's included -- VERIFIED:
 *                     err can grow from a sweep to a later one by at most max_m || W^1/2 J^m W^-1/2 ||^2 = 1.030 (W = I + G:
 *                     the ghost copies the reference's sum counts; scripts/weighted_norm_bound.py, every grid of the
 *                     reference's constructor space), so had a skipped sweep passed, the landing would find err <= 1.030 tol;
 *                     a landing above BCN_CONV_GUARD = 1.035 tol therefore proves that none did, and a landing below it is
 *                     counted ([2]) and the solve repeated under plan 1 (the default of both precisions; grids with a side below 48
 *                     cells, where the bound is larger, take plans 0 / 1 only).  With the grid's slow-mode constants in force
 *                     (bcn_set_slow_mode_bound below; built in for the default grids) the landing threshold drops from 1.035 tol
 *                     to a fraction of a percent above tol late in a solve -- still a proof, and far fewer evaluations.
 *                     What "proof" covers: the recurrence d' = J d in EXACT arithmetic, which BCN_F64 follows to 1e-16.  BCN_F32
 *                     adds rounding of ~eps |phi| per cell and sweep to the increments (0.05 - 0.3 % of tol = 1e-8 on 8192 cells),
 *                     the size of the slow-mode threshold's distance from tol; the kernels put a 0.1 % margin on that threshold
 *                     and the float32 default is VERIFIED EMPIRICALLY on top (option "verify_conv": every sweep evaluated next to
 *                     the plan over whole episodes of the bench workload, no BCN_ST_PLAN: tests/test_gpu_parity.py), not proven;
 *                     a caller who wants the float32 proof unconditional clears the slow-mode constants
 *                     (bcn_set_slow_mode_bound(h, 0, ...)): BCN_CONV_GUARD's 3.5 % dwarf the rounding;
 *                 2 = plan 3's extrapolation WITHOUT the verification (round 2's rule: it notices a landing only when the
 *                     landing itself passes, and then only counts it): kept for measurements, never a default
 *   "plan_overshoot" 0..64, TEST HOOK: lengthens every skip of plans 2 / 3 by that many sweeps, so that landings fall behind
 *                 the stop sweep (tests/test_gpu_parity.py: the adversarial right-hand side)
 *   "verify_conv" 1 = evaluate every sweep anyway and raise BCN_ST_PLAN if a sweep the plan skips passes the test
 *   "spec_start"  0..17: open a solve with unevaluated double sweeps up to spec_start/8 of the previous timestep's sweep
 *                 count (1..16), or -- 17, the default -- up to 15/16 of it minus 1.25 times the stretch in front of the stop in
 *                 which the residual is already below the landing guard (log2(1.035) / the previous solve's decay per sweep);
 *                 the landing behind them is verified like any other (plan 3: above 1.035 tol, else the solve is
 *                 repeated without the guess).  BCN_F32 rayleigh only; ignored by BCN_F64 handles, off for mixing
 *   "transport_iter" 0..64 (mixing, BCN_F32, 64 < ny <= 128): the ordered part of the scalar transport -- mixing.py:478-497 sweeps the
 *                 array in place, so a cell reads the NEW values of its west and south neighbours: S' = A + aW S'(i-1,j) + aS S'(i,j-1),
 *                 a lower-triangular system -- as the Neumann series sum_m L^m A, one parallel pass of every wave per term, M terms with
 *                 rho^(M+1) <= 2^-27, rho = max(|aW| + |aS|) measured in every timestep (0.2 at the reference's u_max: M = 12; 7e-9 of
 *                 a scalar in [0, 1], below the rounding of the sweep itself).  The option is the largest M allowed (default 24); where
 *                 the measured rho needs more, and with 0, the ordered sweep of the reference runs (one wave, nx + ny/2 dependent steps).
 *                 BCN_F64 always runs the ordered sweep
 *   "sched_tail"  short chunks that end a step of the ticket scheduler (0 = default 6; see bcn_set_sched)
 *   "generic_threads" 256 / 1024: workgroup size of the generic 2D kernel (0 = chosen by grid size)
 *   "cells_per_thread" (1D envs) 1, 2, 4, 8 cells per thread (0 = chosen from grid and batch); "one_wave" (1D envs) 0 / 1:
 *                 grids up to 512 cells as one wave per replica with DPP halos (default 1)
 * No environment variable changes any of these (round 5: the library reads none).
 * Returns BCN_ERR_ARG for unknown names. */
BCN_API int bcn_set_option(bcn_env_t h, const char* name, int value);
/* Slow-mode landing guard of conv_plan 3 (no reference counterpart; beacon_amd/stoprule.py has the derivation).  Within the span of
 * the Jacobi matrix's eigenvectors with |lambda| >= cutoff[k], the reference norm can grow from a sweep to any later one by at most
 * bound[k] (>= 1: 1.0002 / 1.0057 for cutoffs 0.9 / 0.8 on the 128x64 grid, against 1.030 over all modes), and what lies outside that
 * span has decayed to cutoff^(j-1) |d_1| by sweep j.  With the constants set, a landing behind the last evaluated sweep i is verified
 * when it finds err > min(BCN_CONV_GUARD, min_k bound[k] (1 + 2 sqrt(3 |d_1|^2 / tol) cutoff[k]^i)^2 (1 + 0.001)) tol -- a fraction of
 * a percent above tol late in a solve instead of 3.5 %, i.e. a dozen sweeps fewer that must be evaluated one by one.  The constants
 * are properties of (nx, ny, boundary kind, cx): built in for the reference's default grids (rayleigh 50x50, mixing 100x100) and for
 * 128x64; for any other grid the host computes them (beacon_amd/stoprule.py, cached) and passes n <= 2 pairs here; n = 0 clears them
 * (the guard is then BCN_CONV_GUARD alone).  Used by the one-row and two-rows-per-lane kernels (ny <= 128); the hybrid kernel
 * (ny > 128) keeps BCN_CONV_GUARD.  bcn_get_slow_mode_bound returns the number of pairs in force and writes them (arrays of 2). */
BCN_API int bcn_set_slow_mode_bound(bcn_env_t h, int n, const double* cutoff, const double* bound);
BCN_API int bcn_get_slow_mode_bound(bcn_env_t h, double* cutoff, double* bound);
/* Work scheduling of the register-resident 2D kernels when replicas outnumber the CUs (no reference
 * counterpart: the reference steps one env per process, rayleigh.py:138-157).  mode: -1 = default
 * (2), 0 = one workgroup per replica in one launch, 1 = two launches with the
 * replicas re-ordered longest-first, 2 = persistent workgroups drawing (chunk of q timesteps, replica)
 * tickets; grid = persistent workgroups (0 = one per CU); q = timesteps per chunk (0 = kernel default);
 * lpt_min_batch = smallest batch that mode 1 splits (0 = CUs + 1).  Results do not depend on the mode. */
BCN_API int bcn_set_sched(bcn_env_t h, int mode, int grid, int q, int lpt_min_batch);
/* name of the kernel the last *_step dispatched, e.g. "ns2d_fast_sched" (before the first step: the
 * variant's plain kernel); for profiles */
BCN_API const char* bcn_kernel_name(bcn_env_t h);
BCN_API int bcn_destroy(bcn_env_t h);
BCN_API const char* bcn_last_error(void);
BCN_API const char* bcn_version(void);

#ifdef __cplusplus
}
#endif
#endif /* BEACON_HIP_H */
