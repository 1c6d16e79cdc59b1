/* Plain-C consumer of libbeacon_hip.so (no Python, no torch): burgers-v0, 4 replicas, float64, 3 action steps.
 * Prints "obs <step> <replica> v0 .. v4" and "rwd <step> <replica> r" lines that tests/test_gpu_parity.py compares
 * with the oracle.  Build: gcc -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude burgers_smoke.c
 *        -Lbeacon_amd -lbeacon_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,... */
#include <stdio.h>
#include <stdlib.h>
#include <hip/hip_runtime_api.h>

#include "beacon_hip.h"

#define CHECK(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, bcn_last_error()); return 1; } } while (0)
#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(void) {
  enum { B = 4, NOBS = 5, STEPS = 3 };
  bcn_burgers_cfg c;
  c.nx = 500; c.ndt_act = 62; c.n_act = 200; c.ctrl_pos = 250; c.n_obs_pts = NOBS;
  c.dx = 2.0 / 500; c.dt = 0.2 * c.dx; c.amp = 10.0; c.u_target = 0.5;
  bcn_env_t h = NULL;
  CHECK(bcn_burgers_create(&c, B, BCN_F64, 0, &h));
  if (bcn_batch(h) != B || bcn_n_obs(h) != NOBS || bcn_dtype(h) != BCN_F64) { fprintf(stderr, "handle query mismatch\n"); return 1; }
  double *obs, *rwd, *act, *noise;
  uint8_t *done, *trunc;
  int32_t* status;
  HIP(hipMalloc((void**)&obs, B * NOBS * sizeof(double)));
  HIP(hipMalloc((void**)&rwd, B * sizeof(double)));
  HIP(hipMalloc((void**)&act, B * sizeof(double)));
  HIP(hipMalloc((void**)&noise, B * sizeof(double)));
  HIP(hipMalloc((void**)&done, B));
  HIP(hipMalloc((void**)&trunc, B));
  HIP(hipMalloc((void**)&status, B * sizeof(int32_t)));
  CHECK(bcn_burgers_reset(h, obs, NULL));
  for (int s = 0; s < STEPS; s++) {
    double ha[B], hn[B], ho[B * NOBS], hr[B];
    for (int b = 0; b < B; b++) { ha[b] = 0.25 * (b - 1.5) * (s + 1); hn[b] = 0.02 * (b + 1) - 0.01 * s; }
    HIP(hipMemcpy(act, ha, sizeof(ha), hipMemcpyHostToDevice));
    HIP(hipMemcpy(noise, hn, sizeof(hn), hipMemcpyHostToDevice));
    CHECK(bcn_burgers_step(h, act, noise, obs, rwd, done, trunc, status, NULL));
    HIP(hipDeviceSynchronize());
    HIP(hipMemcpy(ho, obs, sizeof(ho), hipMemcpyDeviceToHost));
    HIP(hipMemcpy(hr, rwd, sizeof(hr), hipMemcpyDeviceToHost));
    for (int b = 0; b < B; b++) {
      printf("obs %d %d", s, b);
      for (int k = 0; k < NOBS; k++) printf(" %.17g", ho[b * NOBS + k]);
      printf("\nrwd %d %d %.17g\n", s, b, hr[b]);
    }
  }
  CHECK(bcn_destroy(h));
  printf("kernel ok\n");
  return 0;
}
