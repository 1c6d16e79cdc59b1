// ns2d_sched.h -- ticketed chunk scheduler shared by the register-resident kernels
// (ns2d_fast.hip, ns2d_fast2.hip).
#pragma once
#include <stdlib.h>

#include "ns2d.h"

// With more replicas than CUs a replica is no longer tied to one workgroup: the step is cut into
// chunks of A.sched_q timesteps and persistent workgroups (one per CU) draw (chunk, replica) units
// from a global ticket counter in chunk-major order, so every CU stays busy until the slowest
// replica's chain of chunks ends (makespan ~ max(critical path, mean) instead of the sum of whichever
// two replicas a CU happened to get).  A replica's state moves between CUs through HBM; the hand-off
// follows the agent-scope release/acquire recipe of the CDNA programming guide (Guideline 16):
//   producer: stores -> workgroup barrier -> lane 0: fence(release, agent); s_waitcnt vmcnt(0);
//             relaxed agent store progress[r] = c+1
//   consumer: lane 0 polls progress[r] (relaxed, agent, s_sleep) -> fence(acquire, agent);
//             s_waitcnt vmcnt(0) -> workgroup barrier -> plain loads.
// Tickets are drawn in order, so when unit (c, r) is drawn unit (c-1, r) has already been drawn by
// a workgroup that never waits on a later ticket: every wait is finite whatever the residency.
// Spins are bounded anyway: on timeout the abort word is set, every workgroup drains, and the
// affected replicas report BCN_ST_ITMAX.
struct SchedCtl {
  unsigned int ticket;
  unsigned int abort;
  unsigned int pad[14];
  unsigned int progress[1];   // [B]
};

// The persistent workgroup's loop.  `s_words`: two LDS words the unit does not touch;
// `unit(b, it0, it1, first_chunk, last_chunk)` runs timesteps [it0, it1) of replica b (state HBM ->
// chip -> HBM) and is called by every thread of the workgroup.
template <typename real, typename Unit>
__device__ __forceinline__ void ns2d_sched_loop(const NS2DArgs<real>& A, SchedCtl* ctl, int batch, int nchunk,
                                                unsigned int* s_words, Unit unit) {
  unsigned int& s_ticket = s_words[0];
  unsigned int& s_ok = s_words[1];
  const unsigned int total = (unsigned int)batch * (unsigned int)nchunk;
  for (;;) {
    if (threadIdx.x == 0) s_ticket = atomicAdd(&ctl->ticket, 1u);
    __syncthreads();
    const unsigned int t = s_ticket;
    if (t >= total) break;
    const int c = (int)(t / (unsigned int)batch), b = (int)(t % (unsigned int)batch);
    const bool skip = A.mask && !A.mask[b];
    if (threadIdx.x == 0) {
      unsigned int ok = 1;
      if (!skip && c > 0) {
        unsigned int spins = 0;
        while (__hip_atomic_load(&ctl->progress[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned int)c) {
          __builtin_amdgcn_s_sleep(32);
          if (++spins > (1u << 24) || __hip_atomic_load(&ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            __hip_atomic_store(&ctl->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = 0;
            break;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      s_ok = ok;
    }
    __syncthreads();
    if (!skip) {
      if (s_ok) {
        // chunk c: the first sched_nbig chunks are twice as long (half the hand-offs through HBM where the order of the
        // units cannot matter yet), the last ones short (the step ends when the slowest replica's last chunk does)
        const int it0 = c < A.sched_nbig ? c * 2 * A.sched_q : (c + A.sched_nbig) * A.sched_q;
        const int it1 = (c == nchunk - 1) ? A.ndt_act : it0 + (c < A.sched_nbig ? 2 : 1) * A.sched_q;
        unit(b, it0, it1, c == 0, c == nchunk - 1);
      } else if (threadIdx.x == 0) {
        A.status[b] = BCN_ST_ITMAX;   // handle-owned when the caller passed none: never NULL here
      }
      __syncthreads();   // every wave's stores are issued and waited for (barrier implies vmcnt(0))
      if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&ctl->progress[b], (unsigned int)(c + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __syncthreads();     // s_ticket / s_ok are rewritten next trip
  }
}

// "first launch of this kernel on the current device": the dynamic-LDS attribute is per device, and a process may
// hold handles on several devices (one process per GPU is the deployment, not a requirement of the ABI)
inline bool ns2d_first_on_device(unsigned long long& seen) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (seen & bit) return false;
  seen |= bit;
  return true;
}

// host side, per handle (bcn_set_sched -> NS2DArgs::sched_mode / sched_grid / sched_q_user / lpt_min_batch; -1 / 0 = default):
// mode 0 plain launch, 1 two-launch LPT split (ns2d_fast only), 2 ticketed chunks (default); grid = persistent workgroups
// (default: one per CU of the handle's device); q = timesteps per chunk.  No environment variable changes any of it.
// chunks of one step: nbig long ones (2 q timesteps) followed by short ones (q; the last takes the remainder), the short
// tail covering at least the last `tail` * q timesteps (bcn_set_option "sched_tail", default 6; a tail >= ndt / q gives
// uniform chunks).
inline void ns2d_sched_chunks(int ndt, int q, int tail_user, int* nbig, int* nchunk) {
  const int tail = tail_user > 0 ? tail_user : 6;
  int nb = (ndt - tail * q) / (2 * q);
  if (nb < 0) nb = 0;
  int rem = ndt - nb * 2 * q;
  int ns = rem / q;
  if (ns < 1) { ns = 1; }
  *nbig = nb;
  *nchunk = nb + ns;
}
struct SchedParams {
  int mode, grid, q;
  bool q_set;   // chunk length given (environment or handle): overrides the per-kernel default
  int lpt_min_batch;
};
inline int ns2d_cu_count() {   // of the current device (one handle per device; a process may hold several)
  static int ncu[64] = {0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  int& n = ncu[dev & 63];
  if (n <= 0) {
    n = 256;
    (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
  }
  return n;
}
template <typename real>
inline SchedParams ns2d_sched_params(const NS2DArgs<real>& a) {
  SchedParams p;
  p.mode = a.sched_mode >= 0 ? a.sched_mode : 2;
  const int ncu = ns2d_cu_count();
  p.grid = a.sched_grid > 0 ? a.sched_grid : ncu;
  const int q = a.sched_q_user;
  p.q_set = q > 0;
  p.q = q > 0 ? q : 10;
  p.lpt_min_batch = a.lpt_min_batch > 0 ? a.lpt_min_batch : ncu + 1;
  return p;
}
