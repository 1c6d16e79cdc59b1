// beacon_torch.cpp -- the thin PyTorch-ROCm extension over the C ABI (include/beacon_hip.h): torch.library ops
//   beacon::{rayleigh,mixing,burgers,shkadov,sloshing}_{step,reset}(int handle, Tensor ...) -> ()
// Each op is ONE dispatcher call that takes device tensors, reads torch's current HIP stream in C++ and forwards to the
// bcn_* entry point of libbeacon_hip.so -- no ctypes marshalling, no Python-side stream query (what the per-call host cost of
// the ctypes binding was made of: scripts/host_cost.py), and an op CUDA-graph capture and fake-tensor tracing can see (Meta kernels below).  The ops
// mutate their output tensors in place and return nothing; the size, dtype and device of EVERY tensor are checked here (a short buffer is an error, not an out-of-bounds write), the values by the library.
// Host code only: compiled with g++ against the torch headers (beacon_amd/torch_ext.py), linked to libbeacon_hip.so.
//
// The boundary each op stands in for is the reference's env method (rayleigh.py:89-157, mixing.py:73-135, burgers.py:68-117,
// shkadov.py:113-185, sloshing.py:92-166): reset() -> obs, step(a) -> (obs, rwd, done, trunc).
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include "../../../include/beacon_hip.h"

namespace {

using at::Tensor;
using OptT = const std::optional<Tensor>&;

inline bcn_env_t H(int64_t h) { return reinterpret_cast<bcn_env_t>(static_cast<intptr_t>(h)); }

inline void* stream_of(const Tensor& t) {
  return static_cast<void*>(c10::hip::getCurrentHIPStream(t.device().index()).stream());
}

inline void check(int rc, const char* what) {
  TORCH_CHECK(rc == BCN_OK, "libbeacon_hip: ", what, " failed with error ", rc, ": ", bcn_last_error());
}

// device pointer of a contiguous tensor on the handle's device, of the handle's dtype where `real`
inline void on_device(const Tensor& t, bcn_env_t h, const char* name) {
  TORCH_CHECK(t.is_cuda() && t.is_contiguous(), name, ": contiguous device tensor expected");
  TORCH_CHECK(t.device().index() == bcn_device(h), name, ": lives on device ", (int)t.device().index(), " but the handle was created on device ",
              bcn_device(h));
}
inline void* dp(const Tensor& t, bcn_env_t h, bool real, const char* name) {
  on_device(t, h, name);
  if (real) {
    const auto want = bcn_dtype(h) == BCN_F64 ? at::kDouble : at::kFloat;
    TORCH_CHECK(t.scalar_type() == want, name, ": dtype ", t.scalar_type(), " but the handle computes in ", want);
  }
  return t.data_ptr();
}
inline void* dpo(OptT t, bcn_env_t h, bool real, const char* name) { return t.has_value() ? dp(*t, h, real, name) : nullptr; }
inline void rows(const Tensor& t, bcn_env_t h, int64_t per, const char* name) {
  TORCH_CHECK(t.numel() == (int64_t)bcn_batch(h) * per, name, ": ", t.numel(), " elements, expected batch ", bcn_batch(h), " x ", per);
}
// per-replica output words: `per` elements per replica, on the handle's device
inline uint8_t* u8(const Tensor& t, bcn_env_t h, int64_t per, const char* name) {
  on_device(t, h, name);
  TORCH_CHECK(t.scalar_type() == at::kByte, name, ": uint8 tensor expected");
  rows(t, h, per, name);
  return t.data_ptr<uint8_t>();
}
inline int32_t* i32(const Tensor& t, bcn_env_t h, int64_t per, const char* name) {
  on_device(t, h, name);
  TORCH_CHECK(t.scalar_type() == at::kInt, name, ": int32 tensor expected");
  rows(t, h, per, name);
  return t.data_ptr<int32_t>();
}
// optional real input of `per` elements per replica (actions, noise)
inline void* dpr(OptT t, bcn_env_t h, int64_t per, const char* name) {
  if (!t.has_value()) return nullptr;
  rows(*t, h, per, name);
  return dp(*t, h, true, name);
}

// ---- rayleigh (rayleigh.py:89-157) -------------------------------------------------------------------------------------
void rayleigh_reset(int64_t h_, OptT init_fields, const Tensor& obs) {
  bcn_env_t h = H(h_);
  rows(obs, h, bcn_n_obs(h), "obs");
  check(bcn_rayleigh_reset(h, dpo(init_fields, h, true, "init_fields"), dp(obs, h, true, "obs"), stream_of(obs)), "bcn_rayleigh_reset");
}
void rayleigh_step(int64_t h_, OptT actions, const Tensor& actions_norm, const Tensor& obs, const Tensor& rwd, const Tensor& done,
                   const Tensor& trunc, const Tensor& status, const Tensor& sweeps) {
  bcn_env_t h = H(h_);
  rows(obs, h, bcn_n_obs(h), "obs");
  rows(rwd, h, 1, "rwd");
  rows(actions_norm, h, bcn_n_act(h), "actions_norm");
  check(bcn_rayleigh_step(h, dpr(actions, h, bcn_n_act(h), "actions"), dp(actions_norm, h, true, "actions_norm"), dp(obs, h, true, "obs"),
                          dp(rwd, h, true, "rwd"), u8(done, h, 1, "done"), u8(trunc, h, 1, "trunc"), i32(status, h, 1, "status"), i32(sweeps, h, bcn_ndt_act(h), "sweeps"),
                          stream_of(obs)),
        "bcn_rayleigh_step");
}

// ---- mixing (mixing.py:73-135) -----------------------------------------------------------------------------------------
void mixing_reset(int64_t h_, const Tensor& obs) {
  bcn_env_t h = H(h_);
  rows(obs, h, bcn_n_obs(h), "obs");
  check(bcn_mixing_reset(h, dp(obs, h, true, "obs"), stream_of(obs)), "bcn_mixing_reset");
}
void mixing_step(int64_t h_, OptT actions, const Tensor& obs, const Tensor& rwd, const Tensor& done, const Tensor& trunc,
                 const Tensor& status, const Tensor& sweeps) {
  bcn_env_t h = H(h_);
  rows(obs, h, bcn_n_obs(h), "obs");
  rows(rwd, h, 1, "rwd");
  const int32_t* a = actions.has_value() ? i32(*actions, h, 1, "actions") : nullptr;
  check(bcn_mixing_step(h, a, dp(obs, h, true, "obs"), dp(rwd, h, true, "rwd"), u8(done, h, 1, "done"), u8(trunc, h, 1, "trunc"),
                        i32(status, h, 1, "status"), i32(sweeps, h, bcn_ndt_act(h), "sweeps"), stream_of(obs)),
        "bcn_mixing_step");
}

// ---- burgers (burgers.py:68-117) ---------------------------------------------------------------------------------------
void burgers_reset(int64_t h_, const Tensor& obs) {
  bcn_env_t h = H(h_);
  rows(obs, h, bcn_n_obs(h), "obs");
  check(bcn_burgers_reset(h, dp(obs, h, true, "obs"), stream_of(obs)), "bcn_burgers_reset");
}
void burgers_step(int64_t h_, OptT actions, OptT noise, const Tensor& obs, const Tensor& rwd, const Tensor& done, const Tensor& trunc,
                  const Tensor& status) {
  bcn_env_t h = H(h_);
  rows(obs, h, bcn_n_obs(h), "obs");
  rows(rwd, h, 1, "rwd");
  check(bcn_burgers_step(h, dpr(actions, h, 1, "actions"), dpr(noise, h, 1, "noise"), dp(obs, h, true, "obs"),
                         dp(rwd, h, true, "rwd"), u8(done, h, 1, "done"), u8(trunc, h, 1, "trunc"), i32(status, h, 1, "status"), stream_of(obs)),
        "bcn_burgers_step");
}

// ---- shkadov (shkadov.py:113-185) --------------------------------------------------------------------------------------
void shkadov_reset(int64_t h_, OptT init_fields, const Tensor& obs) {
  bcn_env_t h = H(h_);
  rows(obs, h, bcn_n_obs(h), "obs");
  check(bcn_shkadov_reset(h, dpo(init_fields, h, true, "init_fields"), dp(obs, h, true, "obs"), stream_of(obs)), "bcn_shkadov_reset");
}
void shkadov_step(int64_t h_, OptT actions, OptT noise, const Tensor& obs, const Tensor& rwd, const Tensor& done, const Tensor& trunc,
                  const Tensor& status) {
  bcn_env_t h = H(h_);
  rows(obs, h, bcn_n_obs(h), "obs");
  rows(rwd, h, 1, "rwd");
  check(bcn_shkadov_step(h, dpr(actions, h, bcn_n_act(h), "actions"), dpr(noise, h, bcn_ndt_act(h), "noise"), dp(obs, h, true, "obs"),
                         dp(rwd, h, true, "rwd"), u8(done, h, 1, "done"), u8(trunc, h, 1, "trunc"), i32(status, h, 1, "status"), stream_of(obs)),
        "bcn_shkadov_step");
}

// ---- sloshing (sloshing.py:92-166) -------------------------------------------------------------------------------------
void sloshing_reset(int64_t h_, OptT init_fields, const Tensor& obs) {
  bcn_env_t h = H(h_);
  rows(obs, h, bcn_n_obs(h), "obs");
  check(bcn_sloshing_reset(h, dpo(init_fields, h, true, "init_fields"), dp(obs, h, true, "obs"), stream_of(obs)), "bcn_sloshing_reset");
}
void sloshing_step(int64_t h_, OptT actions, const Tensor& obs, const Tensor& rwd, const Tensor& done, const Tensor& trunc,
                   const Tensor& status) {
  bcn_env_t h = H(h_);
  rows(obs, h, bcn_n_obs(h), "obs");
  rows(rwd, h, 1, "rwd");
  check(bcn_sloshing_step(h, dpr(actions, h, 1, "actions"), dp(obs, h, true, "obs"), dp(rwd, h, true, "rwd"), u8(done, h, 1, "done"),
                          u8(trunc, h, 1, "trunc"), i32(status, h, 1, "status"), stream_of(obs)),
        "bcn_sloshing_step");
}

// Meta (fake-tensor) kernels: the ops return nothing and their outputs keep their shapes, so tracing needs no more than this.
void reset2_meta(int64_t, const Tensor&) {}
void reset3_meta(int64_t, OptT, const Tensor&) {}
void rayleigh_step_meta(int64_t, OptT, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&) {}
void mixing_step_meta(int64_t, OptT, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&) {}
void noisy_step_meta(int64_t, OptT, OptT, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&) {}
void sloshing_step_meta(int64_t, OptT, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&) {}

}  // namespace

// Outputs are written in place (annotated (x!)); the ops return nothing.  `handle` is the bcn_env_t of bcn_*_create as an
// integer.  Registered for the CUDA dispatch key, which is what ROCm tensors carry.
TORCH_LIBRARY(beacon, m) {
  m.def("rayleigh_reset(int handle, Tensor? init_fields, Tensor(a!) obs) -> ()");
  m.def("rayleigh_step(int handle, Tensor? actions, Tensor(a!) actions_norm, Tensor(b!) obs, Tensor(c!) rwd, Tensor(d!) done, "
        "Tensor(e!) trunc, Tensor(f!) status, Tensor(g!) sweeps) -> ()");
  m.def("mixing_reset(int handle, Tensor(a!) obs) -> ()");
  m.def("mixing_step(int handle, Tensor? actions, Tensor(a!) obs, Tensor(b!) rwd, Tensor(c!) done, Tensor(d!) trunc, "
        "Tensor(e!) status, Tensor(f!) sweeps) -> ()");
  m.def("burgers_reset(int handle, Tensor(a!) obs) -> ()");
  m.def("burgers_step(int handle, Tensor? actions, Tensor? noise, Tensor(a!) obs, Tensor(b!) rwd, Tensor(c!) done, Tensor(d!) trunc, "
        "Tensor(e!) status) -> ()");
  m.def("shkadov_reset(int handle, Tensor? init_fields, Tensor(a!) obs) -> ()");
  m.def("shkadov_step(int handle, Tensor? actions, Tensor? noise, Tensor(a!) obs, Tensor(b!) rwd, Tensor(c!) done, Tensor(d!) trunc, "
        "Tensor(e!) status) -> ()");
  m.def("sloshing_reset(int handle, Tensor? init_fields, Tensor(a!) obs) -> ()");
  m.def("sloshing_step(int handle, Tensor? actions, Tensor(a!) obs, Tensor(b!) rwd, Tensor(c!) done, Tensor(d!) trunc, "
        "Tensor(e!) status) -> ()");
}

TORCH_LIBRARY_IMPL(beacon, CUDA, m) {
  m.impl("rayleigh_reset", &rayleigh_reset);
  m.impl("rayleigh_step", &rayleigh_step);
  m.impl("mixing_reset", &mixing_reset);
  m.impl("mixing_step", &mixing_step);
  m.impl("burgers_reset", &burgers_reset);
  m.impl("burgers_step", &burgers_step);
  m.impl("shkadov_reset", &shkadov_reset);
  m.impl("shkadov_step", &shkadov_step);
  m.impl("sloshing_reset", &sloshing_reset);
  m.impl("sloshing_step", &sloshing_step);
}

TORCH_LIBRARY_IMPL(beacon, Meta, m) {
  m.impl("rayleigh_reset", &reset3_meta);
  m.impl("rayleigh_step", &rayleigh_step_meta);
  m.impl("mixing_reset", &reset2_meta);
  m.impl("mixing_step", &mixing_step_meta);
  m.impl("burgers_reset", &reset2_meta);
  m.impl("burgers_step", &noisy_step_meta);
  m.impl("shkadov_reset", &reset3_meta);
  m.impl("shkadov_step", &noisy_step_meta);
  m.impl("sloshing_reset", &reset3_meta);
  m.impl("sloshing_step", &sloshing_step_meta);
}
