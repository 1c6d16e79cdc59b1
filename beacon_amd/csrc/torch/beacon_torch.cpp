// beacon_torch.cpp -- the thin PyTorch-ROCm extension over the C ABI (include/beacon_hip.h): torch.library ops
//   beacon::{rayleigh,mixing,burgers,shkadov,sloshing}_{step,reset}(int handle, Tensor ...) -> ()
// Each op is ONE dispatcher call that takes device tensors, reads torch's current HIP stream in C++ and forwards to the
// bcn_* entry point of libbeacon_hip.so -- no ctypes marshalling, no Python-side stream query (what the per-call host cost of
// the ctypes binding was made of: scripts/host_cost.py), and an op torch.compile / CUDA-graph capture can see.  The ops
// mutate their output tensors in place and return nothing; shapes and dtypes are checked here, the values by the library.
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
inline void* dp(const Tensor& t, bcn_env_t h, bool real, const char* name) {
  TORCH_CHECK(t.is_cuda() && t.is_contiguous(), name, ": contiguous device tensor expected");
  if (real) {
    const auto want = bcn_dtype(h) == BCN_F64 ? at::kDouble : at::kFloat;
    TORCH_CHECK(t.scalar_type() == want, name, ": dtype ", t.scalar_type(), " but the handle computes in ", want);
  }
  return t.data_ptr();
}
inline void* dpo(OptT t, bcn_env_t h, bool real, const char* name) { return t.has_value() ? dp(*t, h, real, name) : nullptr; }
inline uint8_t* u8(const Tensor& t, const char* name) {
  TORCH_CHECK(t.is_cuda() && t.is_contiguous() && t.scalar_type() == at::kByte, name, ": contiguous uint8 device tensor expected");
  return t.data_ptr<uint8_t>();
}
inline int32_t* i32(const Tensor& t, const char* name) {
  TORCH_CHECK(t.is_cuda() && t.is_contiguous() && t.scalar_type() == at::kInt, name, ": contiguous int32 device tensor expected");
  return t.data_ptr<int32_t>();
}
inline void rows(const Tensor& t, bcn_env_t h, int64_t per, const char* name) {
  TORCH_CHECK(t.numel() == (int64_t)bcn_batch(h) * per, name, ": ", t.numel(), " elements, expected batch ", bcn_batch(h), " x ", per);
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
  if (actions.has_value()) rows(*actions, h, bcn_n_act(h), "actions");
  check(bcn_rayleigh_step(h, dpo(actions, h, true, "actions"), dp(actions_norm, h, true, "actions_norm"), dp(obs, h, true, "obs"),
                          dp(rwd, h, true, "rwd"), u8(done, "done"), u8(trunc, "trunc"), i32(status, "status"), i32(sweeps, "sweeps"),
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
  const int32_t* a = actions.has_value() ? (rows(*actions, h, 1, "actions"), i32(*actions, "actions")) : nullptr;
  check(bcn_mixing_step(h, a, dp(obs, h, true, "obs"), dp(rwd, h, true, "rwd"), u8(done, "done"), u8(trunc, "trunc"),
                        i32(status, "status"), i32(sweeps, "sweeps"), stream_of(obs)),
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
  check(bcn_burgers_step(h, dpo(actions, h, true, "actions"), dpo(noise, h, true, "noise"), dp(obs, h, true, "obs"),
                         dp(rwd, h, true, "rwd"), u8(done, "done"), u8(trunc, "trunc"), i32(status, "status"), stream_of(obs)),
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
  if (actions.has_value()) rows(*actions, h, bcn_n_act(h), "actions");
  check(bcn_shkadov_step(h, dpo(actions, h, true, "actions"), dpo(noise, h, true, "noise"), dp(obs, h, true, "obs"),
                         dp(rwd, h, true, "rwd"), u8(done, "done"), u8(trunc, "trunc"), i32(status, "status"), stream_of(obs)),
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
  check(bcn_sloshing_step(h, dpo(actions, h, true, "actions"), dp(obs, h, true, "obs"), dp(rwd, h, true, "rwd"), u8(done, "done"),
                          u8(trunc, "trunc"), i32(status, "status"), stream_of(obs)),
        "bcn_sloshing_step");
}

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
