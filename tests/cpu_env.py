"""CPU stand-in env for the host-logic tests (no GPU, no kernels): it REUSES beacon_amd.vec.VecEnv's own host code --
packed and double-buffered outputs, replica masks, reset(mask) / step(mask) / reset_done() -- and replaces only the three
calls that would reach libbeacon_hip.so (_apply_mask, _reset, _step) by a few tensor operations on CPU tensors with a
per-replica episode counter.  Test infrastructure, not a fallback: the product classes raise without a GPU."""
import torch

from beacon_amd.vec import VecEnv


class CpuVecEnv(VecEnv):
    action_is_int = False

    def __init__(self, batch, obs_dim=3, n_actions=3, n_act=4):
        self.batch, self.obs_dim, self.n_actions, self.n_act = int(batch), obs_dim, n_actions, n_act
        self.device, self.tdtype = torch.device("cpu"), torch.float32
        self.h = None
        self.state = torch.zeros((self.batch, obs_dim))
        self.stp = torch.zeros((self.batch,), dtype=torch.int32)
        self._mask = None
        self._alloc_outputs()
        self.gen = torch.Generator()

    def _apply_mask(self, mask):
        self._mask = None if mask is None else torch.as_tensor(mask).to(torch.uint8).reshape(self.batch).clone()

    def _sel(self):
        return torch.ones((self.batch,), dtype=torch.bool) if self._mask is None else self._mask.bool()

    def _reset(self):
        m = self._sel()
        self.state[m] = 0.0
        self.stp[m] = 0
        self.obs[m] = -1.0

    def _step(self, actions, noise=None):
        m = self._sel()
        a = torch.as_tensor(actions, dtype=torch.float32).reshape(self.batch, self.n_actions)
        self.state[m] = self.state[m] * 0.5 + a[m]
        end = self.stp == self.n_act - 1
        self.stp[m] += 1
        self.obs[m] = self.state[m] * 2.0
        self.rwd[m] = a[m].sum(1)
        self.done[m] = end[m].to(torch.uint8)
        self.trunc[m] = end[m].to(torch.uint8)
        self.status[m] = 0

    def set_noise_seed(self, seed, replica_offset=0):
        self.seed, self.replica_offset = int(seed), int(replica_offset)

    def check_status(self):
        return self.status

    def close(self):
        pass
