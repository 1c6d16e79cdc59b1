"""Multi-GPU replica sharding: replicas are independent, so the env batch is
partitioned by replica index -- rank r owns global replicas [r*B_local, (r+1)*B_local) -- with NO
data-path collective inside the solver.  The only exchange per step() is the trainer-facing one:
rank 0 scatters actions[B_global, n_act] and gathers obs / rwd / done / trunc / status, over
torch.distributed (backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).

One process per GPU; messages are <= ~100 KB per rank per step, i.e. latency-bound."""
import torch
import torch.distributed as dist


def shard_bounds(n_global, world, rank):
    """Contiguous, equal shards; n_global must divide evenly (replica counts are ours to pick)."""
    if n_global % world:
        raise ValueError("global batch %d is not divisible by world size %d" % (n_global, world))
    per = n_global // world
    return rank * per, (rank + 1) * per


class ReplicaSharder(object):
    """Collectives of one sharded env batch.  Pure torch.distributed: works on CPU tensors with
    gloo (tests) and on device tensors with nccl/RCCL (production)."""

    def __init__(self, local_batch, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.local_batch = int(local_batch)
        self.global_batch = self.local_batch * self.world
        self._bufs = {}

    def scatter_actions(self, actions_global, like):
        """rank 0 holds actions_global[B_global, ...]; every rank returns its [B_local, ...] slice.
        `like`: a tensor giving shape[1:], dtype and device of the local slice."""
        if self.world == 1:
            return actions_global
        out = torch.empty((self.local_batch,) + tuple(like.shape[1:]), dtype=like.dtype, device=like.device)
        chunks = None
        if self.rank == 0:
            a = actions_global.to(device=like.device, dtype=like.dtype).contiguous()
            chunks = list(a.reshape((self.world, self.local_batch) + tuple(like.shape[1:])).unbind(0))
        dist.scatter(out, chunks, src=0, group=self.group)
        return out

    def gather(self, name, local):
        """Gather one per-replica tensor to rank 0 -> [B_global, ...] there, None elsewhere.
        Receive buffers are allocated once per name and reused."""
        if self.world == 1:
            return local
        local = local.contiguous()
        bufs = None
        if self.rank == 0:
            key = (name, tuple(local.shape), local.dtype, str(local.device))
            if key not in self._bufs:
                self._bufs[key] = torch.empty((self.world,) + tuple(local.shape), dtype=local.dtype,
                                              device=local.device)
            full = self._bufs[key]
            bufs = list(full.unbind(0))
        dist.gather(local, bufs, dst=0, group=self.group)
        if self.rank == 0:
            return full.reshape((self.global_batch,) + tuple(local.shape[1:]))
        return None


class ShardedVecEnv(object):
    """Wraps the local VecEnv of each rank behind a rank-0-facing global batch.

        env = ShardedVecEnv(VecRayleigh(B_local, device=f"cuda:{local_rank}", ...))
        obs, _ = env.reset()                       # rank 0: [B_global, n_obs]; other ranks: None
        obs, rwd, done, trunc, _ = env.step(actions_global_or_None_on_other_ranks)
    """

    def __init__(self, local_env, group=None):
        self.env = local_env
        self.sh = ReplicaSharder(local_env.batch, group)
        self.global_batch = self.sh.global_batch
        self.lo, self.hi = shard_bounds(self.global_batch, self.sh.world, self.sh.rank)

    def _like_actions(self):
        e = self.env
        if e.action_is_int:
            return torch.empty((e.batch,), dtype=torch.int32, device=e.device)
        shape = (e.batch,) if e.n_actions == 1 else (e.batch, e.n_actions)
        return torch.empty(shape, dtype=e.tdtype, device=e.device)

    def reset(self):
        obs, _ = self.env.reset()
        return self.sh.gather("obs", obs), None

    def step(self, actions_global=None, noise=None, scattered=False):
        """actions_global: full [B_global, ...] on rank 0 (ignored elsewhere) unless
        scattered=True, in which case every rank passes its own local slice."""
        if scattered or self.sh.world == 1:
            local = actions_global
        else:
            local = self.sh.scatter_actions(actions_global, self._like_actions())
        obs, rwd, done, trunc, _ = self.env.step(local, noise)
        return (self.sh.gather("obs", obs), self.sh.gather("rwd", rwd), self.sh.gather("done", done),
                self.sh.gather("trunc", trunc), None)

    def gather_status(self):
        return self.sh.gather("status", self.env.status)

    def close(self):
        self.env.close()
