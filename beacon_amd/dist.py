"""Multi-GPU replica sharding: replicas are independent, so the env batch is
partitioned by replica index -- rank r owns global replicas [r*B_local, (r+1)*B_local) -- with NO
data-path collective inside the solver.  The only exchange per step() is the trainer-facing one:
rank 0 scatters actions[B_global, n_act] and gathers the packed per-step outputs (obs, rwd, status,
done, trunc: ONE byte buffer per rank, ONE collective) over torch.distributed (backend "nccl" == RCCL
over xGMI on ROCm; "gloo" in the CPU tests).

One process per GPU; the message is ~100-800 KB per rank per step, i.e. latency-bound."""
import torch
import torch.distributed as dist

from .vec import unpack_outputs


def shard_bounds(n_global, world, rank):
    """Contiguous, equal shards; n_global must divide evenly (replica counts are ours to pick)."""
    if n_global % world:
        raise ValueError("global batch %d is not divisible by world size %d" % (n_global, world))
    per = n_global // world
    return rank * per, (rank + 1) * per


class ReplicaSharder(object):
    """Collectives of one sharded env batch.  Pure torch.distributed: works on CPU tensors with
    gloo (tests) and on device tensors with nccl/RCCL (production)."""

    def __init__(self, local_batch, group=None, always_collective=False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # a world of one rank normally skips the collectives; always_collective sends its scatter / gather through the
        # backend all the same (the one-GPU rehearsal of the RCCL path: tests/test_gpu_parity.py)
        self.collective = self.world > 1 or (always_collective and dist.is_initialized())
        self.local_batch = int(local_batch)
        self.global_batch = self.local_batch * self.world
        self._bufs = {}

    def scatter_actions(self, actions_global, like):
        """rank 0 holds actions_global[B_global, ...]; every rank returns its [B_local, ...] slice.
        `like`: a tensor giving shape[1:], dtype and device of the local slice."""
        if not self.collective:
            return actions_global
        out = torch.empty((self.local_batch,) + tuple(like.shape[1:]), dtype=like.dtype, device=like.device)
        chunks = None
        if self.rank == 0:
            a = actions_global.to(device=like.device, dtype=like.dtype).contiguous()
            chunks = list(a.reshape((self.world, self.local_batch) + tuple(like.shape[1:])).unbind(0))
        dist.scatter(out, chunks, src=0, group=self.group)
        return out

    def gather(self, name, local):
        """Gather one per-replica tensor to rank 0 -> [world, ...local.shape] there, None elsewhere.
        Receive buffers are allocated once per name and reused."""
        if not self.collective:
            return local.unsqueeze(0)
        local = local.contiguous()
        bufs = None
        full = None
        if self.rank == 0:
            key = (name, tuple(local.shape), local.dtype, str(local.device))
            if key not in self._bufs:
                self._bufs[key] = torch.empty((self.world,) + tuple(local.shape), dtype=local.dtype,
                                              device=local.device)
            full = self._bufs[key]
            bufs = list(full.unbind(0))
        dist.gather(local, bufs, dst=0, group=self.group)
        return full

    def gather_outputs(self, out_buf, obs_dim, tdtype):
        """ONE collective per step: every rank's packed output buffer (vec.out_layout) to rank 0, which
        returns (obs[B_global, n], rwd, status, done, trunc) assembled from the per-rank segments;
        other ranks return None."""
        full = self.gather("out", out_buf)
        if full is None:
            return None
        parts = [unpack_outputs(full[r], self.local_batch, obs_dim, tdtype) for r in range(full.shape[0])]
        if len(parts) == 1:
            return parts[0]
        return tuple(torch.cat([p[k] for p in parts], dim=0) for k in range(5))


class ShardedVecEnv(object):
    """Wraps the local VecEnv of each rank behind a rank-0-facing global batch.

        env = ShardedVecEnv(VecRayleigh(B_local, device=f"cuda:{local_rank}", ...))
        obs, _ = env.reset()                       # rank 0: [B_global, n_obs]; other ranks: None
        obs, rwd, done, trunc, _ = env.step(actions_global_or_None_on_other_ranks)

    Envs that draw inlet noise on the device (burgers, shkadov) are re-seeded with seed + global replica
    offset when the batch is sharded, so that replica i of different ranks does not receive the same noise
    stream; `seed` defaults to the seed the env itself was constructed with (VecBurgers(seed=...)), and an
    unsharded env (world size 1) keeps its generator untouched."""

    def __init__(self, local_env, group=None, seed=None, always_collective=False):
        self.env = local_env
        self.sh = ReplicaSharder(local_env.batch, group, always_collective)
        self.global_batch = self.sh.global_batch
        self.lo, self.hi = shard_bounds(self.global_batch, self.sh.world, self.sh.rank)
        self.status = None
        if getattr(local_env, "gen", None) is not None and (self.sh.world > 1 or seed is not None):
            base = int(getattr(local_env, "seed", 0) if seed is None else seed)
            local_env.gen.manual_seed(base + self.lo)
            if hasattr(local_env, "set_noise_seed"):       # in-kernel noise: one seed, keyed by the GLOBAL replica index
                local_env.set_noise_seed(base, self.lo)

    def _like_actions(self):
        e = self.env
        if e.action_is_int:
            return torch.empty((e.batch,), dtype=torch.int32, device=e.device)
        shape = (e.batch,) if e.n_actions == 1 else (e.batch, e.n_actions)
        return torch.empty(shape, dtype=e.tdtype, device=e.device)

    def _gather(self):
        e = self.env
        return self.sh.gather_outputs(e.out_buf, e.obs_dim, e.tdtype)

    def reset(self):
        self.env.reset()
        g = self._gather()
        return (g[0] if g is not None else None), None

    def step(self, actions_global=None, noise=None, scattered=False):
        """actions_global: full [B_global, ...] on rank 0 (ignored elsewhere) unless
        scattered=True, in which case every rank passes its own local slice."""
        if scattered or not self.sh.collective:
            local = actions_global
        else:
            local = self.sh.scatter_actions(actions_global, self._like_actions())
        self.env.step(local, noise)
        g = self._gather()
        if g is None:
            return None, None, None, None, None
        obs, rwd, self.status, done, trunc = g
        return obs, rwd, done, trunc, None

    def gather_status(self):
        """Status words [B_global] of the last step() -- NOT a collective: they travelled with that step's packed
        outputs, so this returns rank 0's cached copy; None before the first step() and on every other rank."""
        return self.status

    def close(self):
        self.env.close()
