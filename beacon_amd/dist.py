"""Multi-GPU replica sharding: replicas are independent, so the env batch is
partitioned by replica index -- rank r owns global replicas [r*B_local, (r+1)*B_local) -- with NO
data-path collective inside the solver.  The only exchange per step() is the trainer-facing one:
rank 0 scatters actions[B_global, n_act] (and replica masks) and gathers the packed per-step outputs
(obs, rwd, status, done, trunc: ONE byte buffer per rank, ONE collective) over torch.distributed
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).

One process per GPU; the message is ~100-800 KB per rank per step, i.e. latency-bound.

The gather does not have to sit between two steps: with `overlap=True` the local env rotates through THREE
packed output buffers (VecEnv.double_buffer: step k writes buffer k % 3) and step_async() issues the gather
of step k on a SIDE stream behind an event, so the kernel of step k + 1 is enqueued directly behind the kernel
of step k; the gather of step k is waited for only by whoever reads its result and by step k + 3, which
overwrites the buffer it reads.  (Three, not two: the step kernels are persistent and occupy every CU, so the
gather's kernel gets a CU only while step k + 1 drains -- with two buffers step k + 2 would have to wait for it.)  (A trainer whose actions of step k + 1 depend on the observations of step k
calls step(), which is step_async().wait(): nothing can overlap then, by the data dependence itself.)"""
import os
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

    def scatter(self, full, like):
        """rank 0 holds full[B_global, ...] (actions, a replica mask, ...); every rank returns its [B_local, ...] slice.
        `like`: a tensor giving shape[1:], dtype and device of the local slice."""
        if not self.collective:
            return full
        out = torch.empty((self.local_batch,) + tuple(like.shape[1:]), dtype=like.dtype, device=like.device)
        chunks = None
        if self.rank == 0:
            a = torch.as_tensor(full).to(device=like.device, dtype=like.dtype).contiguous()
            chunks = list(a.reshape((self.world, self.local_batch) + tuple(like.shape[1:])).unbind(0))
        dist.scatter(out, chunks, src=0, group=self.group)
        return out

    scatter_actions = scatter

    def recv_buffer(self, name, local):
        """rank 0: the persistent [world, ...local.shape] receive buffer of this name (allocated once); None elsewhere."""
        if self.rank != 0:
            return None
        key = (name, tuple(local.shape), local.dtype, str(local.device))
        if key not in self._bufs:
            self._bufs[key] = torch.empty((self.world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        return self._bufs[key]

    def gather(self, name, local, async_op=False):
        """Gather one per-replica tensor to rank 0 -> [world, ...local.shape] there, None elsewhere.
        Receive buffers are allocated once per name and reused.  async_op: returns (full, work)."""
        if not self.collective:
            full = local.unsqueeze(0)
            return (full, None) if async_op else full
        local = local.contiguous()
        full = self.recv_buffer(name, local)
        bufs = list(full.unbind(0)) if full is not None else None
        work = dist.gather(local, bufs, dst=0, group=self.group, async_op=async_op)
        return (full, work) if async_op else full

    def assemble(self, full, obs_dim, tdtype):
        """(obs[B_global, n], rwd, status, done, trunc) from the gathered per-rank packed buffers [world, bytes]."""
        parts = [unpack_outputs(full[r], self.local_batch, obs_dim, tdtype) for r in range(full.shape[0])]
        if len(parts) == 1:
            return parts[0]
        return tuple(torch.cat([p[k] for p in parts], dim=0) for k in range(5))

    def gather_outputs(self, out_buf, obs_dim, tdtype, name="out"):
        """ONE collective per step: every rank's packed output buffer (vec.out_layout) to rank 0, which
        returns (obs[B_global, n], rwd, status, done, trunc) assembled from the per-rank segments;
        other ranks return None."""
        full = self.gather(name, out_buf)
        if full is None:
            return None
        return self.assemble(full, obs_dim, tdtype)


class PendingStep(object):
    """The outputs of one ShardedVecEnv.step_async(): the step kernel is enqueued on the caller's stream, the gather of its
    packed outputs on the env's side stream.  wait() makes the CALLER'S CURRENT STREAM wait for that gather (no host
    synchronisation on device tensors; with gloo / CPU tensors it blocks the host until the collective is done) and
    returns (obs, rwd, done, trunc, None) on rank 0, (None, ...) elsewhere.  The tensors are views of a receive buffer
    that the gather NBUF steps later overwrites; that gather waits for everything enqueued until then on the stream
    wait() was called on, so a consumer on its own stream (a trainer's copy stream) needs no further synchronisation."""

    def __init__(self, senv, full, work, done_event):
        self.senv, self.full, self.work, self.done_event = senv, full, work, done_event
        self.result = None
        self.consumer = None          # the stream wait() was called on

    def _finish(self):
        """Host side of the collective (gloo): block until it is done.  Device side (nccl): nothing to do here."""
        if self.work is not None and self.done_event is None:
            self.work.wait()
            self.work = None

    def wait(self):
        if self.result is not None:
            return self.result
        self._finish()
        if self.done_event is not None:
            self.consumer = torch.cuda.current_stream(self.senv.env.device)
            self.consumer.wait_event(self.done_event)
        e = self.senv.env
        if self.full is None:
            self.result = (None, None, None, None, None)
        else:
            obs, rwd, self.senv.status, done, trunc = self.senv.sh.assemble(self.full, e.obs_dim, e.tdtype)
            self.result = (obs, rwd, done, trunc, None)
        return self.result


class ShardedVecEnv(object):
    """Wraps the local VecEnv of each rank behind a rank-0-facing global batch.

        env = ShardedVecEnv(VecRayleigh(B_local, device=f"cuda:{local_rank}", ...))
        obs, _ = env.reset()                       # rank 0: [B_global, n_obs]; other ranks: None
        obs, rwd, done, trunc, _ = env.step(actions_global_or_None_on_other_ranks)
        obs, _ = env.reset_done()                  # auto-reset of the replicas whose episode ended (every rank its own)
        obs, _ = env.reset(mask=global_mask_on_rank_0)          # the trainer resets single envs (rayleigh.py:89-99)
        p = env.step_async(a); ...; obs, rwd, done, trunc, _ = p.wait()     # overlap=True: gather beside the next step

    Envs that draw inlet noise on the device (burgers, shkadov) are re-seeded with seed + global replica
    offset when the batch is sharded, so that replica i of different ranks does not receive the same noise
    stream; `seed` defaults to the seed the env itself was constructed with (VecBurgers(seed=...)), and an
    unsharded env (world size 1) keeps its generator untouched."""

    NBUF = 3      # default number of output buffers of the overlapped path (constructor argument `nbuf`, >= 2)

    def __init__(self, local_env, group=None, seed=None, always_collective=False, overlap=False, nbuf=None, check_calls=None):
        """nbuf: output buffers of the overlapped path (default: environment BEACON_NBUF, else 3; at least 2 -- step k + nbuf
        overwrites what gather k reads).  check_calls (default: environment BEACON_DIST_CHECK=1): every reset() / step()
        first verifies, with one tiny all-reduce and a host read, that all ranks made the same call with the same
        mask / no-mask shape -- a debugging aid for SPMD trainer scripts (it serialises host and device; leave it off when
        timing)."""
        if nbuf is None:
            raw = os.environ.get("BEACON_NBUF", str(ShardedVecEnv.NBUF))
            try:
                nbuf = int(raw)
            except ValueError:
                raise ValueError("BEACON_NBUF=%r is not an integer" % raw)
        if int(nbuf) < 2:
            raise ValueError("ShardedVecEnv: nbuf = %s, need at least 2 output buffers" % nbuf)
        self.NBUF = int(nbuf)
        self.check_calls = (os.environ.get("BEACON_DIST_CHECK") == "1") if check_calls is None else bool(check_calls)
        self.env = local_env
        self.sh = ReplicaSharder(local_env.batch, group, always_collective)
        self.global_batch = self.sh.global_batch
        self.lo, self.hi = shard_bounds(self.global_batch, self.sh.world, self.sh.rank)
        self.status = None
        self.overlap = bool(overlap) and self.sh.collective
        self._pending = [None] * self.NBUF    # gather in flight per output buffer
        self._consumers = [None] * self.NBUF  # stream that took delivery of the previous gather into that receive area
        self._side = None
        if self.overlap:
            local_env.double_buffer(True, self.NBUF)
            if torch.device(local_env.device).type == "cuda":
                self._side = torch.cuda.Stream(device=local_env.device)
        if getattr(local_env, "gen", None) is not None and (self.sh.world > 1 or seed is not None):
            base = int(getattr(local_env, "seed", 0) if seed is None else seed)
            local_env.gen.manual_seed(base + self.lo)
            if hasattr(local_env, "set_noise_seed"):       # in-kernel noise: one seed, keyed by the GLOBAL replica index
                local_env.set_noise_seed(base, self.lo)

    # -- scatter helpers ------------------------------------------------------------------------
    def _like_actions(self):
        e = self.env
        if e.action_is_int:
            return torch.empty((e.batch,), dtype=torch.int32, device=e.device)
        shape = (e.batch,) if e.n_actions == 1 else (e.batch, e.n_actions)
        return torch.empty(shape, dtype=e.tdtype, device=e.device)

    def _check_call(self, what, has_mask, scattered):
        """check_calls: all ranks are in the same call with the same shape (which collectives follow depends on it: a rank that
        passes mask=None while rank 0 passes a mask would skip the mask scatter and meet rank 0 in the wrong collective)."""
        if not (self.check_calls and self.sh.collective):
            return
        code = float(1 + 4 * ("reset", "step", "reset_done").index(what) + (1 if has_mask else 0) + (2 if scattered else 0))
        t = torch.tensor([code, -code], dtype=torch.float32, device=self.env.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.sh.group)
        hi, lo = float(t[0]), -float(t[1])
        if hi != lo:
            raise RuntimeError("ShardedVecEnv.%s: the ranks disagree on the call (rank %d: %s, mask %s, scattered %s; codes %g..%g): "
                               "every rank must make the same call, with a non-None `mask` placeholder wherever rank 0 passes a mask"
                               % (what, self.sh.rank, what, "given" if has_mask else "None", scattered, lo, hi))

    def _local_mask(self, mask_global, scattered):
        """mask_global: [B_global] bool / uint8 on rank 0, or every rank's own [B_local] slice when `scattered`; None = all
        replicas.  As with torch.distributed's own scatter, the CALL is collective and only the payload is rank 0's:
        when rank 0 passes a mask, every other rank passes a non-None placeholder (its content is ignored) -- the ranks of
        an SPMD trainer script make the same calls, and `check_calls` verifies it."""
        if mask_global is None or scattered or not self.sh.collective:
            return mask_global
        like = torch.empty((self.env.batch,), dtype=torch.uint8, device=self.env.device)
        if self.sh.rank == 0:
            mask_global = torch.as_tensor(mask_global).to(torch.uint8)
        return self.sh.scatter(mask_global, like)

    # -- gather ---------------------------------------------------------------------------------
    def _drain(self):
        """Everything in flight is finished from the caller's stream's point of view (before a blocking collective on
        the main stream touches the buffers)."""
        main = torch.cuda.current_stream(self.env.device) if self._side is not None else None
        for k, p in enumerate(self._pending):
            cons = self._consumers[k]
            if p is not None:
                p.wait()
                cons = p.consumer if p.consumer is not None else cons
                self._pending[k] = None
            if cons is not None and cons != main:     # delivered on another stream: its reads of the receive area first
                main.wait_stream(cons)
            self._consumers[k] = None

    def _gather(self):
        e = self.env
        self._drain()
        g = self.sh.gather_outputs(e.out_buf, e.obs_dim, e.tdtype, name="out%d" % getattr(e, "_cur", 0))
        return g

    def _gather_async(self):
        """Gather of the buffer the last step wrote, beside whatever the caller enqueues next."""
        e = self.env
        k = getattr(e, "_cur", 0)
        name = "out%d" % k
        if self._side is None:            # CPU tensors / no side stream: the backend's own asynchrony (gloo worker thread)
            full, work = self.sh.gather(name, e.out_buf, async_op=True)
            p = PendingStep(self, full, work, None)
        else:
            main = torch.cuda.current_stream(e.device)
            ready = torch.cuda.Event()
            ready.record(main)                         # the step kernel (and everything before it) on the caller's stream
            done = torch.cuda.Event()
            with torch.cuda.stream(self._side):
                self._side.wait_event(ready)
                cons = self._consumers[k]              # whoever read this receive area NBUF steps ago, on its own stream
                if cons is not None and cons != main:
                    self._side.wait_stream(cons)
                self._consumers[k] = None
                full, work = self.sh.gather(name, e.out_buf, async_op=True)
                if work is not None:
                    work.wait()                        # side stream waits for the backend's stream; the host does not
                done.record(self._side)
            p = PendingStep(self, full, None, done)
        self._pending[k] = p
        return p

    # -- Gym surface ----------------------------------------------------------------------------
    def reset(self, mask=None, scattered=False):
        """Reset every replica, or those selected by `mask` ([B_global] on rank 0, or the local slice on every rank with
        scattered=True): what the reference's trainer does when it calls reset() on the one env whose episode ended
        (rayleigh.py:89-99).  Returns the global observations on rank 0."""
        self._drain()
        self._check_call("reset", mask is not None, scattered)
        m = self._local_mask(mask, scattered)
        self.env.reset(mask=m)
        g = self._gather()
        return (g[0] if g is not None else None), None

    def reset_done(self):
        """Auto-reset on every rank of its own replicas whose last step() returned done (VecEnv.reset_done: on the device,
        no mask travels), then the gather of the refreshed observations."""
        self._drain()
        self._check_call("reset_done", False, False)
        self.env.reset_done()
        g = self._gather()
        return (g[0] if g is not None else None), None

    def step_async(self, actions_global=None, noise=None, scattered=False, mask=None):
        """Enqueue one step and the gather of its outputs; returns a PendingStep.  actions_global / mask: full
        [B_global, ...] on rank 0 (ignored elsewhere) unless scattered=True, in which case every rank passes its own local
        slice.  With overlap=False the gather is a blocking collective on the caller's stream and the PendingStep is
        already complete."""
        e = self.env
        self._check_call("step", mask is not None, scattered)
        if scattered or not self.sh.collective:
            local = actions_global
        else:
            local = self.sh.scatter(actions_global, self._like_actions())
        m = self._local_mask(mask, scattered)
        if not self.overlap:
            e.step(local, noise, mask=m)
            g = self._gather()
            p = PendingStep(self, None, None, None)
            if g is None:
                p.result = (None, None, None, None, None)
            else:
                obs, rwd, self.status, done, trunc = g
                p.result = (obs, rwd, done, trunc, None)
            return p
        # the step about to be enqueued writes the NEXT buffer of the ring: the gather that still reads it (NBUF steps back) first
        nxt = (e._cur + 1) % self.NBUF
        old = self._pending[nxt]
        if old is not None:
            old._finish()
            if old.done_event is not None:
                torch.cuda.current_stream(e.device).wait_event(old.done_event)
            self._consumers[nxt] = old.consumer
            self._pending[nxt] = None
        e.step(local, noise, mask=m)
        return self._gather_async()

    def step(self, actions_global=None, noise=None, scattered=False, mask=None):
        return self.step_async(actions_global, noise, scattered, mask).wait()

    def gather_status(self):
        """Status words [B_global] of the last step() -- NOT a collective: they travelled with that step's packed
        outputs, so this returns rank 0's cached copy; None before the first step() and on every other rank."""
        return self.status

    def close(self):
        self._drain()
        self.env.close()
