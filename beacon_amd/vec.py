"""Batched (vectorised) envs: B independent replicas of one beacon env advanced by one HIP
launch per step().  Same reset()/step() surface as the reference's Gym classes, with a
leading batch dimension; tensors stay on the GPU (torch is only the allocator / stream
provider -- all arithmetic happens in libbeacon_hip.so).

Derived parameters are computed exactly as the reference constructors do (citations are
file:line into /root/reference/beacon/)."""
import ctypes as C
import os
import math

import numpy as np
import torch

from . import _lib
from . import spaces
from .spaces import Box, Discrete   # noqa: F401  (stand-in classes, kept importable from here)

_DT = {"f32": (torch.float32, _lib.F32), "f64": (torch.float64, _lib.F64),
       "float32": (torch.float32, _lib.F32), "float64": (torch.float64, _lib.F64),
       torch.float32: (torch.float32, _lib.F32), torch.float64: (torch.float64, _lib.F64)}


def out_layout(batch, obs_dim, esz):
    """Byte offsets of the packed per-step outputs of `batch` replicas."""
    def up(x):
        return (x + 15) // 16 * 16
    o_obs = 0
    o_rwd = up(o_obs + batch * obs_dim * esz)
    o_status = up(o_rwd + batch * esz)
    o_done = up(o_status + batch * 4)
    o_trunc = up(o_done + batch)
    return {"obs": o_obs, "rwd": o_rwd, "status": o_status, "done": o_done, "trunc": o_trunc,
            "bytes": up(o_trunc + batch)}


def unpack_outputs(buf, batch, obs_dim, tdtype):
    """Typed views (obs[B, n], rwd[B], status[B] int32, done[B] u8, trunc[B] u8) of one packed byte buffer."""
    esz = torch.empty((), dtype=tdtype).element_size()
    lay = out_layout(batch, obs_dim, esz)
    obs = buf[lay["obs"]:lay["obs"] + batch * obs_dim * esz].view(tdtype).view(batch, obs_dim)
    rwd = buf[lay["rwd"]:lay["rwd"] + batch * esz].view(tdtype)
    status = buf[lay["status"]:lay["status"] + batch * 4].view(torch.int32)
    done = buf[lay["done"]:lay["done"] + batch]
    trunc = buf[lay["trunc"]:lay["trunc"] + batch]
    return obs, rwd, status, done, trunc


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


_OPS = ("rayleigh_reset", "rayleigh_step", "mixing_reset", "mixing_step", "burgers_reset", "burgers_step", "shkadov_reset",
        "shkadov_step", "sloshing_reset", "sloshing_step")


def _op_table():
    """{name: torch.ops.beacon.<name>.default} of the torch extension (beacon_amd/torch_ext.py), or None without it."""
    from . import torch_ext
    ops = torch_ext.load()
    return None if ops is None else {n: getattr(ops, n).default for n in _OPS}


class VecEnv(object):
    """Common machinery.  Subclasses set self.cfg and implement _create/_reset/_step."""

    action_is_int = False
    needs_noise = False
    _plugin_defs = None      # extra -D flags of this class's on-demand kernels (tests: the deliberately broken plugin)

    def __init__(self, batch, device="cuda:0", dtype="f32"):
        if not torch.cuda.is_available():
            raise RuntimeError("beacon_amd needs a ROCm GPU: the solver path is HIP-only (no CPU fallback)")
        self.lib = _lib.load()
        # the torch.library ops over the same C ABI (beacon_amd/torch_ext.py): one dispatcher call per reset() / step();
        # None (no compiler and no prebuilt extension, or BEACON_TORCH_EXT=0): the ctypes binding below
        self._ops = _op_table()
        self.batch = int(batch)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("device must be a cuda (ROCm) device")
        self.dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", self.dev_index)
        self.tdtype, self.cdtype = _DT[dtype]
        self.h = C.c_void_p()
        self._mask = None
        self._create()
        self.obs_dim = self.lib.bcn_n_obs(self.h)       # observation length per replica
        self.n_actions = self.lib.bcn_n_act(self.h)
        self._alloc_outputs()

    def _alloc_outputs(self):
        """Per-step outputs live in ONE byte buffer [obs | rwd | status | done | trunc] (segments 16-byte
        aligned), so the trainer-facing gather of a sharded batch is a single collective on `out_buf`
        (beacon_amd/dist.py); obs / rwd / ... are typed views of it.  There may be two such buffers
        (double_buffer): `out_buf`, `obs`, `rwd`, `status`, `done`, `trunc` always name the one the last
        reset() / step() wrote."""
        B, esz = self.batch, torch.empty((), dtype=self.tdtype).element_size()
        self.out_layout = out_layout(B, self.obs_dim, esz)
        self.out_bufs, self._views, self._rotate = [], [], 0
        self._add_out_buf()
        self._bind_outputs(0)

    def _add_out_buf(self):
        buf = torch.zeros((self.out_layout["bytes"],), dtype=torch.uint8, device=self.device)
        self.out_bufs.append(buf)
        self._views.append(unpack_outputs(buf, self.batch, self.obs_dim, self.tdtype))

    def _bind_outputs(self, k):
        self._cur = k
        self.out_buf = self.out_bufs[k]
        self.obs, self.rwd, self.status, self.done, self.trunc = self._views[k]

    def double_buffer(self, on=True, nbuf=2):
        """Rotate through `nbuf` packed output buffers: step k writes buffer k % nbuf, so that a consumer of step k's
        outputs on another stream -- the sharded batch's gather to rank 0 (beacon_amd/dist.py), a device-to-host copy --
        may still be reading them while the next nbuf - 1 steps run.  After every step() the attributes obs / rwd / done /
        trunc / status / out_buf are re-bound to the buffer that step wrote (so hold on to the tensors a step RETURNS, not
        to the attributes, and expect them to be overwritten nbuf steps later).  reset() writes the current buffer.
        A step with a replica mask first copies the previous buffer (the rows of skipped replicas keep their values).
        Not for captured graphs (StepGraph records fixed addresses)."""
        while on and len(self.out_bufs) < int(nbuf):
            self._add_out_buf()
        self._rotate = int(nbuf) if on else 0
        if not on:
            self._bind_outputs(self._cur)
        return self

    def _next_outputs(self, carry):
        """Called by step() before the launch: with rotating outputs, switch to the next buffer."""
        if not self._rotate:
            return
        prev = self.out_buf
        self._bind_outputs((self._cur + 1) % self._rotate)
        if carry:
            self.out_buf.copy_(prev)

    # -- plumbing ---------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _real(self, x, shape):
        """actions / noise / init fields -> contiguous device tensor of the env dtype."""
        if x is None:
            return None
        if (torch.is_tensor(x) and x.dtype == self.tdtype and x.device == self.device and tuple(x.shape) == tuple(shape)
                and x.is_contiguous()):
            return x                      # what a trainer passes every step: nothing to convert
        if not torch.is_tensor(x):
            x = torch.as_tensor(np.asarray(x, dtype=np.float64))
        x = x.to(device=self.device, dtype=self.tdtype).reshape(shape).contiguous()
        return x

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self.lib.bcn_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- state ------------------------------------------------------------------------------
    def state_shape(self):
        raise NotImplementedError

    def get_state(self):
        """Solver fields of every replica as one device tensor (layout: include/beacon_hip.h)."""
        out = torch.empty((self.batch,) + self.state_shape(), dtype=self.tdtype, device=self.device)
        _lib.check(self.lib.bcn_get_state(self.h, _ptr(out), 1, self._stream()))
        return out

    def set_state(self, state):
        st = self._real(state, (self.batch,) + self.state_shape())
        _lib.check(self.lib.bcn_set_state(self.h, _ptr(st), 1, self._stream()))
        torch.cuda.current_stream(self.device).synchronize()  # `st` may be a temporary

    def get_stp(self):
        buf = (C.c_int32 * self.batch)()
        _lib.check(self.lib.bcn_get_stp(self.h, buf, self._stream()))
        return np.frombuffer(buf, dtype=np.int32).copy()

    def set_stp(self, stp):
        arr = np.ascontiguousarray(np.broadcast_to(np.asarray(stp, dtype=np.int32), (self.batch,)))
        _lib.check(self.lib.bcn_set_stp(self.h, arr.ctypes.data_as(_lib.c_i32p), self._stream()))

    def set_variant(self, v):
        return self.lib.bcn_set_variant(self.h, int(v))

    def set_sched(self, mode=-1, grid=0, q=0, lpt_min_batch=0):
        """Scheduling of the register-resident 2D kernels (include/beacon_hip.h: bcn_set_sched);
        results do not depend on it."""
        _lib.check(self.lib.bcn_set_sched(self.h, int(mode), int(grid), int(q), int(lpt_min_batch)))

    def _attach_plugin(self, kind):
        """2D envs: a grid without a built-in register-resident kernel gets one compiled for it (beacon_amd/jit.py);
        when that is not possible the generic kernel stays selected."""
        if self.lib.bcn_set_variant(self.h, 1) == 1 and os.environ.get("BEACON_JIT_FORCE") != "1":
            return
        from . import jit
        f64 = self.tdtype == torch.float64
        p = jit.plugin_for(self.nx, self.ny, f64, kind, getattr(self, "_plugin_defs", None))
        if p is None:
            return
        if p.verified is None and not jit.CHECKING:
            # first use of this shared object: compared with the generic kernel before any env runs on it (jit.verify)
            ctor, cls, dev, dt = dict(self._ctor), type(self), self.device, ("f64" if f64 else "f32")
            jit.verify(p, lambda batch: cls(batch, dev, dt, **ctor), kind, f64)
        if p.verified or jit.CHECKING:
            _lib.check(self.lib.bcn_set_fast_plugin(self.h, p.fn, p.scratch))
            self._plugin = p

    def _slow_mode_bound(self, kind):
        """The per-grid constants of conv_plan 3's slow-mode landing guard (beacon_amd/stoprule.py), for grids the library has
        not built in.  The one-row and two-rows-per-lane kernels (ny <= 128) use them; the hybrid and the generic kernel do not."""
        if self.ny > 128 or min(self.nx, self.ny) < 48 or os.environ.get("BEACON_STOPRULE") == "0":
            return
        two = C.c_double * 2
        if self.lib.bcn_get_slow_mode_bound(self.h, two(), two()) > 0:
            return
        from . import stoprule
        cx = self.dy * self.dy / (2.0 * (self.dx * self.dx + self.dy * self.dy))
        b = stoprule.bounds(self.nx, self.ny, kind, cx)
        if b:
            _lib.check(self.lib.bcn_set_slow_mode_bound(self.h, len(b), two(*[c for c, _ in b]), two(*[v for _, v in b])))

    def set_slow_mode_bound(self, pairs):
        """pairs: up to two (cutoff, bound) of this grid (beacon_amd/stoprule.py: bounds); [] clears them.  The library trusts the
        caller: constants below the grid's true ones void the proof of conv_plan 3's landings (tests/test_gpu_parity.py shows it)."""
        two = C.c_double * 2
        _lib.check(self.lib.bcn_set_slow_mode_bound(self.h, len(pairs), two(*[c for c, _ in pairs]), two(*[b for _, b in pairs])))

    def slow_mode_bound(self):
        """[(cutoff, bound)] in force (include/beacon_hip.h: bcn_get_slow_mode_bound); empty: the guard is BCN_CONV_GUARD alone."""
        two = C.c_double * 2
        c, b = two(), two()
        n = self.lib.bcn_get_slow_mode_bound(self.h, c, b)
        return [(c[k], b[k]) for k in range(max(n, 0))]

    def use_torch_ops(self, on=True):
        """Switch this env between the two bindings of the C ABI: the torch.library ops (default when the extension is built)
        and ctypes.  Returns whether the ops are in use.  Results do not depend on it (tests/test_gpu_parity.py)."""
        self._ops = _op_table() if on else None
        return self._ops is not None

    def set_option(self, name, value):
        """Solver options by name (include/beacon_hip.h: bcn_set_option), e.g. ("conv_plan", 0)."""
        _lib.check(self.lib.bcn_set_option(self.h, name.encode(), int(value)))

    def set_noise_seed(self, seed, replica_offset=0):
        """Envs with inlet noise (burgers, shkadov): step() without an explicit `noise` tensor lets the step kernel draw
        uniform(-sigma, sigma) itself (include/beacon_hip.h: bcn_set_noise) -- keyed by `seed`, the global replica index
        `replica_offset + b`, the replica's count of such steps and the timestep.
        Two things to know: (1) `env.gen` (a torch generator) only feeds draw_noise() and reset_random(); seeding IT does not
        change the noise of step(a) without a `noise` tensor -- this method does.  (2) sigma, seed and offset are kernel
        ARGUMENTS: a graph recorded by capture() keeps the values it was recorded with (only the per-replica draw counters
        live on the device and advance at replay), so re-capture after changing the seed."""
        self.seed, self.replica_offset = int(seed), int(replica_offset)
        _lib.check(self.lib.bcn_set_noise(self.h, float(self.sigma), self.seed, self.replica_offset))

    def get_counters(self):
        """uint64 [B, 4] of the last step, per replica: shader cycles inside the Jacobi loop / in the whole replica, late
        stops of the extrapolating residual plan, repeated timesteps (include/beacon_hip.h: bcn_get_counters)."""
        n = _lib.COUNTER_WORDS
        buf = (C.c_uint64 * (n * self.batch))()
        _lib.check(self.lib.bcn_get_counters_n(self.h, buf, n, self._stream()))     # the sized form: the buffer cannot be overrun
        return np.frombuffer(buf, dtype=np.uint64).reshape(self.batch, n).copy()

    @property
    def kernel_name(self):
        return self.lib.bcn_kernel_name(self.h).decode()

    def check_status(self):
        """Synchronise and raise if any replica reported a solver failure (the reference
        prints and exit(1)s on Poisson non-convergence: rayleigh.py:221-224)."""
        st = self.status.cpu().numpy()
        if (st & _lib.ST_ITMAX).any():
            bad = np.nonzero(st & _lib.ST_ITMAX)[0]
            raise RuntimeError("Exceeded max number of iterations in solver (replicas %s)" % bad[:8].tolist())
        return st

    # -- replica masks ----------------------------------------------------------------------
    def _apply_mask(self, mask):
        """mask: None (all replicas) or a [B] bool/uint8 tensor/array; replicas with 0 are skipped
        by the next reset/step launch (their state and output rows stay as they are)."""
        if mask is None:
            self._mask = None
            _lib.check(self.lib.bcn_set_mask(self.h, None))
            return
        if not torch.is_tensor(mask):
            mask = torch.as_tensor(np.asarray(mask))
        self._mask = mask.to(device=self.device, dtype=torch.uint8).reshape(self.batch).contiguous()
        _lib.check(self.lib.bcn_set_mask(self.h, _ptr(self._mask)))

    # -- Gym surface ------------------------------------------------------------------------
    def reset(self, mask=None):
        """Reset every replica, or only those selected by `mask` (what a trainer does when it calls
        reset() on the one env whose episode ended)."""
        self._apply_mask(mask)
        try:
            self._reset()
        finally:
            if mask is not None:
                self._apply_mask(None)
        return self.obs, None

    def step(self, actions=None, noise=None, mask=None):
        if self._rotate:
            self._next_outputs(carry=mask is not None)
        if mask is None and getattr(self, "_mask", None) is None:
            self._step(actions, noise)            # the common case: no mask now, none set -- nothing to tell the library
        else:
            self._apply_mask(mask)
            try:
                self._step(actions, noise)
            finally:
                if mask is not None:
                    self._apply_mask(None)
        return self.obs, self.rwd, self.done, self.trunc, None

    def capture(self, actions, noise=None, n_steps=None, keep_steps=True):
        """Record step() calls into ONE HIP graph (torch.cuda.CUDAGraph) and return it as a StepGraph: replay()
        relaunches them with a single host call.  step() only enqueues work on the current stream (no host
        synchronisation, no host read), so it can be captured -- on its own, as here, or inside a caller's graph next to
        the policy network.  For the 1D envs at the reference batch sizes the kernel of one step (38 us, burgers B=1024)
        is shorter than the host work of launching it through Python; a graph of n steps runs at the kernel rate.
        `actions` (and `noise`): STATIC device tensors the graph reads at every replay -- [B, ...] for one step, or
        [n_steps, B, ...] for n_steps steps (step k uses actions[k]); overwrite them in place between replays.
        keep_steps=False leaves out the per-step copies of obs / rwd / done / trunc (only the last step's stay, in the
        env's own tensors): every extra graph node costs a few microseconds between two kernels.
        The recorded kernels keep the noise seed / sigma they were captured with (set_noise_seed): change the seed, then
        capture again."""
        return StepGraph(self, actions, noise, n_steps, keep_steps)

    def reset_done(self):
        """Auto-reset: re-initialise the replicas whose last step() returned done (their rows of
        `obs` become the reset observation).  Entirely on the device, no host synchronisation."""
        self._done_mask = self.done.clone()
        return self.reset(mask=self._done_mask)

    def warmup(self, n_steps, actions=None):
        """Equivalent of the reference's init.py generators (rayleigh/init.py:13-28): n uncontrolled action steps
        (zero / repeated action), e.g. to develop the flow on a grid that ships no init_field.dat.  The per-step
        rewards are kept in `self.warmup_rwd` [n_steps, B] (rayleigh: minus the Nusselt history the reference's
        generator plots); episode counters are reset afterwards."""
        rw = []
        for _ in range(int(n_steps)):
            self._step(actions, None)
            rw.append(self.rwd.clone())
        self.warmup_rwd = torch.stack(rw) if rw else torch.empty((0, self.batch), dtype=self.tdtype, device=self.device)
        self.check_status()
        self.set_stp(0)
        return self.get_state()


class StepGraph(object):
    """n step() calls of one VecEnv on static inputs, as a HIP graph (VecEnv.capture).  After replay() the env's own
    obs / rwd / done / trunc hold the last step's results; `obs_seq`, `rwd_seq`, `done_seq`, `trunc_seq` ([n, B, ...])
    hold every step's."""

    def __init__(self, env, actions, noise=None, n_steps=None, keep_steps=True):
        self.env, self.actions, self.noise = env, actions, noise
        self.n = 1 if n_steps is None else int(n_steps)
        seq = n_steps is not None
        n = self.n if keep_steps else 0
        self.obs_seq = torch.empty((n,) + tuple(env.obs.shape), dtype=env.obs.dtype, device=env.device)
        self.rwd_seq = torch.empty((n,) + tuple(env.rwd.shape), dtype=env.rwd.dtype, device=env.device)
        self.done_seq = torch.empty((n,) + tuple(env.done.shape), dtype=env.done.dtype, device=env.device)
        self.trunc_seq = torch.empty((n,) + tuple(env.trunc.shape), dtype=env.trunc.dtype, device=env.device)
        self.graph = torch.cuda.CUDAGraph()
        gen = getattr(env, "gen", None)
        if gen is not None and noise is None:
            self.graph.register_generator_state(gen)      # the env draws its own noise: a graph-safe generator
        torch.cuda.synchronize(env.device)
        with torch.cuda.graph(self.graph):
            for k in range(self.n):
                a = actions[k] if seq else actions
                z = None if noise is None else (noise[k] if seq else noise)
                env._step(a, z)
                if not keep_steps:
                    continue
                self.obs_seq[k].copy_(env.obs)
                self.rwd_seq[k].copy_(env.rwd)
                self.done_seq[k].copy_(env.done)
                self.trunc_seq[k].copy_(env.trunc)

    def replay(self):
        self.graph.replay()
        return self.obs_seq, self.rwd_seq, self.done_seq, self.trunc_seq


# ---------------------------------------------------------------------------------------------
class VecRayleigh(VecEnv):
    """rayleigh/rayleigh.py:16-366.  `init_fields`: [4, nx+2, ny+2] in the reference's [i, j]
    layout (u, v, p, T) -- what load() parses from init_field.dat (:356-362) -- or None
    (init=False: all-zero fields)."""

    def __init__(self, batch, device="cuda:0", dtype="f32", init_fields=None,
                 L=1.0, H=1.0, n_sgts=10, ra=1.0e4):
        self._derive(L, H, n_sgts, ra)
        self._ctor = dict(L=L, H=H, n_sgts=n_sgts, ra=ra)     # a twin of this env (the JIT self-check builds some)
        self._init_np = None if init_fields is None else np.asarray(init_fields, dtype=np.float64)
        super().__init__(batch, device, dtype)
        self._post_init()

    def _derive(self, L=1.0, H=1.0, n_sgts=10, ra=1.0e4):
        self.L, self.H, self.ra, self.n_sgts = L, H, ra, n_sgts
        self.nx, self.ny = int(50 * L), int(50 * H)                       # :26-27
        self.pr, self.Tc, self.Th, self.C = 0.71, -0.5, 0.5, 0.75         # :29-32
        self.dt, self.dt_act, self.t_warmup, self.t_act = 0.01, 2.0, 200.0, 200.0
        self.nx_obs_pts, self.ny_obs_pts, self.n_obs_steps = 4 * int(L), 4 * int(H), 4
        self.dx, self.dy = float(L / self.nx), float(H / self.ny)         # :45-46
        self.ndt_act = int(self.dt_act / self.dt)                         # :48
        self.n_act = int(self.t_act / self.dt_act)                        # :50
        self.n_warmup = int(self.t_warmup / self.dt_act)
        self.nx_sgts = self.nx // n_sgts                                  # :52
        self.n_obs_tot = 3 * self.n_obs_steps * self.nx_obs_pts * self.ny_obs_pts
        self.nx_obs, self.ny_obs = self.nx // self.nx_obs_pts, self.ny // self.ny_obs_pts
        self.tol, self.itmax = 1.0e-8, 300000                             # :414-417
        return self

    def _make_spaces(self):
        self.action_space = spaces.box(-self.C, self.C, (self.n_sgts,))         # rayleigh.py:75-78
        self.observation_space = spaces.sym_box(1.0, self.n_obs_tot)            # :81-86
        return self

    def _post_init(self):
        n_sgts = self.n_sgts
        self._make_spaces()
        self.actions_norm = torch.zeros((self.batch, n_sgts), dtype=self.tdtype, device=self.device)
        self.sweeps = torch.zeros((self.batch, self.ndt_act), dtype=torch.int32, device=self.device)
        self._init_dev = None
        if self._init_np is not None:
            assert self._init_np.shape == (4, self.nx + 2, self.ny + 2)
            # reference arrays are [i, j]; the device layout is [j, i] (x fastest)
            self._init_dev = self._real(np.ascontiguousarray(self._init_np.transpose(0, 2, 1)),
                                        (4, self.ny + 2, self.nx + 2))

    def _create(self):
        c = _lib.RayleighCfg(nx=self.nx, ny=self.ny, ndt_act=self.ndt_act, n_act=self.n_act,
                             n_sgts=self.n_sgts, nx_sgts=self.nx_sgts, nx_obs_pts=self.nx_obs_pts,
                             ny_obs_pts=self.ny_obs_pts, nx_obs=self.nx_obs, ny_obs=self.ny_obs,
                             n_obs_steps=self.n_obs_steps, itmax=self.itmax, dx=self.dx, dy=self.dy,
                             dt=self.dt, pr=self.pr, ra=self.ra, Tc=self.Tc, Th=self.Th, C=self.C,
                             tol=self.tol)
        self.cfg = c
        _lib.check(self.lib.bcn_rayleigh_create(C.byref(c), self.batch, self.cdtype, self.dev_index,
                                                C.byref(self.h)))
        self._attach_plugin(0)
        self._slow_mode_bound(0)

    def set_ndt_act(self, n):
        """Test hook: shorten the action step (the goldens for big grids use ndt_act=5)."""
        self.close()
        self.ndt_act = int(n)
        self.h = C.c_void_p()
        self._create()
        self.sweeps = torch.zeros((self.batch, self.ndt_act), dtype=torch.int32, device=self.device)

    def state_shape(self):
        return (4, self.ny + 2, self.nx + 2)

    def perturbed_conduction_state(self, seed=2024):
        """Start state of a warm-up on a grid without an init file, [4, nx+2, ny+2] in the reference's [i, j] layout:
        u = v = p = 0, T = the conduction profile plus five seeded long-wave perturbations of amplitude <= 0.02 (from
        the reference's all-zero start, rayleigh/init.py:13, an exactly x-uniform state never leaves pure conduction)."""
        rng = np.random.default_rng(seed)
        xm = (np.arange(self.nx + 2) - 0.5) * self.dx
        ym = (np.arange(self.ny + 2) - 0.5) * self.dy
        X, Y = np.meshgrid(xm, ym, indexing="ij")
        T = self.Th + (self.Tc - self.Th) * Y / self.H
        for k in range(1, 6):
            T += 0.02 * rng.uniform(-1, 1) * np.sin(np.pi * Y / self.H) * np.cos(k * np.pi * X / self.L + rng.uniform(0, 6.28))
        T[:, 0] = 0.0
        T[:, -1] = 0.0
        st = np.zeros((4, self.nx + 2, self.ny + 2))
        st[3] = T
        return st

    def develop(self, n_steps=None, seed=2024):
        """The reference's init.py on the device (rayleigh/init.py:13-28: n_warmup uncontrolled action steps, then
        dump): every replica starts from perturbed_conduction_state(seed) and takes n_steps (default n_warmup = 100)
        zero-action steps.  Returns the developed fields of replica 0 as [4, nx+2, ny+2] float64 in the reference's
        layout -- what VecRayleigh(init_fields=...) takes -- and keeps minus the Nusselt history in `warmup_rwd`."""
        st0 = self.perturbed_conduction_state(seed)
        self.reset()
        self.set_state(np.tile(np.ascontiguousarray(st0.transpose(0, 2, 1))[None], (self.batch, 1, 1, 1)))
        zero = torch.zeros((self.batch, self.n_sgts), dtype=self.tdtype, device=self.device)
        st = self.warmup(self.n_warmup if n_steps is None else n_steps, zero)
        return np.ascontiguousarray(st[0].double().cpu().numpy().transpose(0, 2, 1))

    def _reset(self):
        if self._ops is not None:
            return self._ops["rayleigh_reset"](self.h.value, self._init_dev, self.obs)
        _lib.check(self.lib.bcn_rayleigh_reset(self.h, _ptr(self._init_dev), _ptr(self.obs), self._stream()))

    def _step(self, actions, noise=None):
        a = self._real(actions, (self.batch, self.n_sgts))
        self._keep = a
        if self._ops is not None:
            return self._ops["rayleigh_step"](self.h.value, a, self.actions_norm, self.obs, self.rwd, self.done, self.trunc,
                                              self.status, self.sweeps)
        _lib.check(self.lib.bcn_rayleigh_step(self.h, _ptr(a), _ptr(self.actions_norm), _ptr(self.obs),
                                              _ptr(self.rwd), _ptr(self.done), _ptr(self.trunc),
                                              _ptr(self.status), _ptr(self.sweeps), self._stream()))


class VecMixing(VecEnv):
    """mixing/mixing.py:16-378"""

    action_is_int = True

    def __init__(self, batch, device="cuda:0", dtype="f32", L=1.0, H=1.0, re=100.0, pe=10000.0,
                 side=0.5, C0=1.0):
        self._derive(L, H, re, pe, side, C0)
        self._ctor = dict(L=L, H=H, re=re, pe=pe, side=side, C0=C0)
        super().__init__(batch, device, dtype)
        self._make_spaces()
        self.sweeps = torch.zeros((self.batch, self.ndt_act), dtype=torch.int32, device=self.device)

    def _make_spaces(self):
        self.action_space = spaces.discrete(4)                                  # mixing.py:61
        self.observation_space = spaces.sym_box(1.0, self.n_obs_tot)            # :65-70
        return self

    def _derive(self, L=1.0, H=1.0, re=100.0, pe=10000.0, side=0.5, C0=1.0):
        self.L, self.H, self.re, self.pe, self.side, self.C0 = L, H, re, pe, side, C0
        self.nx, self.ny = int(100 * L), int(100 * H)                    # :27-28
        self.nu = 0.01
        self.u_max = re * self.nu / L                                    # :34
        self.dt, self.dt_act, self.t_act = 0.002, 0.5, 50.0
        self.nx_obs_pts, self.ny_obs_pts, self.n_obs_steps = 4 * int(L), 4 * int(H), 4
        self.dx, self.dy = float(L / self.nx), float(H / self.ny)
        self.ndt_act = int(self.dt_act / self.dt)
        self.n_act = int(self.t_act / self.dt_act)
        self.n_obs_tot = 3 * self.n_obs_steps * self.nx_obs_pts * self.ny_obs_pts
        self.nx_obs, self.ny_obs = self.nx // self.nx_obs_pts, self.ny // self.ny_obs_pts
        self.tol, self.itmax = 1.0e-4, 300000                            # :423-426
        self.i_min = math.floor(0.5 * (L - side) / self.dx)              # :90-93
        self.i_max = self.i_min + math.floor(side / self.dx)
        self.j_min = math.floor(0.5 * (H - side) / self.dy)
        self.j_max = self.j_min + math.floor(side / self.dy)
        return self

    def _create(self):
        c = _lib.MixingCfg(nx=self.nx, ny=self.ny, ndt_act=self.ndt_act, n_act=self.n_act,
                           nx_obs_pts=self.nx_obs_pts, ny_obs_pts=self.ny_obs_pts, nx_obs=self.nx_obs,
                           ny_obs=self.ny_obs, n_obs_steps=self.n_obs_steps, itmax=self.itmax,
                           i_min=self.i_min, i_max=self.i_max, j_min=self.j_min, j_max=self.j_max,
                           dx=self.dx, dy=self.dy, dt=self.dt, re=self.re, pe=self.pe, u_max=self.u_max,
                           C0=self.C0, ref_c=(self.side * self.side) / (self.L * self.H) * self.C0,
                           tol=self.tol)
        self.cfg = c
        _lib.check(self.lib.bcn_mixing_create(C.byref(c), self.batch, self.cdtype, self.dev_index,
                                              C.byref(self.h)))
        self._attach_plugin(1)
        self._slow_mode_bound(1)

    def set_ndt_act(self, n):
        self.close()
        self.ndt_act = int(n)
        self.h = C.c_void_p()
        self._create()
        self.sweeps = torch.zeros((self.batch, self.ndt_act), dtype=torch.int32, device=self.device)

    def state_shape(self):
        return (4, self.ny + 2, self.nx + 2)

    def _reset(self):
        if self._ops is not None:
            return self._ops["mixing_reset"](self.h.value, self.obs)
        _lib.check(self.lib.bcn_mixing_reset(self.h, _ptr(self.obs), self._stream()))

    def _step(self, actions, noise=None):
        a = None
        if actions is not None:
            if not torch.is_tensor(actions):
                actions = torch.as_tensor(np.asarray(actions, dtype=np.int64))
            a = actions.to(device=self.device, dtype=torch.int32).reshape(self.batch).contiguous()
        self._keep = a
        if self._ops is not None:
            return self._ops["mixing_step"](self.h.value, a, self.obs, self.rwd, self.done, self.trunc, self.status, self.sweeps)
        _lib.check(self.lib.bcn_mixing_step(self.h, _ptr(a), _ptr(self.obs), _ptr(self.rwd), _ptr(self.done),
                                            _ptr(self.trunc), _ptr(self.status), _ptr(self.sweeps),
                                            self._stream()))


class VecBurgers(VecEnv):
    """burgers/burgers.py:17-227.  `nx` is a kwarg here (a literal 500 in the reference, :26)."""

    needs_noise = True

    def __init__(self, batch, device="cuda:0", dtype="f32", u_target=0.5, amp=10.0, sigma=0.1,
                 ctrl_pos=1.0, L=2.0, nx=500, seed=0):
        self._derive(u_target, amp, sigma, ctrl_pos, L, nx)
        self.seed, self.replica_offset = int(seed), 0
        super().__init__(batch, device, dtype)
        self._make_spaces()
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(self.seed)

    def _make_spaces(self):
        self.action_space = spaces.box(-1.0, 1.0, (1,))                         # burgers.py:54-57
        self.observation_space = spaces.box(np.zeros(self.n_obs_pts), np.ones(self.n_obs_pts), (self.n_obs_pts,))  # :60-65
        return self

    def _derive(self, u_target=0.5, amp=10.0, sigma=0.1, ctrl_pos=1.0, L=2.0, nx=500):
        self.L, self.nx, self.amp, self.sigma, self.u_target = L, nx, amp, sigma, u_target
        self.t_max, self.dt_act, self.n_obs_pts = 10.0, 0.05, 5
        self.dx = float(L / nx)                                          # :36
        self.ctrl_pos = int(ctrl_pos / self.dx)                          # :37
        self.dt = 0.2 * self.dx                                          # :38
        self.ndt_act = int(self.dt_act / self.dt)                        # :40
        self.n_act = int(self.t_max / self.dt_act)                       # :42
        return self

    def _create(self):
        c = _lib.BurgersCfg(nx=self.nx, ndt_act=self.ndt_act, n_act=self.n_act, ctrl_pos=self.ctrl_pos,
                            n_obs_pts=self.n_obs_pts, dx=self.dx, dt=self.dt, amp=self.amp,
                            u_target=self.u_target)
        self.cfg = c
        _lib.check(self.lib.bcn_burgers_create(C.byref(c), self.batch, self.cdtype, self.dev_index,
                                               C.byref(self.h)))
        self.set_noise_seed(self.seed, self.replica_offset)

    def state_shape(self):
        return (3, self.nx)

    def draw_noise(self):
        """An explicit noise tensor of the reference's law, np.random.uniform(-sigma, sigma, 1) (burgers.py:127), from the
        env's torch generator -- for callers that want to see or reuse the draws.  step(a) without `noise` needs none:
        the kernel draws its own (set_noise_seed)."""
        r = torch.rand((self.batch,), generator=self.gen, device=self.device, dtype=self.tdtype)
        return (2.0 * r - 1.0) * self.sigma

    def _reset(self):
        if self._ops is not None:
            return self._ops["burgers_reset"](self.h.value, self.obs)
        _lib.check(self.lib.bcn_burgers_reset(self.h, _ptr(self.obs), self._stream()))

    def _step(self, actions, noise=None):
        a = self._real(actions, (self.batch,))
        nz = None if noise is None else self._real(noise, (self.batch,))     # None: drawn inside the kernel
        self._keep = (a, nz)
        if self._ops is not None:
            return self._ops["burgers_step"](self.h.value, a, nz, self.obs, self.rwd, self.done, self.trunc, self.status)
        _lib.check(self.lib.bcn_burgers_step(self.h, _ptr(a), _ptr(nz), _ptr(self.obs), _ptr(self.rwd),
                                             _ptr(self.done), _ptr(self.trunc), _ptr(self.status),
                                             self._stream()))


class VecShkadov(VecEnv):
    """shkadov/shkadov.py:16-372.  `init_fields`: [2, >=nx] (h_init, q_init) as load() parses
    them (:364-368), or None for the flat film h=q=1."""

    needs_noise = True

    def __init__(self, batch, device="cuda:0", dtype="f32", init_fields=None, L0=150.0, n_jets=5,
                 jet_pos=150.0, jet_space=10.0, delta=0.1, t_act=20.0, seed=0):
        self._derive(L0, n_jets, jet_pos, jet_space, delta, t_act)
        self._init_np = None if init_fields is None else np.asarray(init_fields, dtype=np.float64)
        self.seed, self.replica_offset = int(seed), 0
        super().__init__(batch, device, dtype)
        self._make_spaces()
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(self.seed)
        self._init_dev = None
        if self._init_np is not None:
            self._init_dev = self._real(np.ascontiguousarray(self._init_np[:, :self.nx]), (2, self.nx))

    def _make_spaces(self):
        self.action_space = spaces.box(-1.0, 1.0, (self.n_jets,))               # shkadov.py:96-99
        self.observation_space = spaces.sym_box(1.0, self.n_obs * self.n_jets)  # :105-110
        return self

    def _derive(self, L0=150.0, n_jets=5, jet_pos=150.0, jet_space=10.0, delta=0.1, t_act=20.0):
        self.L = L0 + jet_space * (n_jets + 2)                           # :32
        self.nx = int(5 * self.L)                                        # :33
        self.dt, self.dt_act, self.t_act = 0.001, 0.05, t_act
        self.n_warmup_ref = int(200.0 / self.dt_act)                     # :36,62 t_warmup = 200 -> 4000 action steps (init.py)
        self.sigma, self.delta, self.n_jets, self.jet_amp = 5.0e-4, delta, n_jets, 5.0
        self.eps, self.blowup_rwd, self.h_max = 1.0e-8, -1.0, 5.0
        self.dx = float(self.L / self.nx)                                # :56
        self.ndt_act = int(self.dt_act / self.dt)
        self.n_act = int(t_act / self.dt_act)
        self.n_interp = int(0.02 / self.dt)                              # :63
        self.jet_pos = int(jet_pos / self.dx)                            # :64
        self.jet_hw = int(2.0 / self.dx)                                 # :65
        self.jet_space = int(jet_space / self.dx)                        # :67
        self.l_rwd = int(10.0 / self.dx)                                 # :70
        self.n_obs = int(10.0)                                           # :71
        self.l_obs = int(10.0 / self.dx)                                 # :72
        self.obs_stride = int(1.0 / self.dx)                             # :245
        return self

    def _create(self):
        c = _lib.ShkadovCfg(nx=self.nx, ndt_act=self.ndt_act, n_act=self.n_act, n_jets=self.n_jets,
                            jet_pos=self.jet_pos, jet_hw=self.jet_hw, jet_space=self.jet_space,
                            l_obs=self.l_obs, l_rwd=self.l_rwd, n_obs=self.n_obs,
                            obs_stride=self.obs_stride, n_interp=self.n_interp, dx=self.dx, dt=self.dt,
                            delta=self.delta, jet_amp=self.jet_amp, eps=self.eps,
                            h_blow=5.0 * self.h_max, blowup_rwd=self.blowup_rwd)
        self.cfg = c
        _lib.check(self.lib.bcn_shkadov_create(C.byref(c), self.batch, self.cdtype, self.dev_index,
                                               C.byref(self.h)))
        self.set_noise_seed(self.seed, self.replica_offset)

    def state_shape(self):
        return (4, self.nx)

    def draw_noise(self):
        r = torch.rand((self.batch, self.ndt_act), generator=self.gen, device=self.device, dtype=self.tdtype)
        return (2.0 * r - 1.0) * self.sigma

    def reset_random(self, rand_steps=400, n_steps=None):
        """reset() followed by a per-replica random number of uncontrolled steps, the batched form
        of shkadov.reset with rand_init (shkadov.py:119-123: n = random.randint(0, rand_steps)).
        n_steps: optional explicit int tensor [B]; drawn on the device otherwise."""
        self.reset()
        if n_steps is None:
            n_steps = torch.randint(0, rand_steps + 1, (self.batch,), generator=self.gen, device=self.device)
        n_steps = torch.as_tensor(n_steps).to(self.device)
        self.n_rand = n_steps
        for i in range(int(n_steps.max().item())):
            self.step(None, None, mask=(n_steps > i))
        self.set_stp(0)
        return self.obs, None

    def _reset(self):
        if self._ops is not None:
            return self._ops["shkadov_reset"](self.h.value, self._init_dev, self.obs)
        _lib.check(self.lib.bcn_shkadov_reset(self.h, _ptr(self._init_dev), _ptr(self.obs), self._stream()))

    def _step(self, actions, noise=None):
        a = self._real(actions, (self.batch, self.n_jets))
        nz = None if noise is None else self._real(noise, (self.batch, self.ndt_act))   # None: drawn inside the kernel
        self._keep = (a, nz)
        if self._ops is not None:
            return self._ops["shkadov_step"](self.h.value, a, nz, self.obs, self.rwd, self.done, self.trunc, self.status)
        _lib.check(self.lib.bcn_shkadov_step(self.h, _ptr(a), _ptr(nz), _ptr(self.obs), _ptr(self.rwd),
                                             _ptr(self.done), _ptr(self.trunc), _ptr(self.status),
                                             self._stream()))


class VecSloshing(VecEnv):
    """sloshing/sloshing.py:16-320.  `init_fields`: [2, nx+2] (h_init, q_init incl. ghosts)."""

    def __init__(self, batch, device="cuda:0", dtype="f32", init_fields=None, L=2.5, amp=5.0,
                 alpha=0.0005, g=9.81):
        self._derive(L, amp, alpha, g)
        self._init_np = None if init_fields is None else np.asarray(init_fields, dtype=np.float64)
        super().__init__(batch, device, dtype)
        self._make_spaces()
        self._init_dev = None
        if self._init_np is not None:
            self._init_dev = self._real(self._init_np, (2, self.nx + 2))

    def _make_spaces(self):
        self.action_space = spaces.box(-1.0, 1.0, (1,))                         # sloshing.py:73-76
        self.observation_space = spaces.sym_box(1.0, self.n_obs)                # :81-86
        return self

    def _derive(self, L=2.5, amp=5.0, alpha=0.0005, g=9.81):
        self.L, self.amp, self.alpha, self.g = L, amp, alpha, g
        self.nx = int(80 * L)                                            # :24
        self.dt, self.dt_act, self.t_warmup, self.t_act = 0.001, 0.05, 2.0, 10.0
        self.dx = float(L / self.nx)
        self.ndt_act = int(self.dt_act / self.dt)
        self.n_act = int(self.t_act / self.dt_act)
        self.n_warmup = int(self.t_warmup / self.dt_act)
        self.n_interp = int(0.01 / self.dt)                              # :48
        self.n_obs = self.nx // 2 + (1 if self.nx % 2 else 0)            # :39
        return self

    def _create(self):
        c = _lib.SloshingCfg(nx=self.nx, ndt_act=self.ndt_act, n_act=self.n_act, n_interp=self.n_interp,
                             dx=self.dx, dt=self.dt, g=self.g, amp=self.amp, alpha=self.alpha)
        self.cfg = c
        _lib.check(self.lib.bcn_sloshing_create(C.byref(c), self.batch, self.cdtype, self.dev_index,
                                                C.byref(self.h)))

    def state_shape(self):
        return (4, self.nx + 2)

    @staticmethod
    def signal(t, dt):
        """Excitation used by the reference's warm-up generator (sloshing.py:134-138)."""
        return 0.5 * (np.cos(np.pi * t) + 3.0 * np.cos(4.0 * np.pi * t))

    def _reset(self):
        if self._ops is not None:
            return self._ops["sloshing_reset"](self.h.value, self._init_dev, self.obs)
        _lib.check(self.lib.bcn_sloshing_reset(self.h, _ptr(self._init_dev), _ptr(self.obs), self._stream()))

    def _step(self, actions, noise=None):
        a = self._real(actions, (self.batch,))
        self._keep = a
        if self._ops is not None:
            return self._ops["sloshing_step"](self.h.value, a, self.obs, self.rwd, self.done, self.trunc, self.status)
        _lib.check(self.lib.bcn_sloshing_step(self.h, _ptr(a), _ptr(self.obs), _ptr(self.rwd), _ptr(self.done),
                                              _ptr(self.trunc), _ptr(self.status), self._stream()))
