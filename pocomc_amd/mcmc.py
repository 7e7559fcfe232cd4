"""MCMC kernels -- host-side mirror of ``pocomc/mcmc.py`` behind the same contract

    results = kernel(state_dict, function_dict, option_dict)

(``pocomc/sampler.py:568-631``).  The particle state lives in HBM for the whole
call; per step the device runs  propose -> flow inverse -> scaler inverse  and
accept -> reductions  (``include/pocomc_amd.h``), the host only evaluates the
user's prior / likelihood black boxes on the compacted finite rows
(``mcmc.py:100-121``) and the scalar adaptation / stopping logic
(``mcmc.py:152-180``).

``StepEngine`` is the reusable object (bench.py drives it directly); the four
functions ``preconditioned_pcn | preconditioned_rwm | pcn | rwm`` wrap it with
the reference's signature.

Multi-GPU: the walkers are row-sharded, one process per GPU.  The only
exchange per step is the all-reduce of ``D+4`` float64 sums (RCCL over xGMI via
``torch.distributed``) that makes sigma, mu and the stop decision identical on
every rank.
"""
from __future__ import annotations

import ctypes as C
import os
import time

import numpy as np
import torch

from . import _lib

KINDS = ("preconditioned_pcn", "preconditioned_rwm", "pcn", "rwm")
PMC_KIND_TPCN, PMC_KIND_RWM = 0, 1


# --------------------------------------------------------------------------
# host-side scalar logic, shared with the CPU tests of the sharded path
# --------------------------------------------------------------------------
PMC_ADAPT_TPCN, PMC_ADAPT_PRWM, PMC_ADAPT_RWM, PMC_ADAPT_MU = 1, 2, 3, 8          # include/pocomc_amd.h


class Adaptation:
    """sigma / mu adaptation and the plateau stop of one kernel call
    (``mcmc.py:152-180`` and the variants ``:314-336``, ``:476-502``, ``:627-650``).

    Fed with GLOBAL sums (already all-reduced), so every rank takes the same
    decisions."""

    def __init__(self, kind, n_dim, n_total, n_steps, n_max, sigma0, mu0, logp2_0):
        self.kind = kind
        self.tpcn = kind in ("preconditioned_pcn", "pcn")
        self.D = n_dim
        self.N = n_total
        self.n_steps = n_steps
        self.n_max = n_max
        self.sigma = np.minimum(sigma0, 0.99) if self.tpcn else sigma0        # mcmc.py:54
        self.mu = None if mu0 is None else np.array(mu0, dtype=np.float64)
        self.logp2_val = logp2_0
        self.cnt = 0
        self.i = 0
        self.mean_alpha = 0.0

    def coefficients(self):
        """What the update that follows the step now in flight multiplies with: ``(mode, c_sigma, c_mu, cap)`` for
        the device-side copy of this update (``pmc_step_t.adapt_*``).  Same expressions as in :meth:`update`, so
        host and device hold the same sigma and mu bit for bit."""
        i = self.i + 1
        cap = min(2.38 / self.D ** 0.5, 0.99)
        if self.tpcn:
            mode, c = PMC_ADAPT_TPCN, 1 / (i + 1) ** 0.75
        elif self.kind == "preconditioned_rwm":
            mode, c = PMC_ADAPT_PRWM, 1 / (i + 1)
        else:
            mode, c = PMC_ADAPT_RWM, 1 / (i + 1)
        if self.kind == "preconditioned_pcn":
            mode |= PMC_ADAPT_MU
        return mode, float(c), 1.0 / (i + 1.0), cap

    def update(self, sums):
        """``sums`` = [sum alpha, sum(logl+logp), sum(logl+logp+logdetj), n_accept, sum theta_j...].
        Returns True when the loop must stop."""
        self.i += 1
        i, D, N = self.i, self.D, self.N
        # (Python floats: the same IEEE double operations as the reference's numpy scalars, without their dispatch cost --
        #  this runs once per step on the driver thread)
        mean_alpha = float(sums[0]) / N
        self.mean_alpha = mean_alpha
        cap = min(2.38 / D ** 0.5, 0.99)
        sigma = float(self.sigma)
        if self.tpcn:
            self.sigma = abs(min(sigma + 1 / (i + 1) ** 0.75 * (mean_alpha - 0.234), cap))
        elif self.kind == "preconditioned_rwm":
            self.sigma = sigma + 1 / (i + 1) * (mean_alpha - 0.234)
        else:
            self.sigma = abs(sigma + 1 / (i + 1) * (mean_alpha - 0.234))
        if self.kind == "preconditioned_pcn":
            # np.mean of the float32 theta array is a float32 (mcmc.py:156)
            mean_theta = (np.asarray(sums[4:4 + D]) / N).astype(np.float32)
            self.mu = self.mu + 1.0 / (i + 1.0) * (mean_theta - self.mu)
        new = float(sums[1] if self.tpcn else sums[2]) / N
        if new > self.logp2_val:
            self.cnt = 0
            self.logp2_val = new
        else:
            self.cnt += 1
            ratio = (2.38 / D ** 0.5) / self.sigma if self.sigma != 0.0 else float("inf")       # (numpy: division by zero -> inf)
            if self.kind == "preconditioned_rwm":
                ratio = min(1.0, ratio)
            if self.cnt >= self.n_steps * ratio ** 2.0:
                return True
        return i >= self.n_max


# How long the driver thread spins on a completion word before it gives up (seconds).  In the sharded pipelined step the
# word sits behind the all-reduce, i.e. behind the slowest rank's host likelihood: a run with an expensive or imbalanced
# likelihood raises it through option_dict["wait_timeout"] / PMC_WAIT_TIMEOUT.  It is a property of the engine
# (``StepEngine.wait_timeout``), never of the process; values <= 0 or above the cap mean the cap (a dead peer or a hung
# device must surface as an error, not as a silent spin).
WAIT_TIMEOUT_DEFAULT_S = float(os.environ.get("PMC_WAIT_TIMEOUT", "600"))
WAIT_TIMEOUT_CAP_S = 7 * 24 * 3600.0


def _wait_timeout(value):
    v = WAIT_TIMEOUT_DEFAULT_S if value is None else float(value)
    return WAIT_TIMEOUT_CAP_S if (v <= 0.0 or v > WAIT_TIMEOUT_CAP_S) else v

_POOLS = {}
_PINNED_FREE = {}


def _pinned_take(shape, dtype):
    """A pinned host tensor of this shape / dtype from the free list (contents undefined), or a new one."""
    free = _PINNED_FREE.get((shape, dtype))
    if free:
        return free.pop()
    return torch.empty(*shape, dtype=dtype).pin_memory()


def _pinned_give(t):
    """Back to the free list (the engine that held it has synchronised with the device)."""
    free = _PINNED_FREE.setdefault((tuple(t.shape), t.dtype), [])
    if len(free) < 8:
        free.append(t)


def _host_pool(n_threads, cores=None):
    """Thread pool for the host black boxes (numpy releases the GIL in its loops); one pool per (size, cores) for the
    process.  ``cores``: pin worker i to cores[i] -- threads created by a pinned driver thread would otherwise
    inherit its one-core affinity mask."""
    import itertools
    import os
    from concurrent.futures import ThreadPoolExecutor
    key = (int(n_threads), None if cores is None else tuple(cores))
    if key not in _POOLS:
        counter = itertools.count()

        def pin():
            if cores is not None:
                try:
                    os.sched_setaffinity(0, {cores[next(counter) % len(cores)]})
                except OSError:
                    pass
        pool = ThreadPoolExecutor(max_workers=max(1, int(n_threads)), initializer=pin)
        list(pool.map(lambda _: time.sleep(0.01), range(max(1, int(n_threads)))))      # start every worker now
        _POOLS[key] = pool
    return _POOLS[key]


def allreduce_sums(sums_dev, group=None):
    """Sum the per-shard reductions over all ranks (no-op without a process group) IN RANK ORDER: beyond two ranks the
    shards are all-gathered and added ((r0 + r1) + r2) + ..., so the bits of sigma, mu and of the stop rule do not depend
    on the backend's reduction algorithm (gloo's ring, RCCL's tree) and equal the library's own exchange
    (``pmc_comm_adapt_update`` adds the ranks' mailbox slots in the same order).  D + 4 doubles per rank: the gather costs
    what the all-reduce costs."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        world = dist.get_world_size(group)
        if world == 2:
            dist.all_reduce(sums_dev, op=dist.ReduceOp.SUM, group=group)      # (a + b: the same double in either order)
        else:
            parts = [torch.empty_like(sums_dev) for _ in range(world)]
            dist.all_gather(parts, sums_dev, group=group)
            acc = parts[0].clone()
            for p_ in parts[1:]:
                acc += p_
            sums_dev.copy_(acc)
    return sums_dev


_COMMS = {}


def _group_key(group):
    """What identifies a process group across its lifetime: the global ranks of its members (``id(group)`` can be recycled
    for another subgroup once a group is gone)."""
    import torch.distributed as dist
    if group is None:
        return ("world", dist.get_world_size())
    try:
        return tuple(dist.get_process_group_ranks(group))
    except Exception:
        return ("id", id(group))


def drop_comms():
    """Destroy every communicator of this process (their mailboxes, mappings and shared-memory objects); the next sharded
    step builds new ones.  Call before ``dist.destroy_process_group()`` in a long-lived process."""
    from . import _lib as L
    lib = L.load()
    for h, _ in _COMMS.values():
        if h:
            lib.pmc_comm_destroy(h)
    _COMMS.clear()


def small_comm(lib, group, width):
    """The library's own all-reduce for the D + 4 sums of a sharded step (``pmc_comm_*``: a mailbox per rank, sums in rank
    order, so every rank holds the same bits): one communicator per process group and process, created on first use -- the
    64-byte handles travel through ``torch.distributed.all_gather_object`` (any backend), everything after that is device
    to device.  Two kinds of mailbox, tried in this order (``PMC_COMM_MAILBOX=device|host`` fixes one): uncached HBM shared
    through hipIpc handles (xGMI peer stores), then pinned host memory in POSIX shared memory (PCIe).  One node, one
    process per GPU, <= 8 ranks; returns None where that does not hold, where neither kind passes its self-test exchange,
    or with ``PMC_C_ALLREDUCE=0`` (the step then exchanges through ``torch.distributed``)."""
    import torch.distributed as dist
    if os.environ.get("PMC_C_ALLREDUCE", "1") == "0":
        return None
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if world > 8:
        return None
    key = (_group_key(group), world, rank)
    c = _COMMS.get(key)
    if c is not None and c[1] >= width:
        return c[0]
    w = max(int(width), 260)                          # (D <= 256: one communicator serves every engine of the group)
    want = os.environ.get("PMC_COMM_MAILBOX", "")
    kinds = [k for k in ("device", "host") if want in ("", k)]
    h = None
    for kind in kinds:
        h = _try_comm(lib, group, world, rank, w, kind)
        if h:
            break
    _COMMS[key] = (h, w if h else 1 << 30)
    return h


def _try_comm(lib, group, world, rank, w, kind):
    import torch.distributed as dist
    h = (lib.pmc_comm_create if kind == "device" else lib.pmc_comm_create_host)(rank, world, w)
    ok = bool(h)
    buf = (C.c_ubyte * 64)()
    if ok:
        ok = lib.pmc_comm_handle(h, buf) == 0
    mine = (bytes(buf), ok, os.uname().nodename)
    allh = [None] * world
    dist.all_gather_object(allh, mine, group=group)
    if not all(a[1] for a in allh) or len({a[2] for a in allh}) != 1:
        if h:
            lib.pmc_comm_destroy(h)
        return None
    blob = b"".join(a[0] for a in allh)
    good = lib.pmc_comm_connect(h, blob) == 0
    flags = [None] * world
    dist.all_gather_object(flags, good, group=group)
    if all(flags):
        lib.pmc_comm_unlink(h)                           # (host mailboxes: every rank has mapped every segment, the names can go)
        # one exchange with known values before any step depends on the mailboxes: rank r sends r + 1 in every word (a
        # mapping that opened but whose stores do not arrive shows up here, as a timeout or a wrong sum, on every rank
        # alike -- the step then takes the next kind of mailbox, or torch.distributed)
        dev = torch.device("cuda", torch.cuda.current_device())
        part = torch.full((8,), float(rank + 1), dtype=torch.float64, device=dev)
        tot = torch.zeros(8, dtype=torch.float64, device=dev)
        parts = (C.c_void_p * 1)(part.data_ptr())
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = lib.pmc_comm_adapt_update(h, parts, 1, 4, tot.data_ptr(), None, None, 0, 0.0, 0.0, 0.0, 1.0, None, 10.0, stream)
        torch.cuda.synchronize(dev)
        good = rc == 0 and bool((tot == world * (world + 1) / 2).all().item())
        dist.all_gather_object(flags, good, group=group)
    if not all(flags):
        lib.pmc_comm_destroy(h)
        return None
    return h


# --------------------------------------------------------------------------
class StepEngine:
    """Device-resident state + buffers of one MCMC kernel call."""

    def __init__(self, kind, n, n_dim, flow, scaler, device=None, group=None, shard_offset=0, seed=0,
                 x_order="C"):
        """``x_order='F'`` hands the host callbacks x' as an (n, D) Fortran-ordered array (same
        values; numpy's inner loops then run over the n walkers instead of over D)."""
        assert kind in KINDS
        self.kind = kind
        self.pre = kind.startswith("preconditioned")
        self.tpcn = kind in ("preconditioned_pcn", "pcn")
        self.n, self.D = int(n), int(n_dim)
        self.lib = _lib.load()
        self.device = torch.device(device) if device is not None else _lib.require_gpu()
        self.flow = flow
        self.scaler = scaler
        self.group = group
        self.seed = int(seed) & (2 ** 64 - 1)
        self.offset = int(shard_offset)
        n, D, dev = self.n, self.D, self.device
        f64 = lambda *s: torch.empty(*s, dtype=torch.float64, device=dev)
        f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        # current state
        self.theta32 = f32(n, D) if self.pre else None
        self.u, self.x = f64(n, D), f64(n, D)
        self.logdetj, self.logl, self.logp = f64(n), f64(n), f64(n)
        self.ldjf = f32(n) if self.pre else None
        # proposal
        self.p_theta64 = f64(n, D)
        self.p_theta32 = f32(n, D) if self.pre else None
        self.p_u32 = f32(n, D) if self.pre else None
        self.p_ldjf = f32(n) if self.pre else None
        self.p_u, self.p_x = f64(n, D), f64(n, D)
        self.p_logdetj, self.p_logl, self.p_logp = f64(n), f64(n), f64(n)
        self.p_fin = torch.empty(n, dtype=torch.int32, device=dev)
        self.quad, self.p_quad = (f64(n), f64(n)) if self.tpcn else (None, None)
        self.alpha = f64(n)
        self.accept = torch.empty(n, dtype=torch.int32, device=dev)
        self.sums = f64(D + 4)
        self.ws = torch.zeros(max(int(self.lib.pmc_accept_workspace_bytes(n, D)), 8), dtype=torch.uint8, device=dev)   # zeroed once: pmc_accept_armed
        # geometry
        self.mu_d, self.inv_cov_d, self.chol_d = f64(D), f64(D, D), f64(D, D)
        # replay variates
        self.r_gamma, self.r_normal, self.r_uniform = f64(n), f64(n, D), f64(n)
        # pinned host mirrors
        # pinned host mirrors come from a process-wide free list: page-locking costs ~1 ms per buffer, and the
        # Sampler builds an engine per kernel call
        self._pins = []

        def pin(*s, dt=torch.float64):
            t = _pinned_take(tuple(int(v) for v in s), dt)
            self._pins.append(t)
            return t
        assert x_order in ("C", "F")
        self.x_order = x_order
        self.p_xT = f64(D, n) if x_order == "F" else None
        self.h_x, self.h_fin = (pin(D, n) if x_order == "F" else pin(n, D)), pin(n, dt=torch.int32)
        self.h_logl, self.h_logp = pin(n), pin(n)
        self.h_sums = pin(D + 4)
        self.h_accept = pin(n, dt=torch.int32)
        self.h_mu = pin(D)
        self._np_x = self.h_x.numpy().T if x_order == "F" else self.h_x.numpy()
        self._np_fin = self.h_fin.numpy()
        self._np_logl, self._np_logp = self.h_logl.numpy(), self.h_logp.numpy()
        self._np_mu, self._np_sums = self.h_mu.numpy(), self.h_sums.numpy()
        self.scaler_desc = scaler.device_descriptor()
        self._state = _lib.pmc_state_t(
            theta32=self.theta32.data_ptr() if self.pre else None, u=self.u.data_ptr(), x=self.x.data_ptr(),
            logdetj=self.logdetj.data_ptr(), logl=self.logl.data_ptr(), logp=self.logp.data_ptr(),
            logdetj_flow=self.ldjf.data_ptr() if self.pre else None)
        self._prop = _lib.pmc_proposal_t(
            theta64=self.p_theta64.data_ptr() if self.pre else None, u=self.p_u.data_ptr(), x=self.p_x.data_ptr(),
            logdetj=self.p_logdetj.data_ptr(), logl=self.p_logl.data_ptr(), logp=self.p_logp.data_ptr(),
            logdetj_flow=self.p_ldjf.data_ptr() if self.pre else None,
            quad=self.quad.data_ptr() if self.tpcn else None, quad_prop=self.p_quad.data_ptr() if self.tpcn else None)
        self._step = _lib.pmc_step_t(
            kind=PMC_KIND_TPCN if self.tpcn else PMC_KIND_RWM, preconditioned=int(self.pre), n=n, D=D,
            inverse_algo=flow.inverse_algo if self.pre else 0,
            maf=C.cast(C.pointer(flow._desc), C.c_void_p) if self.pre else None,
            scaler=C.cast(C.pointer(self.scaler_desc), C.c_void_p), cur=self._state,
            mu=self.mu_d.data_ptr(), inv_cov=self.inv_cov_d.data_ptr(), chol=self.chol_d.data_ptr(),
            p_theta64=self.p_theta64.data_ptr(), p_theta32=self.p_theta32.data_ptr() if self.pre else None,
            p_u32=self.p_u32.data_ptr() if self.pre else None, p_ldjf=self.p_ldjf.data_ptr() if self.pre else None,
            p_u=self.p_u.data_ptr(), p_x=self.p_x.data_ptr(),
            p_xT=self.p_xT.data_ptr() if self.p_xT is not None else None, p_logdetj=self.p_logdetj.data_ptr(),
            p_fin=self.p_fin.data_ptr(), quad=self.quad.data_ptr() if self.tpcn else None,
            p_quad=self.p_quad.data_ptr() if self.tpcn else None, p_logl=self.p_logl.data_ptr(),
            p_logp=self.p_logp.data_ptr(), alpha=self.alpha.data_ptr(), accept=self.accept.data_ptr(),
            sums=self.sums.data_ptr(), ws=self.ws.data_ptr(),
            h_mu=self.h_mu.data_ptr() if self.tpcn else None, h_x=self.h_x.data_ptr(), h_fin=self.h_fin.data_ptr(),
            h_logl=self.h_logl.data_ptr(), h_logp=self.h_logp.data_ptr(), h_sums=self.h_sums.data_ptr(),
            h_accept=self.h_accept.data_ptr(),
            no_fuse=int(os.environ.get("PMC_NO_FUSE", "0")))     # A/B: bit 0 separate proposal / inverse, bit 1 separate scaler
        self._rng_fast = _lib.pmc_rng_t(gamma=None, normal=None, uniform=None, seed=self.seed, step=0,
                                        offset=self.offset)
        # throughput mode: the Philox variates of step k+1 are generated behind step k's kernels, while the host
        # evaluates the likelihood (pmc_step_t.rng_normal ...); two buffer sets, used alternately
        self.rng_prefill = True
        self._pf_normal = [f64(n, D), f64(n, D)]
        self._pf_gamma = [f64(n), f64(n)]
        self._pf_uniform = [f64(n), f64(n)]
        self._rng_ready = C.c_int64(-1)
        self._ev_pre = self.lib.pmc_event_create()
        for b in range(2):
            self._step.rng_normal[b] = self._pf_normal[b].data_ptr()
            self._step.rng_gamma[b] = self._pf_gamma[b].data_ptr()
            self._step.rng_uniform[b] = self._pf_uniform[b].data_ptr()
        self._step.ev_pre_done = self._ev_pre
        # completion words the kernels store to pinned host memory (host_direct): the driver thread spins on
        # them instead of waking up through the runtime
        self.spin_wait = True
        self.h_done = pin(2, dt=torch.int64)
        self.h_done.zero_()
        self._done_ticket = torch.zeros(1, dtype=torch.int32, device=dev)
        # "every row is clean" word of the fused pre-step (pmc_step_t.h_clean): rows with a non-finite x' or logp'
        self.h_clean = pin(1, dt=torch.int64)
        self.h_clean.fill_(-1)
        self._np_clean = self.h_clean.numpy()
        self._clean_count = torch.zeros(1, dtype=torch.int32, device=dev)
        # rows that do not reach the likelihood (x' or logp' not finite): their HOST rows of x' carry the walker's current x
        # (pmc_step_t.fill_rejected), so that up to fill_rejected_max * n such rows cost a few wasted evaluations instead of
        # the gather x'[mask] of mcmc.py:117 (280 us for 6.5e3 x 50 doubles); their logl' is -inf either way (:118-121)
        self.fill_rejected = True
        self.fill_rejected_max = 0.05
        self._direct_now = False
        self._pre_cfg = None     # switches the composite pre-step's struct fields were last written for
        # adaptation on the device (pmc_step_t.adapt_state): {sigma, cn_a, mu[D]}; see run_pipelined
        self.adapt_state = f64(D + 2)
        self._h_adapt = pin(D + 2)
        self.device_adapt = False
        self.prior_desc = None   # pmc_prior_t when Prior.logpdf runs on the device (set_device_prior)
        self.composite = True    # one C call before / after the host black boxes (pmc_step_pre / _post)
        # x_order 'F' on the composite path: the scaler kernel writes x', the finite mask and logp' straight
        # into the pinned host buffers (and the proposal reads mu from one) -- no copy operations in the pre-step
        self.host_direct = (x_order == "F")
        self._post_uploads = False
        self.step_idx = 0
        self.host_threads = 1    # >1: evaluate the black boxes on row chunks, this thread + (host_threads - 1) pool threads
        self.host_cores = None   # optional list of cores the pool's threads are pinned to (one each)
        self.prefetcher = None   # handle of pmc_prefetcher_create (shared by the lanes of a walker set), see host_prefetch()
        self._pool = None
        self.stream = None       # torch.cuda.Stream of the composite path (LanedEngine); None: the current one
        self.events = None       # bench.py: list of per-step HIP event tuples when not None
        self.host_timers = None  # bench.py: dict of accumulated host seconds when not None
        self.wait_timeout = _wait_timeout(None)      # seconds the driver thread spins on a completion word before it raises

    def __del__(self):
        if getattr(self, "_recycle", False):       # (only after the owner synchronised with the device: mcmc._run)
            for t in getattr(self, "_pins", ()):
                _pinned_give(t)
        self._pins = []
        ev = getattr(self, "_ev_pre", None)
        if ev:
            try:
                self.lib.pmc_event_destroy(ev)
            except Exception:
                pass
            self._ev_pre = None

    def _ev(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream())
        return e

    # ---------------------------------------------------------------- setup
    def load_state(self, u, x, logdetj, logl, logp):
        """numpy arrays or (device) tensors: a walker set that already lives in HBM (the Sampler's pool) is copied
        device to device."""
        up = lambda a: (a.to(torch.float64) if isinstance(a, torch.Tensor)
                        else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)))
        self.u.copy_(up(u)); self.x.copy_(up(x))
        self.logdetj.copy_(up(logdetj)); self.logl.copy_(up(logl)); self.logp.copy_(up(logp))
        if self.pre:
            # theta, logdetj_flow = flow.forward(u)  through tools.py:336-340 (float32, sign flipped)
            u32 = self.u.to(torch.float32)
            with torch.cuda.device(self.device):
                _lib.check(self.lib.pmc_maf_forward(C.byref(self.flow._desc), _lib.ptr(u32), _lib.ptr(self.theta32),
                                                    _lib.ptr(self.ldjf), None, self.n, _lib.stream_handle()),
                           "pmc_maf_forward")
            self.ldjf.neg_()

    def set_geometry(self, mu=None, cov=None):
        """tpCN: ``t_mean, t_cov`` (``mcmc.py:63-68``); RWM: ``normal_cov`` (``:237-238``)."""
        cov = np.asarray(cov, dtype=np.float64)
        chol = np.linalg.cholesky(cov)
        self.chol_d.copy_(torch.from_numpy(np.ascontiguousarray(chol)))
        if self.tpcn:
            self.inv_cov_d.copy_(torch.from_numpy(np.ascontiguousarray(np.linalg.inv(cov))))
            self.set_mu(mu)

    def set_device_prior(self, prior):
        """Evaluate ``prior.logpdf`` (a product of uniform / normal ``scipy.stats`` factors) on the
        device right after the scaler instead of on the host.  Returns False if the prior has a factor
        the device does not know (then it stays a host callback)."""
        desc = prior.device_descriptor(self.device) if hasattr(prior, "device_descriptor") else None
        if desc is None:
            return False
        self._prior_obj = prior                      # keeps the descriptor's tensors alive
        self.prior_desc = desc
        self._step.prior = C.cast(C.pointer(desc), C.c_void_p)
        self._step.h_logp_out = self.h_logp.data_ptr()
        return True

    def set_mu(self, mu):
        # pinned staging + async copy; the previous upload was consumed by a kernel that has
        # completed (accept_reduce synchronises) before the host buffer is rewritten
        self._np_mu[:] = mu
        if not (self.composite and self.events is None):
            self.mu_d.copy_(self.h_mu, non_blocking=True)      # the composite pre-step uploads h_mu itself

    # ----------------------------------------------------------------- step
    def _rng(self, replay):
        if replay is not None:
            if self.tpcn:
                self.r_gamma.copy_(torch.from_numpy(np.ascontiguousarray(replay["gamma"], dtype=np.float64)))
            self.r_normal.copy_(torch.from_numpy(np.ascontiguousarray(replay["z"], dtype=np.float64)))
            self.r_uniform.copy_(torch.from_numpy(np.ascontiguousarray(replay["u"], dtype=np.float64)))
            return _lib.pmc_rng_t(gamma=self.r_gamma.data_ptr() if self.tpcn else None,
                                  normal=self.r_normal.data_ptr(), uniform=self.r_uniform.data_ptr(),
                                  seed=0, step=self.step_idx, offset=self.offset)
        return _lib.pmc_rng_t(gamma=None, normal=None, uniform=None, seed=self.seed, step=self.step_idx,
                              offset=self.offset)

    def can_pipeline(self):
        """Adaptation on the device + the next pre-step enqueued behind the accept: composite path with the
        kernels reading / writing pinned host memory themselves."""
        return bool(self.composite and self.events is None and self.host_direct and self.x_order == "F"
                    and self.spin_wait and self.D <= 256)

    def adapt_upload(self, sigma, mu=None):
        """Start value of the device-side adaptation state (sigma, (1-sigma^2)^0.5, mu)."""
        a = self._h_adapt.numpy()
        a[0] = sigma
        a[1] = (1.0 - sigma ** 2.0) ** 0.5 if self.tpcn else 0.0
        a[2:] = 0.0 if mu is None else mu
        self.adapt_state.copy_(self._h_adapt, non_blocking=True)
        self._step.adapt_state = self.adapt_state.data_ptr()
        self.device_adapt = True

    def propose(self, sigma, nu=0.0, replay=None, step=None):
        """propose -> flow inverse -> scaler inverse, then start the D2H of x'.  ``step``: the step number the
        launch belongs to when it is enqueued ahead of time (run_pipelined), default: the current one."""
        lib, n, D = self.lib, self.n, self.D
        if self.composite and self.events is None:
            if replay is not None:
                self._rng_cur = self._rng(replay)
            else:
                self._rng_fast.step = self.step_idx if step is None else int(step)
                self._rng_cur = self._rng_fast
            self._step.adapt_mode = 1 if self.device_adapt else 0         # (pre: any non-zero mode = read the state)
            self._configure_step()
            if self.device_adapt:
                sigma, cn_a = 0.0, 0.0                      # the kernels read adapt_state instead
            else:
                cn_a = float((1.0 - sigma ** 2.0) ** 0.5) if self.tpcn else 0.0    # mcmc.py:85
            self._stream = self.stream.cuda_stream if self.stream is not None else _lib.stream_handle()
            _lib.check(lib.pmc_step_pre(C.byref(self._step), C.byref(self._rng_cur), float(nu), float(sigma), cn_a,
                                        self._stream), "pmc_step_pre")
            if self.prefetcher is not None and self._direct_now:
                # helper threads read x' once as soon as the completion word of this pre-step shows up
                lib.pmc_prefetcher_submit(self.prefetcher, self.h_done.data_ptr(), int(self._rng_cur.step) + 1,
                                          self.h_x.data_ptr(), n * D * 8, self.wait_timeout)
            self._post_uploads = True
            return
        self._post_uploads = False
        self._np_clean[0] = -1        # (the fine-grained launches count no rows: a word left by a composite pre-step is stale)
        self._rng_cur = self._rng(replay)
        st = _lib.stream_handle()
        kind = PMC_KIND_TPCN if self.tpcn else PMC_KIND_RWM
        cn_a = float((1.0 - sigma ** 2.0) ** 0.5) if self.tpcn else 0.0            # mcmc.py:85
        timed = self.events is not None
        with torch.cuda.device(self.device):
            e0 = self._ev() if timed else None
            _lib.check(lib.pmc_propose(
                kind, _lib.ptr(self.theta32) if self.pre else None, None if self.pre else _lib.ptr(self.u),
                _lib.ptr(self.mu_d), _lib.ptr(self.inv_cov_d), _lib.ptr(self.chol_d), float(nu), float(sigma), cn_a,
                C.byref(self._rng_cur), _lib.ptr(self.p_theta64), _lib.ptr(self.p_theta32) if self.pre else None,
                _lib.ptr(self.quad) if self.tpcn else None, _lib.ptr(self.p_quad) if self.tpcn else None,
                n, D, st), "pmc_propose")
            e1 = self._ev() if timed else None
            e2 = e1
            if self.pre:
                _lib.check(lib.pmc_maf_inverse(C.byref(self.flow._desc), _lib.ptr(self.p_theta32), _lib.ptr(self.p_u32),
                                               _lib.ptr(self.p_ldjf), n, self.flow.inverse_algo, st), "pmc_maf_inverse")
                e2 = self._ev() if timed else None
                _lib.check(lib.pmc_scaler_inverse(C.byref(self.scaler_desc), _lib.ptr(self.p_u32), None,
                                                  _lib.ptr(self.p_u), _lib.ptr(self.p_x), _lib.ptr(self.p_xT),
                                                  _lib.ptr(self.p_logdetj),
                                                  _lib.ptr(self.p_fin), n, st), "pmc_scaler_inverse")
            else:
                _lib.check(lib.pmc_scaler_inverse(C.byref(self.scaler_desc), None, _lib.ptr(self.p_theta64),
                                                  _lib.ptr(self.p_u), _lib.ptr(self.p_x), _lib.ptr(self.p_xT),
                                                  _lib.ptr(self.p_logdetj),
                                                  _lib.ptr(self.p_fin), n, st), "pmc_scaler_inverse")
            e3 = self._ev() if timed else None
            if self.prior_desc is not None:
                _lib.check(lib.pmc_prior_logpdf(C.byref(self.prior_desc), _lib.ptr(self.p_x), _lib.ptr(self.p_fin),
                                                _lib.ptr(self.p_logp), n, st), "pmc_prior_logpdf")
        if self.prior_desc is not None:
            self.h_logp.copy_(self.p_logp, non_blocking=True)
        self.h_x.copy_(self.p_xT if self.x_order == "F" else self.p_x, non_blocking=True)
        self.h_fin.copy_(self.p_fin, non_blocking=True)
        if timed:
            self._cur_ev = [e0, e1, e2, e3, self._ev()]

    def _configure_step(self):
        """The fields of the composite entry points' struct that depend on the engine's switches only: written when
        one of them changed."""
        cfg = (self.flow.inverse_algo if self.pre else 0, self.rng_prefill, self.host_direct, self.spin_wait, bool(self.fill_rejected))
        if cfg != self._pre_cfg:
            self._pre_cfg = cfg
            if self.pre:
                self._step.inverse_algo = self.flow.inverse_algo
            self._step.rng_ready = C.cast(C.pointer(self._rng_ready), C.c_void_p) if self.rng_prefill else None
            direct = bool(self.host_direct and self.x_order == "F")
            self._step.host_direct = int(direct)
            self._step.p_xT = None if (direct or self.p_xT is None) else self.p_xT.data_ptr()
            self._direct_now = direct and self.spin_wait
            self._step.h_done = self.h_done.data_ptr() if self._direct_now else None
            self._step.done_ticket = self._done_ticket.data_ptr() if self._direct_now else None
            self._step.h_clean = self.h_clean.data_ptr() if self._direct_now else None
            self._step.clean_count = self._clean_count.data_ptr() if self._direct_now else None
            self._step.ev_pre_done = None if self._direct_now else self._ev_pre     # (the completion word replaces it)
            self._step.fill_rejected = int(bool(self.fill_rejected) and self._direct_now)

    def evaluate(self, log_prior, log_like, have_blobs=False, blobs=None, waited=False):
        """Host black boxes on the compacted rows, ``mcmc.py:100-121``.  Returns
        ``(n_calls, blobs_prime)``."""
        tm = self.host_timers
        t0 = time.perf_counter() if tm is not None else 0.0
        if waited:
            pass                                      # (pmc_pipeline_next returned behind this lane's completion word)
        elif self._post_uploads:
            # x', finite, logp' are complete at this event / completion word; the next step's variates are
            # generated behind it
            if self._direct_now:
                _lib.check(self.lib.pmc_wait_flag(self.h_done.data_ptr(), self.step_idx + 1, self.wait_timeout), "pmc_wait_flag")
            else:
                _lib.check(self.lib.pmc_event_synchronize(self._ev_pre), "pmc_event_synchronize")
        else:
            torch.cuda.current_stream().synchronize()
        if tm is not None:
            t1 = time.perf_counter(); tm["wait_device"] += t1 - t0
            _lp, _ll = log_prior, log_like

            def log_prior(a, _f=_lp):
                ta = time.perf_counter(); r = _f(a); tm["prior"] += time.perf_counter() - ta
                return r

            def log_like(a, _f=_ll):
                ta = time.perf_counter(); r = _f(a); tm["likelihood"] += time.perf_counter() - ta
                return r
        if self.prior_desc is not None:
            log_prior = None                          # logp' came back from the device with x'
            if ((waited or self._post_uploads) and self._direct_now and self._np_clean[0] == 0
                    and self.host_threads <= 1 and not have_blobs):
                # the fused pre-step counted no row with a non-finite x' or logp' (pmc_step_t.h_clean): both masks of
                # mcmc.py:100-109 are all-true, x'[mask] is x' itself
                self._np_logl[:] = log_like(self._np_x)[0]
                return self.n, None
            bad = int(self._np_clean[0])
            if ((waited or self._post_uploads) and self._direct_now and self._step.fill_rejected
                    and 0 < bad <= self.fill_rejected_max * self.n and self.host_threads <= 1 and not have_blobs):
                # a few rows do not reach the likelihood; their host rows hold the walkers' current x (fill_rejected): the
                # whole block goes to the likelihood, those rows' values are dropped -- the calls counted are the rows of
                # mcmc.py:117's x'[mask]
                good = self._np_fin.astype(bool) & np.isfinite(self._np_logp)
                ll = log_like(self._np_x)[0]
                np.copyto(self._np_logl, ll)
                self._np_logl[~good] = -np.inf
                return int(good.sum()), None
        n = self.n
        x_prime = self._np_x
        logp_prime = self._np_logp
        logl_prime = self._np_logl
        fin_i = self._np_fin
        blobs_prime = None
        if fin_i.all() and self.host_threads > 1 and not have_blobs:
            # rows are independent for a vectorised likelihood (the reference itself calls it on
            # arbitrary compacted subsets, mcmc.py:106,117): evaluate row chunks concurrently
            if self._pool is None:
                self._pool = _host_pool(self.host_threads - 1, self.host_cores)
            k = self.host_threads
            bounds = [(i * n // k, (i + 1) * n // k) for i in range(k)]

            def work(b):
                lo, hi = b
                xs = x_prime[lo:hi]
                if log_prior is not None:
                    logp_prime[lo:hi] = log_prior(xs)
                lp = logp_prime[lo:hi]
                ok = np.isfinite(lp)
                if ok.all():
                    logl_prime[lo:hi] = log_like(xs)[0]
                    return hi - lo
                ll = np.full(hi - lo, -np.inf)
                ll[ok] = log_like(xs[ok])[0]
                logl_prime[lo:hi] = ll
                return int(ok.sum())
            # the calling thread takes the last chunk itself, the pool's threads the others
            futs = [self._pool.submit(work, b) for b in bounds[:-1]]
            calls = work(bounds[-1]) + sum(f.result() for f in futs)
            self._upload_logs()
            return calls, None
        if fin_i.all():
            # every proposal is finite (the usual case): x'[mask] of mcmc.py:106 is x' itself
            if log_prior is not None:
                logp_prime[:] = log_prior(x_prime)
            finite = np.isfinite(logp_prime)
            if finite.all():
                if have_blobs:
                    blobs_prime = np.empty(n, dtype=np.dtype((blobs[0].dtype, blobs[0].shape)))
                    logl_prime[:], blobs_prime[:] = log_like(x_prime)
                else:
                    logl_prime[:], _ = log_like(x_prime)
                self._upload_logs()
                return n, blobs_prime
        else:
            finite = fin_i.astype(bool)
            if log_prior is not None:
                logp_prime[finite] = log_prior(x_prime[finite])
                logp_prime[~finite] = -np.inf
            finite = finite & np.isfinite(logp_prime)
        if have_blobs:
            blobs_prime = np.empty(n, dtype=np.dtype((blobs[0].dtype, blobs[0].shape)))
            logl_prime[finite], blobs_prime[finite] = log_like(x_prime[finite])
        else:
            logl_prime[finite], _ = log_like(x_prime[finite])
        logl_prime[~finite] = -np.inf
        self._upload_logs()
        return int(np.sum(finite)), blobs_prime

    def _upload_logs(self):
        if not self._post_uploads:                    # the composite post step does the H2D itself
            self.p_logl.copy_(self.h_logl, non_blocking=True)
            if self.prior_desc is None:
                self.p_logp.copy_(self.h_logp, non_blocking=True)

    def accept_enqueue(self, beta, nu=0.0, want_mask=False, host_sums=True, adapt=None, n_total=None, others=()):
        """Composite path: enqueue the Metropolis accept + this engine's sums behind the host's logl' (no wait).
        ``adapt`` = Adaptation.coefficients(): the kernel's last block also updates the device-side sigma / mu.
        ``others``: engines over the other row ranges of the walker set whose accepts are already enqueued on this
        stream -- the kernel adds their sums to its own before the update and the host copy."""
        assert self._post_uploads
        self._want_mask = bool(want_mask)
        self._host_sums = bool(host_sums)
        st = self._step
        if adapt is not None and self.device_adapt:
            st.adapt_mode, st.adapt_c_sigma, st.adapt_c_mu, st.adapt_cap = adapt
            st.adapt_n_total = float(self.n if n_total is None else n_total)
        else:
            st.adapt_mode = 0
        if len(others) != st.adapt_n_other or any(st.adapt_other[k] != o.sums.data_ptr() for k, o in enumerate(others)):
            for k, o in enumerate(others):
                st.adapt_other[k] = o.sums.data_ptr()
            st.adapt_n_other = len(others)
        _lib.check(self.lib.pmc_step_post(C.byref(self._step), C.byref(self._rng_cur), float(beta), float(nu),
                                          int(want_mask), int(host_sums), self._stream), "pmc_step_post")

    def accept_wait(self):
        """Wait for what accept_enqueue started; returns this engine's host sums (valid with host_sums=True)."""
        if self._direct_now and self._host_sums and not self._want_mask:   # (the mask copy is a stream operation)
            _lib.check(self.lib.pmc_wait_flag(self.h_done.data_ptr() + 8, self.step_idx + 1, self.wait_timeout), "pmc_wait_flag")
        else:
            _lib.check(self.lib.pmc_stream_synchronize(self._stream), "pmc_stream_synchronize")
        self.step_idx += 1
        return self._np_sums

    def accept_reduce(self, beta, nu=0.0, want_mask=False):
        """Metropolis accept + global sums; returns the (all-reduced) host copy."""
        if self._post_uploads:
            import torch.distributed as dist
            sharded = dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1
            self.accept_enqueue(beta, nu, want_mask, host_sums=not sharded)
            if sharded:
                allreduce_sums(self.sums, self.group)
                self.h_sums.copy_(self.sums, non_blocking=True)
                torch.cuda.current_stream().synchronize()
                self.step_idx += 1
                return self._np_sums
            return self.accept_wait()
        kind = PMC_KIND_TPCN if self.tpcn else PMC_KIND_RWM
        timed = self.events is not None
        with torch.cuda.device(self.device):
            e5 = self._ev() if timed else None
            _lib.check(self.lib.pmc_accept(kind, int(self.pre), C.byref(self._state), C.byref(self._prop), float(beta),
                                           float(nu), C.byref(self._rng_cur), _lib.ptr(self.alpha), _lib.ptr(self.accept),
                                           _lib.ptr(self.sums), _lib.ptr(self.ws), self.n, self.D,
                                           _lib.stream_handle()), "pmc_accept")
            if timed:
                self.events.append(self._cur_ev + [e5, self._ev()])
        allreduce_sums(self.sums, self.group)
        self.h_sums.copy_(self.sums, non_blocking=True)
        if want_mask:
            self.h_accept.copy_(self.accept, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        self.step_idx += 1
        return self._np_sums

    def download(self, device=False):
        """The walker state as numpy arrays, or (``device=True``) as the engine's own device tensors."""
        g = (lambda t: t) if device else (lambda t: t.cpu().numpy())
        return dict(u=g(self.u), x=g(self.x), logdetj=g(self.logdetj), logl=g(self.logl), logp=g(self.logp))


# --------------------------------------------------------------------------
class LanedEngine:
    """One walker set as K contiguous row ranges (lanes), each a StepEngine on its own HIP stream.

    The rows of a step are independent between the proposal and the accept (``mcmc.py:77-141``: per-walker
    loops and a row-wise likelihood); only the adaptation (``:152-156``) needs all of them.  So while the host
    evaluates the likelihood of lane k, the device still works on the proposals of lane k+1 and on the accept
    of lane k-1: the step costs  device(lane 0) + likelihood(all rows) + accept(last lane)  instead of
    device(all) + likelihood(all) + accept(all).  The Philox counters are keyed on the global walker index
    (``shard_offset``), so a laned step draws exactly the variates of the un-laned one; the only difference is
    the order in which the D+4 sums are added (last bits of mean(alpha), mean(theta))."""

    def __init__(self, kind, n, n_dim, flow, scaler, lanes=2, group=None, shard_offset=0, seed=0, x_order="C",
                 streams=True, first_fraction=None):
        """``streams=False``: all lanes on the current stream, one after the other (the pipelined mode:
        :meth:`start_pipeline` / :meth:`step_pipelined`)."""
        n = int(n)
        lanes = max(1, min(int(lanes), (n + 15) // 16))
        per = ((n + lanes - 1) // lanes + 15) // 16 * 16          # whole 16-row sets per lane
        self.bounds = [(min(k * per, n), min((k + 1) * per, n)) for k in range(lanes)]
        if first_fraction is not None and lanes == 2:
            # two lanes of unequal size: the step costs device(lane 0) + max(likelihood(all), device(lane 1) +
            # likelihood(lane 1)); a somewhat larger first lane balances the two terms
            cut = min(n, max(16, int(round(n * float(first_fraction) / 16.0)) * 16))
            self.bounds = [(0, cut), (cut, n)]
        self.bounds = [b for b in self.bounds if b[1] > b[0]]
        self.n, self.D, self.group = n, int(n_dim), group
        self.device = _lib.require_gpu()
        self.lanes = []
        for k, (lo, hi) in enumerate(self.bounds):
            if streams:
                st = torch.cuda.Stream(device=self.device, priority=-1 if k == 0 else 0)   # lane 0 is waited for first
                with torch.cuda.stream(st):
                    e = StepEngine(kind, hi - lo, n_dim, flow, scaler, group=group,
                                   shard_offset=int(shard_offset) + lo, seed=seed, x_order=x_order)
                e.stream = st
            else:
                e = StepEngine(kind, hi - lo, n_dim, flow, scaler, group=group, shard_offset=int(shard_offset) + lo,
                               seed=seed, x_order=x_order)
            self.lanes.append(e)
        self.lib = self.lanes[0].lib
        self.tpcn, self.pre = self.lanes[0].tpcn, self.lanes[0].pre
        self._tot = torch.zeros(self.D + 4, dtype=torch.float64, device=self.device)
        self._h_tot = torch.zeros(self.D + 4, dtype=torch.float64).pin_memory()
        self._ev = [torch.cuda.Event() for _ in self.lanes]
        self.host_timers = None
        self._h_flag = torch.zeros(1, dtype=torch.int64).pin_memory()
        self._flag_value = 0
        self._parts = (C.c_void_p * len(self.lanes))(*[e.sums.data_ptr() for e in self.lanes])
        self._tot_part = (C.c_void_p * 1)(self._tot.data_ptr())
        # the lane pipeline behind the C ABI (pmc_pipeline_*): default whenever the ranks do not have to all-reduce between
        # the last accept and the adaptation (that exchange is torch.distributed's); PMC_C_PIPELINE=0: the same launches
        # enqueued from Python (round 2; kept as the cross-check of the test suite)
        self.c_pipeline = os.environ.get("PMC_C_PIPELINE", "1") != "0"
        self._pipe = None

    def __del__(self):
        self._drop_pipe()

    def _drop_pipe(self):
        if getattr(self, "_pipe", None):
            try:
                self.lib.pmc_pipeline_destroy(self._pipe)
            except Exception:
                pass
            self._pipe = None

    def _each(self, fn):
        out = []
        for e in self.lanes:
            if e.stream is None:
                out.append(fn(e))
            else:
                with torch.cuda.stream(e.stream):
                    out.append(fn(e))
        return out

    def configure(self, **kw):
        for e in self.lanes:
            for k, v in kw.items():
                setattr(e, k, v)

    def load_state(self, u, x, logdetj, logl, logp):
        for e, (lo, hi) in zip(self.lanes, self.bounds):
            self_stream = torch.cuda.current_stream(self.device) if e.stream is None else e.stream
            with torch.cuda.stream(self_stream):
                e.load_state(u[lo:hi], x[lo:hi], logdetj[lo:hi], logl[lo:hi], logp[lo:hi])
        torch.cuda.synchronize(self.device)

    def can_pipeline(self):
        return all(e.stream is None and e.can_pipeline() for e in self.lanes)

    # ------------------------------------------------------------------ pipelined mode (one stream)
    def start_pipeline(self, sigma, mu, nu):
        """Adaptation state to the device, pre-steps of the first step into the queue."""
        first = self.lanes[0]
        first.adapt_upload(sigma, mu)
        for e in self.lanes[1:]:
            e._step.adapt_state = first.adapt_state.data_ptr()       # one state for all lanes
            e.device_adapt = True
        import torch.distributed as dist
        sharded = dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1
        self._drop_pipe()
        comm = small_comm(self.lib, self.group, self.D + 4) if (sharded and self.c_pipeline) else None
        if self.c_pipeline and (not sharded or comm) and len(self.lanes) <= 8:
            K = len(self.lanes)
            for e in self.lanes:
                e._configure_step()
                e._post_uploads = True
            assert all(e._direct_now for e in self.lanes)
            self._lane_structs = (C.c_void_p * K)(*[C.addressof(e._step) for e in self.lanes])
            offs = (C.c_uint64 * K)(*[e.offset for e in self.lanes])
            stream = first.stream.cuda_stream if first.stream is not None else _lib.stream_handle()
            first._stream = stream
            self._pipe = self.lib.pmc_pipeline_create(self._lane_structs, K, first.seed, offs, first.prefetcher,
                                                      float(first.wait_timeout), stream)
            if not self._pipe:
                _lib.check(1, "pmc_pipeline_create")
            if comm:
                # sharded: the ranks' sums meet inside the library (pmc_comm_adapt_update) between the last accept and the
                # adaptation -- the same C pipeline as a single rank's
                _lib.check(self.lib.pmc_pipeline_set_comm(self._pipe, comm), "pmc_pipeline_set_comm")
            _lib.check(self.lib.pmc_pipeline_start(self._pipe, float(nu), int(first.step_idx)), "pmc_pipeline_start")
            return
        for e in self.lanes:
            e.host_timers = self.host_timers
            e.propose(None, nu)

    def resume_pipeline(self, nu):
        """Pre-steps of the next step into the queue of a pipeline whose last step enqueued none (``more=False``): what
        :meth:`start_pipeline` does at the head of a call, without rebuilding the pipeline object -- the adaptation state
        on the device is the one the last step left."""
        first = self.lanes[0]
        if self._pipe:
            _lib.check(self.lib.pmc_pipeline_start(self._pipe, float(nu), int(first.step_idx)), "pmc_pipeline_start")
            return
        for e in self.lanes:
            e.host_timers = self.host_timers
            e.propose(None, nu, step=e.step_idx)

    def pipeline_stats(self, reset=True):
        """Host seconds the C pipeline spent {waiting for x', waiting for the sums, enqueuing accepts, enqueuing
        pre-steps} and the steps they cover, since the last reset (None without a C pipeline)."""
        if not self._pipe:
            return None
        out = (C.c_double * 6)()
        _lib.check(self.lib.pmc_pipeline_stats(self._pipe, out, int(reset)), "pmc_pipeline_stats")
        return dict(wait_x=out[0], wait_sums=out[1], enqueue_accept=out[2], enqueue_next_pre=out[3], steps=int(out[4]))

    def step_pipelined(self, beta, nu, coefficients, n_total, log_prior, log_like, more=True):
        """One step: per lane  wait x' -> likelihood -> enqueue accept;  then one launch adds the lane sums (and,
        between two of them, the ranks all-reduce), adapts sigma / mu on the device and hands the sums to the host;
        the pre-steps of the next step are enqueued behind it before the host waits for those sums."""
        if self._pipe:
            # pmc_pipeline_next: accept of the lane just evaluated, then the wait for the next lane's x' -- or, behind the
            # last lane, the closing accept, the next pre-steps and the wait for the sums
            nxt, P = self.lib.pmc_pipeline_next, self._pipe
            mode, c_sigma, c_mu, cap = coefficients
            tm = self.host_timers
            if nxt(P, -1, beta, nu, mode, c_sigma, c_mu, cap, n_total, 0):
                _lib.check(1, "pmc_pipeline_next")
            calls = 0
            for k, e in enumerate(self.lanes):
                e.host_timers = tm
                calls += e.evaluate(log_prior, log_like, waited=True)[0]
                if nxt(P, k, beta, nu, mode, c_sigma, c_mu, cap, n_total, int(more)):
                    _lib.check(1, "pmc_pipeline_next")
            for e in self.lanes:
                e.step_idx += 1
            sums = self.lanes[-1]._np_sums
            if sums[0] != sums[0]:                       # (NaN: a rank did not arrive within the timeout, pmc_comm_adapt_update)
                raise _lib.PocomcAmdError("sharded step: a rank did not deliver its sums within wait_timeout")
            return calls, sums
        import torch.distributed as dist
        sharded = dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1
        lib, D, K = self.lib, self.D, len(self.lanes)
        tm = self.host_timers
        clock = time.perf_counter
        calls = 0
        last = self.lanes[-1]
        for e in self.lanes:
            e.host_timers = tm
            calls += e.evaluate(log_prior, log_like)[0]
            t0 = clock() if tm is not None else 0.0
            if e is last and not sharded:
                # the last range's accept closes the set: total sums, sigma / mu update, sums + completion word to the host
                e.accept_enqueue(beta, nu, host_sums=True, adapt=coefficients, n_total=n_total, others=self.lanes[:-1])
            elif e is last:
                # sharded: the last range's accept adds the ranges of this rank into the vector the ranks all-reduce
                # (its "host copy" of the sums points at that device vector)
                if e._step.h_sums != self._tot.data_ptr():
                    e._step.h_sums = self._tot.data_ptr()
                e.accept_enqueue(beta, nu, host_sums=True, others=self.lanes[:-1])
            else:
                e.accept_enqueue(beta, nu, host_sums=False)
            if tm is not None:
                tm["enqueue_accept"] = tm.get("enqueue_accept", 0.0) + clock() - t0
        t0 = clock() if tm is not None else 0.0
        stream = self.lanes[0]._stream
        if sharded:
            # (this rank's total is in self._tot) -> all-reduce over the ranks -> sigma / mu update, sums + completion word
            # to the host
            mode, c_sigma, c_mu, cap = coefficients
            self._flag_value += 1
            done = _lib.pmc_done_t(flag=self._h_flag.data_ptr(), value=self._flag_value, ticket=None)
            state = self.lanes[0].adapt_state.data_ptr()
            allreduce_sums(self._tot, self.group)
            _lib.check(lib.pmc_adapt_update(self._tot_part, 1, D, None, self._h_tot.data_ptr(), state, mode, c_sigma,
                                            c_mu, cap, float(n_total), C.byref(done), stream), "pmc_adapt_update")
        t1 = clock() if tm is not None else 0.0
        if more:
            for e in self.lanes:
                e.propose(None, nu, step=e.step_idx + 1)
        t2 = clock() if tm is not None else 0.0
        if sharded:
            _lib.check(lib.pmc_wait_flag(self._h_flag.data_ptr(), self._flag_value, self.lanes[0].wait_timeout), "pmc_wait_flag")
            sums = self._h_tot.numpy()
        else:
            sums = last.accept_wait()                       # (increments the last lane's step counter)
        if tm is not None:
            t3 = clock()
            tm["enqueue_adapt"] = tm.get("enqueue_adapt", 0.0) + t1 - t0
            tm["enqueue_next_pre"] = tm.get("enqueue_next_pre", 0.0) + t2 - t1
            tm["wait_sums"] = tm.get("wait_sums", 0.0) + t3 - t2
        for e in (self.lanes if sharded else self.lanes[:-1]):
            e.step_idx += 1
        return calls, sums

    def finish_pipeline(self):
        _lib.check(self.lib.pmc_stream_synchronize(self.lanes[0]._stream), "pmc_stream_synchronize")

    def set_geometry(self, mu=None, cov=None):
        self._each(lambda e: e.set_geometry(mu=mu, cov=cov))
        torch.cuda.synchronize(self.device)

    def set_device_prior(self, prior):
        return all(self._each(lambda e: e.set_device_prior(prior)))

    def set_mu(self, mu):
        for e in self.lanes:
            e.set_mu(mu)                                   # composite path: a write to the lane's pinned h_mu

    def step(self, sigma, nu, beta, log_prior, log_like):
        """One MCMC step of all lanes; returns (likelihood calls, global sums on the host)."""
        import torch.distributed as dist
        sharded = dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1
        # (the composite entry points take the lane's stream explicitly: no stream context on this path)
        for e in self.lanes:                               # all proposals are in flight before the first wait
            e.host_timers = self.host_timers
            e.propose(sigma, nu)
        calls = 0
        for e in self.lanes:
            calls += e.evaluate(log_prior, log_like)[0]
            e.accept_enqueue(beta, nu, host_sums=not sharded)
        if sharded:
            # lane sums -> one vector -> all-reduce over the ranks -> host
            main = torch.cuda.current_stream(self.device)
            for e, ev in zip(self.lanes, self._ev):
                ev.record(e.stream)
                main.wait_event(ev)
            torch.stack([e.sums for e in self.lanes]).sum(0, out=self._tot)
            allreduce_sums(self._tot, self.group)
            self._h_tot.copy_(self._tot, non_blocking=True)
            main.synchronize()
            for e in self.lanes:
                e.step_idx += 1
            return calls, self._h_tot.numpy()
        tot = None
        for e in self.lanes:
            sk = e.accept_wait()
            tot = sk.copy() if tot is None else tot + sk
        return calls, tot

    def download(self, device=False):
        parts = [e.download(device) for e in self.lanes]
        cat = torch.cat if device else np.concatenate
        return {k: cat([p[k] for p in parts]) for k in parts[0]}


def _proposal_draws(kind, theta32, geometry, sigma, rows, seed=20240929):
    """``rows`` proposals theta' (float32, on the walkers' device) for a strided sample of the walkers ``theta32``, drawn
    from the step's proposal law: tpCN ``mu + sqrt(1 - sigma^2) (theta - mu) + sigma sqrt(s) L z`` with the Student-t scale
    ``s = 1 / Gamma((D + nu) / 2, 2 / (nu + delta))`` (``mcmc.py:77-85``), RWM ``theta + sigma L z`` (``:251-253``).  Host
    float64 arithmetic from a private numpy generator (no global stream is touched): the input of the 16-bit sweep's guard,
    which must see the heavy-tailed points the sweep will be given, not the walkers' current positions."""
    n, D = theta32.shape
    take = int(min(rows, n))
    idx = torch.linspace(0, n - 1, take, device=theta32.device).long()
    th = theta32[idx].double().cpu().numpy()
    g = np.random.default_rng(seed)
    z = g.standard_normal((take, D))
    if kind == "preconditioned_pcn":
        mu, cov, nu = np.asarray(geometry.t_mean, float), np.asarray(geometry.t_cov, float), float(geometry.t_nu)
        L = np.linalg.cholesky(cov)
        d = th - mu
        delta = np.einsum("ij,ij->i", d @ np.linalg.inv(cov), d)
        s = 1.0 / (g.standard_gamma((D + nu) / 2.0, size=take) * 2.0 / (nu + delta))
        sig = min(float(sigma), 0.99)
        thp = mu + np.sqrt(1.0 - sig ** 2) * d + sig * np.sqrt(s)[:, None] * (z @ L.T)
    else:
        L = np.linalg.cholesky(np.asarray(geometry.normal_cov, float))
        thp = th + float(sigma) * (z @ L.T)
    return torch.from_numpy(thp.astype(np.float32)).to(theta32.device)


def _global_count(n, group):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        return n * dist.get_world_size(group)        # equal shards by construction
    return n


_PREFETCHERS = {}


def host_prefetch(lib, n_threads, cores=None):
    """Process-wide cache warmer (``csrc/host_prefetch.hip``): ``n_threads`` helper threads pinned to ``cores`` (cores
    that share the L3 with the thread that calls the likelihood) read every x' the kernels hand to the host once, as
    soon as it is complete.  Option ``host_prefetch`` / ``host_prefetch_cores`` of the kernels; the handle lives as long
    as the process."""
    key = (int(n_threads), tuple(cores) if cores else None)
    if key not in _PREFETCHERS:
        arr = (C.c_int32 * n_threads)(*[int(c) for c in cores][:n_threads]) if cores and len(cores) >= n_threads else None
        h = lib.pmc_prefetcher_create(int(n_threads), arr)
        if not h:
            raise RuntimeError("pmc_prefetcher_create failed")
        _PREFETCHERS[key] = h
    return _PREFETCHERS[key]


def _run(kind, state_dict, function_dict, option_dict, replay=None, trace=None):
    pre = kind.startswith("preconditioned")
    tpcn = kind in ("preconditioned_pcn", "pcn")
    # numpy arrays (the reference's contract: inputs are copied, mcmc.py:31-35) or device tensors (the Sampler's
    # pool: the engine copies them into its own buffers on the device; with option_dict["device_state"] the
    # results stay there too)
    on_device = isinstance(state_dict.get("u"), torch.Tensor)
    cp = (lambda a: a) if on_device else np.copy
    u = cp(state_dict.get("u"))
    x = cp(state_dict.get("x"))
    logdetj = cp(state_dict.get("logdetj"))
    logl = cp(state_dict.get("logl"))
    logp = cp(state_dict.get("logp"))
    beta = state_dict.get("beta")
    blobs = state_dict.get("blobs")
    have_blobs = blobs is not None

    log_like = function_dict.get("loglike")
    log_prior = function_dict.get("logprior")
    # rows actually handed to the likelihood: with ``fill_rejected`` (default for the pipelined step) a row that does not
    # reach the reference's likelihood call (mcmc.py:117: x'[mask]) is passed with the walker's CURRENT x and its value
    # dropped -- the likelihood must be row-wise and free of side effects (it is for the reference, which calls it on
    # arbitrary compacted subsets); results["calls"] counts the reference's rows, results["evaluations"] these
    _rows_passed = []
    _user_like = log_like

    def log_like(a, _f=_user_like, _seen=_rows_passed.append):
        _seen(len(a))
        return _f(a)
    scaler = function_dict.get("scaler")
    flow = function_dict.get("flow") if pre else None
    geometry = function_dict.get("theta_geometry" if pre else "u_geometry")

    n_max = option_dict.get("n_max")
    n_steps = option_dict.get("n_steps")
    progress_bar = option_dict.get("progress_bar")
    group = option_dict.get("group")
    seed = option_dict.get("seed")
    if seed is None:
        # Philox key; taken from numpy's global stream so that np.random.seed() makes a run
        # reproducible like the reference (not touched when the variates are replayed)
        seed = 0 if replay is not None else (int(np.random.randint(0, 2 ** 31 - 1)) * 2 ** 31
                                             + int(np.random.randint(0, 2 ** 31 - 1)))
    n_walkers, n_dim = x.shape

    import torch.distributed as dist
    sharded = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    x_order = option_dict.get("x_order", "C")
    # lanes > 1: row ranges whose device work overlaps the host likelihood of the others (LanedEngine); opt-in,
    # it pays when the likelihood is expensive next to the device's share of a step
    lanes = int(option_dict.get("lanes") or 1)
    if have_blobs or trace is not None or replay is not None:
        lanes = 1                                   # (blobs / traces / replayed variates: whole-set bookkeeping)
    # pipelined: sigma / mu adapted on the device, the pre-step of step k+1 enqueued behind the accept of step k
    # (needs the kernels to read / write the pinned host buffers themselves: x_order 'F')
    want_pipe = (option_dict.get("pipeline", True) and x_order == "F" and not have_blobs and trace is None
                 and replay is None and all(option_dict.get(k, True) for k in ("host_direct", "spin_wait")))
    if lanes > 1 or want_pipe:                      # (the pipelined step lives behind pmc_pipeline_*: one lane is a pipeline too)
        eng = LanedEngine(kind, n_walkers, n_dim, flow, scaler, lanes=lanes, group=group,
                          first_fraction=option_dict.get("first_lane"),
                          shard_offset=option_dict.get("shard_offset", 0), seed=seed, x_order=x_order,
                          streams=not want_pipe)
        tune = eng.configure
    else:
        eng = StepEngine(kind, n_walkers, n_dim, flow, scaler, group=group,
                         shard_offset=option_dict.get("shard_offset", 0), seed=seed, x_order=x_order)
        tune = lambda **kw: [setattr(eng, k, v) for k, v in kw.items()]
    laned = isinstance(eng, LanedEngine)
    if option_dict.get("wait_timeout") is not None:
        tune(wait_timeout=_wait_timeout(option_dict["wait_timeout"]))
    for key in ("host_direct", "rng_prefill", "spin_wait", "fill_rejected"):
        if key in option_dict:
            tune(**{key: bool(option_dict[key])})
    if option_dict.get("host_threads", 1) > 1:
        # opt-in: the black boxes are called concurrently on row chunks from this many threads (they must be
        # thread-safe; the reference calls them on the calling thread only)
        tune(host_threads=int(option_dict["host_threads"]), host_cores=option_dict.get("host_cores"))
    if int(option_dict.get("host_prefetch") or 0) > 0:
        tune(prefetcher=host_prefetch(eng.lib, int(option_dict["host_prefetch"]), option_dict.get("host_prefetch_cores")))
    owner = getattr(log_prior, "__self__", None)
    if owner is not None and option_dict.get("device_prior", True) and hasattr(owner, "device_descriptor"):
        eng.set_device_prior(owner)                 # Prior.logpdf of uniform / normal factors on the device
    eng.load_state(u, x, logdetj, logl, logp)
    nu = 0.0
    if tpcn:
        nu = float(geometry.t_nu)
        eng.set_geometry(mu=geometry.t_mean, cov=geometry.t_cov)
    else:
        eng.set_geometry(cov=geometry.normal_cov)

    n_total = _global_count(n_walkers, group)
    # stop metric before the first step (mcmc.py:70 / :243), global over the shards
    if on_device:
        from .tools import device_sum
        s_l, s_p, s_j = device_sum(logl.contiguous()), device_sum(logp.contiguous()), device_sum(logdetj.contiguous())
        first = [0.0, s_l + s_p, s_l + s_p + s_j]
    else:
        first = [0.0, float(np.sum(logl + logp)), float(np.sum(logl + logp + logdetj))]
    init = torch.tensor(first + [0.0] * (n_dim + 1), dtype=torch.float64, device=eng.device)
    allreduce_sums(init, group)
    init = init.cpu().numpy()
    ad = Adaptation(kind, n_dim, n_total, n_steps, n_max, option_dict.get("proposal_scale"),
                    geometry.t_mean if tpcn else None, (init[1] if tpcn else init[2]) / n_total)

    if pre and getattr(flow, "_lane16", None) is not None and flow.inverse_guard_enabled:
        # the 16-bit sweep's safety net where the sweep is used: on the points THIS call hands to flow.inverse (mcmc.py:88)
        # -- proposals drawn from the step's own law at the starting sigma / mu, a strided sample over ALL lanes' walkers;
        # float32 from here on if the sweep is not an inverse there.  Sharded: every rank enters the reduction whatever its
        # own state (a rank that already fell back votes for the fallback), one rank's fallback is everybody's.
        armed = flow.inverse_precision_active != "f32"
        guard = None
        if armed:
            th_all = torch.cat([e_.theta32 for e_ in (eng.lanes if laned else [eng])])
            thp = _proposal_draws(kind, th_all, geometry, float(ad.sigma), 4096)
            guard = flow.check_inverse_precision(theta=thp, rows=4096)
        fall = (not armed) or (guard is not None and not guard["passed"])
        if sharded:
            flag = torch.tensor([1.0 if fall else 0.0], device=eng.device)
            dist.all_reduce(flag, group=group)
            fall = float(flag.item()) > 0
        if fall and flow._desc.lane16:
            flow._desc.lane16 = None
            if flow.inverse_guard is not None:
                flow.inverse_guard["passed"] = False

    n_calls = 0
    pipelined = want_pipe and eng.can_pipeline() and (laned or not sharded)
    if pipelined and laned:
        eng.start_pipeline(float(ad.sigma), ad.mu, nu)
    elif pipelined:
        # adaptation on the device: the pre-step of step k+1 is enqueued right behind the accept of step k and
        # runs while the host still waits for / digests the sums of step k (which it needs for the stop rule only)
        eng.adapt_upload(float(ad.sigma), ad.mu)
        eng.propose(ad.sigma, nu)
    while True:
        rp = None
        if replay is not None:
            replay.begin_step()
            rp = dict(gamma=replay.std_gamma((n_dim + nu) / 2, n_walkers) if tpcn else None,
                      z=replay.normal(n_walkers, n_dim), u=replay.uniform(n_walkers))
        if pipelined and laned:
            calls, sums = eng.step_pipelined(beta, nu, ad.coefficients(), n_total, log_prior, log_like,
                                             more=ad.i + 1 < n_max)
        elif pipelined:
            calls, _ = eng.evaluate(log_prior, log_like)
            eng.accept_enqueue(beta, nu, adapt=ad.coefficients(), n_total=n_total)
            if ad.i + 1 < n_max:                         # (the last permitted step has no successor)
                eng.propose(None, nu, step=eng.step_idx + 1)
            sums = eng.accept_wait()
        elif laned:
            calls, sums = eng.step(ad.sigma, nu, beta, log_prior, log_like)
        else:
            eng.propose(ad.sigma, nu, rp)
            calls, blobs_prime = eng.evaluate(log_prior, log_like, have_blobs, blobs)
            sums = eng.accept_reduce(beta, nu, want_mask=have_blobs or trace is not None)
        n_calls += calls
        if have_blobs:
            mask = eng.h_accept.numpy().astype(bool)
            blobs[mask] = blobs_prime[mask]
        stop = ad.update(sums)
        if kind == "preconditioned_pcn" and not pipelined:
            eng.set_mu(ad.mu)
        if trace is not None:
            trace.append(dict(alpha=eng.alpha.cpu().numpy(), accept=eng.h_accept.numpy().astype(bool).copy(),
                              theta_prime=eng.p_theta64.cpu().numpy(), u_prime=eng.p_u.cpu().numpy(),
                              x_prime=eng.p_x.cpu().numpy(), logdetj_prime=eng.p_logdetj.cpu().numpy(),
                              logdetj_flow_prime=eng.p_ldjf.cpu().numpy() if pre else None,
                              finite=eng.p_fin.cpu().numpy().astype(bool), sigma=float(ad.sigma),
                              mu=None if ad.mu is None else ad.mu.copy(), **eng.download()))
        if progress_bar is not None:
            progress_bar.update_stats(dict(calls=progress_bar.info["calls"] + calls, acc=ad.mean_alpha, steps=ad.i,
                                           logP=sums[1] / n_total, eff=ad.sigma / (2.38 / np.sqrt(n_dim))))
        if stop:
            break
    if pipelined:
        # a pre-step launched ahead of a plateau stop is still in flight; it touches proposal buffers only
        if laned:
            eng.finish_pipeline()
        else:
            _lib.check(eng.lib.pmc_stream_synchronize(eng._stream), "pmc_stream_synchronize")

    keep = bool(option_dict.get("device_state", False))
    if keep:
        torch.cuda.synchronize(eng.device)             # (nothing of this call is in flight any more)
    out = eng.download(device=keep)                    # (the numpy download synchronises by itself)
    for e_ in (eng.lanes if laned else [eng]):
        e_._recycle = True
    return dict(u=out["u"], x=out["x"], logdetj=out["logdetj"], logl=out["logl"], logp=out["logp"], blobs=blobs,
                efficiency=ad.sigma, accept=ad.mean_alpha, steps=ad.i, calls=n_calls, proposal_scale=ad.sigma,
                evaluations=int(sum(_rows_passed)))


def preconditioned_pcn(state_dict, function_dict, option_dict, replay=None, trace=None):
    """``pocomc/mcmc.py:8-183``."""
    return _run("preconditioned_pcn", state_dict, function_dict, option_dict, replay, trace)


def preconditioned_rwm(state_dict, function_dict, option_dict, replay=None, trace=None):
    """``pocomc/mcmc.py:186-341``."""
    return _run("preconditioned_rwm", state_dict, function_dict, option_dict, replay, trace)


def pcn(state_dict, function_dict, option_dict, replay=None, trace=None):
    """``pocomc/mcmc.py:344-506``."""
    return _run("pcn", state_dict, function_dict, option_dict, replay, trace)


def rwm(state_dict, function_dict, option_dict, replay=None, trace=None):
    """``pocomc/mcmc.py:508-654``."""
    return _run("rwm", state_dict, function_dict, option_dict, replay, trace)
