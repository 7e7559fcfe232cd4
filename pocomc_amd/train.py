"""Flow training -- host-side mirror of ``Flow.fit`` (``pocomc/flow.py:165-384``).

The loop structure, the split quirk (``validation_split`` is the TRAIN fraction,
``flow.py:248-249``), the weighted loss (``:311-312``), global-norm clipping (``:318``),
AdamW (``:268``), the per-dataset loss normalisation (``:323``, ``:348``),
ReduceLROnPlateau (``:275-283``, ``:352-355``), the best-state snapshot (``:364-367``) and the
early stop at ``int(1.5 * patience)`` stale epochs (``:369-374``) follow the reference.
The arithmetic -- forward, backward, clip, optimizer -- runs in the gfx950 kernels
(``pmc_maf_loss_grad``, ``pmc_adamw_step``, ``pmc_maf_forward``, ``pmc_neg_weighted_sum``).
"""
from __future__ import annotations

import ctypes as C
import time

import numpy as np
import torch

from . import _lib


MAX_SETS = 256              # row sets of 16 per loss/gradient launch (one workgroup each; larger batches come in chunks)
SCRATCH_BUDGET = 1 << 29    # bytes of activation / delta scratch a flow may hold (wide flows keep fewer sets, >= 32)


class TrainState:
    """Training-side device buffers of one Flow (built on first ``fit``)."""

    def __init__(self, flow, light=False):
        """``light``: only what every engine needs (gradient, scalars); the bf16 engine keeps its own images
        (:class:`WideState`) and never touches the float32 training image."""
        spec, dev = flow.spec, flow.device
        n = spec.n_params
        # masked entries stay 0 for ever; one extra element carries the batch loss through the
        # gradient all-reduce of sharded training
        self.grad_ext = torch.zeros(n + 1, dtype=torch.float32, device=dev)
        self.grad = self.grad_ext[:n]
        self.wsum = torch.zeros(1, dtype=torch.float32, device=dev)
        self.scal = torch.zeros(4, dtype=torch.float32, device=dev)      # [loss, spare...]
        self.light = bool(light)
        self.n_sets = 0
        if light:
            self.sq_partial = torch.zeros(256, dtype=torch.float32, device=dev)       # PMC_ADAMW_SCRATCH
            self.desc = None
            return
        L = spec.train_layout()
        pT_idx, gmap = spec.train_index()
        jobs = spec.train_jobs()
        self.packT_idx = torch.from_numpy(pT_idx).to(dev)
        self.gmap = torch.from_numpy(gmap).to(dev)
        self.jobs = torch.from_numpy(jobs.reshape(-1).copy()).to(dev)
        self.n_waves = int(flow.lib.pmc_maf_train_waves(C.byref(flow._desc)))   # waves per chain workgroup
        self.tables = torch.from_numpy(spec.train_tables(self.n_waves)).to(dev)
        self.n_jobs = int(jobs.shape[0])
        self.packedT = torch.zeros(pT_idx.size, dtype=torch.float32, device=dev)
        self.sq_partial = torch.zeros(max(self.n_jobs, 256), dtype=torch.float32, device=dev)   # >= PMC_ADAMW_SCRATCH
        self.par_pt = spec.par_per_transform()
        self.desc = _lib.pmc_maf_train_t(packedT=self.packedT.data_ptr(), gmap=self.gmap.data_ptr(),
                                         pkT_per_transform=L["pkT_per_transform"],
                                         gmap_per_transform=L["gmap_per_transform"],
                                         jobs=self.jobs.data_ptr(), n_jobs=self.n_jobs,
                                         tables=self.tables.data_ptr(), table_waves=self.n_waves,
                                         n_sq_partial=self.sq_partial.numel(), par_per_transform=self.par_pt,
                                         sq_partial=self.sq_partial.data_ptr())
        self.xt_floats = (spec.n_transforms + 1) * spec.Dp * 16
        self.act_floats = spec.n_transforms * 3 * spec.Hp * 16
        self.par_floats = spec.n_transforms * self.par_pt
        per_set = 4 * (self.xt_floats + 2 * self.act_floats + self.par_floats)
        self.set_cap = int(max(32, min(MAX_SETS, SCRATCH_BUDGET // per_set)))
        self.scatter_maps(flow)             # (host-built once per Flow, like the maps above: not a per-fit cost)
        # the stream the validation passes of a fit run on (fit_flow): created and used once here -- a new HIP stream's first
        # launch sets up its hardware queue, tens of milliseconds that do not belong to an epoch
        self.side_stream = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(self.side_stream):
            self.scal.zero_()
        self.side_stream.synchronize()

    def ensure_sets(self, n_rows):
        """Scratch for the row sets of one launch: transform inputs, activations, deltas, output gradients."""
        need = max(1, min(self.set_cap, (int(n_rows) + 15) // 16))
        if need <= self.n_sets:
            return
        dev = self.grad.device
        self.xt_scratch = torch.empty(need * self.xt_floats, dtype=torch.float32, device=dev)
        self.act_scratch = torch.empty(need * self.act_floats, dtype=torch.float32, device=dev)
        self.delta_scratch = torch.empty(need * self.act_floats, dtype=torch.float32, device=dev)
        self.par_scratch = torch.empty(need * self.par_floats, dtype=torch.float32, device=dev)
        self.loss_partial = torch.zeros(need, dtype=torch.float32, device=dev)
        self.n_sets = need
        self.desc.xt_scratch = self.xt_scratch.data_ptr()
        self.desc.act_scratch = self.act_scratch.data_ptr()
        self.desc.delta_scratch = self.delta_scratch.data_ptr()
        self.desc.par_scratch = self.par_scratch.data_ptr()
        self.desc.loss_partial = self.loss_partial.data_ptr()
        self.desc.max_sets = need

    def scatter_maps(self, flow):
        """CSR inverse of the two pack maps (``pmc_adamw_t.scatter_*``): where every parameter sits in the forward /
        inverse image (``Flow._packed``) and in the transposed training image (``packedT``)."""
        if getattr(self, "_scatter", None) is None:
            a = flow._pack_idx.cpu().numpy().astype(np.int64)
            b = self.packT_idx.cpu().numpy().astype(np.int64)
            src = np.concatenate([a, b])                       # parameter index of every image position, -1 = padding
            pos = np.nonzero(src >= 0)[0]
            order = np.argsort(src[pos], kind="stable")
            dst = pos[order].astype(np.int32)
            counts = np.bincount(src[pos], minlength=flow.params.numel())
            ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
            dev = flow.params.device
            self._scatter = (torch.from_numpy(ptr).to(dev), torch.from_numpy(dst).to(dev))
        return self._scatter

    def repack(self, flow):
        if self.light:
            return
        with torch.cuda.device(flow.device):
            _lib.check(flow.lib.pmc_maf_pack(_lib.ptr(flow.params), _lib.ptr(self.packT_idx), _lib.ptr(self.packedT),
                                             self.packedT.numel(), _lib.stream_handle()), "pmc_maf_pack(T)")


def _train_state(flow):
    want_light = _wide_state(flow) is not None
    ts = getattr(flow, "_train", None)
    if ts is None or (ts.light and not want_light):
        flow._train = TrainState(flow, light=want_light)
    return flow._train


WIDE_MIN_HIDDEN = 256       # precision="bf16" flows at least this wide train on the bf16 matrix cores


class WideState:
    """Device side of ``csrc/maf_train_bf16.hip`` (``pmc_maf_wide_t``): the row-major bf16 weight image with its index
    map, the float32 bias image and the activation scratch."""

    def __init__(self, flow):
        spec, dev, lib = flow.spec, flow.device, flow.lib
        L = spec.wide_layout()
        img_idx, bias_idx = spec.wide_index()
        self.image_idx = torch.from_numpy(img_idx).to(dev)
        self.bias_idx = torch.from_numpy(bias_idx).to(dev)
        self.image = torch.zeros(img_idx.size, dtype=torch.int16, device=dev)
        self.bias = torch.zeros(bias_idx.size, dtype=torch.float32, device=dev)
        nbytes = int(lib.pmc_maf_wide_scratch_bytes(C.byref(flow._desc)))
        self.scratch = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        self.sq = torch.zeros(256, dtype=torch.float32, device=dev)      # PMC_ADAMW_SCRATCH
        self.desc = _lib.pmc_maf_wide_t(image=self.image.data_ptr(), image_idx=self.image_idx.data_ptr(),
                                        image_per_transform=L["per_transform"], bias=self.bias.data_ptr(),
                                        bias_idx=self.bias_idx.data_ptr(), bias_per_transform=L["bias_per_transform"],
                                        scratch=self.scratch.data_ptr(), scratch_bytes=nbytes, wsum=None)

    def refresh(self, flow):
        with torch.cuda.device(flow.device):
            _lib.check(flow.lib.pmc_maf_wide_refresh(C.byref(flow._desc), C.byref(self.desc), _lib.ptr(flow.params),
                                                     _lib.stream_handle()), "pmc_maf_wide_refresh")


def _wide_state(flow):
    """The bf16 training engine of this flow, or None (float32 kernels): ``precision="bf16"`` affine flows of hidden
    width >= WIDE_MIN_HIDDEN; ``flow.train_engine = "f32" | "bf16"`` overrides the width rule."""
    engine = getattr(flow, "train_engine", None)
    if engine == "f32" or getattr(flow, "precision", "f32") != "bf16" or flow.spec.univariate != "affine":
        if engine == "bf16":
            raise ValueError("train_engine='bf16' needs an affine flow with precision='bf16'")
        return None
    if engine != "bf16" and flow.spec.hidden < WIDE_MIN_HIDDEN:
        return None
    if getattr(flow, "_wide", None) is None:
        flow._wide = WideState(flow)
    return flow._wide


def loss_and_grad(flow, xb, wb=None, idx=None, refresh=True):
    """Loss of one batch (device scalar tensor) and its gradient (in ``flow._train.grad``).
    ``idx`` (int64, device) selects the batch rows out of ``xb`` / ``wb``.  (``refresh=False``: the caller keeps the
    bf16 training image in step with the parameters itself.)"""
    ts = _train_state(flow)
    n = xb.shape[0] if idx is None else idx.numel()
    ts.scal.zero_()
    ws = _wide_state(flow)
    if ws is not None:
        if refresh:
            ws.refresh(flow)
        with torch.cuda.device(flow.device):
            _lib.check(flow.lib.pmc_maf_loss_grad_bf16(C.byref(flow._desc), C.byref(ws.desc), _lib.ptr(xb),
                                                       _lib.ptr(wb) if wb is not None else None,
                                                       _lib.ptr(idx) if idx is not None else None,
                                                       1000.0, _lib.ptr(ts.grad), _lib.ptr(ts.scal), n,
                                                       _lib.stream_handle()), "pmc_maf_loss_grad_bf16")
        return ts.scal[0]
    ts.ensure_sets(n)
    with torch.cuda.device(flow.device):
        _lib.check(flow.lib.pmc_maf_loss_grad(C.byref(flow._desc), C.byref(ts.desc), _lib.ptr(xb),
                                              _lib.ptr(wb) if wb is not None else None,
                                              _lib.ptr(idx) if idx is not None else None,
                                              1000.0, _lib.ptr(ts.grad), _lib.ptr(ts.scal), n, _lib.stream_handle()),
                   "pmc_maf_loss_grad")
    return ts.scal[0]


def batch_loss(flow, xb, wb=None, group=None, sharded=False):
    """Loss of one batch without gradient (validation, ``flow.py:327-346``).  ``sharded``: ``xb`` is
    this rank's part of the batch, the weight normalisation uses the all-reduced weight sum."""
    ts = _train_state(flow)
    lib = flow.lib
    st = _lib.stream_handle()
    n = xb.shape[0]
    z = torch.empty_like(xb)
    lp = torch.empty(n, dtype=torch.float32, device=flow.device)
    ts.scal.zero_()
    with torch.cuda.device(flow.device):
        _lib.check(lib.pmc_maf_forward(C.byref(flow._desc), _lib.ptr(xb), _lib.ptr(z), None, _lib.ptr(lp), n, st),
                   "pmc_maf_forward")
        if wb is not None:
            _lib.check(lib.pmc_sum_f32(_lib.ptr(wb), C.c_void_p(ts.scal.data_ptr() + 4), n, st), "pmc_sum_f32")
            if sharded:
                import torch.distributed as dist
                dist.all_reduce(ts.scal[1:2], group=group)
        _lib.check(lib.pmc_neg_weighted_sum(_lib.ptr(lp), _lib.ptr(wb) if wb is not None else None,
                                            C.c_void_p(ts.scal.data_ptr() + 4) if wb is not None else None,
                                            1000.0, _lib.ptr(ts.scal), n, st), "pmc_neg_weighted_sum")
    return ts.scal[0]


class AdamW:
    """``torch.optim.AdamW`` on the flat parameter vector (``flow.py:268``)."""

    def __init__(self, flow, lr, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8):
        self.flow, self.lr, self.wd, self.betas, self.eps = flow, float(lr), float(weight_decay), betas, eps
        self.m = torch.zeros_like(flow.params)
        self.v = torch.zeros_like(flow.params)
        self.t = 0
        ws = _wide_state(flow)
        if ws is not None:
            ws.refresh(flow)

    def step(self, max_norm):
        """One clipped step on the gradient in ``flow._train.grad``, then refresh both kernel images."""
        f = self.flow
        ts = _train_state(f)
        self.t += 1
        with torch.cuda.device(f.device):
            _lib.check(f.lib.pmc_adamw_step(_lib.ptr(f.params), _lib.ptr(ts.grad), _lib.ptr(self.m), _lib.ptr(self.v),
                                            f.params.numel(), self.lr, self.betas[0], self.betas[1], self.eps, self.wd,
                                            float(max_norm) if max_norm is not None else 0.0, self.t,
                                            _lib.ptr(ts.sq_partial), _lib.stream_handle()),
                       "pmc_adamw_step")
        f.repack()
        ws = _wide_state(f)
        if ws is not None:
            ws.refresh(f)
        else:
            ts.repack(f)

    def epoch(self, x, w, perm, batch_size, max_norm, loss_acc, gate=None, stream=None, snapshot=None):
        """``flow.py:297-323`` for one epoch in a single library call: every batch's loss/gradient,
        clip, AdamW step and image refresh is enqueued back to back; ``loss_acc`` (f32 [1], device)
        accumulates the batch losses.  ``gate`` (a recorded ``torch.cuda.Event``): the epoch's first optimizer step waits
        for it (``pmc_maf_train_epoch_gated``: the previous epoch's validation pass on another stream)."""
        f = self.flow
        ts = _train_state(f)
        ws = _wide_state(f)
        if ws is not None:
            c = _lib.pmc_adamw_t(params=f.params.data_ptr(), grad=ts.grad.data_ptr(), exp_avg=self.m.data_ptr(),
                                 exp_avg_sq=self.v.data_ptr(), n_params=f.params.numel(), lr=self.lr,
                                 beta1=self.betas[0], beta2=self.betas[1], eps=self.eps, weight_decay=self.wd,
                                 max_norm=float(max_norm) if max_norm is not None else 0.0, step=self.t)
            with torch.cuda.device(f.device):
                _lib.check(f.lib.pmc_maf_train_epoch_bf16(C.byref(f._desc), C.byref(ws.desc), C.byref(c), _lib.ptr(x),
                                                          _lib.ptr(w) if w is not None else None,
                                                          _lib.ptr(perm) if perm is not None else None,
                                                          x.shape[0], int(batch_size), _lib.ptr(loss_acc),
                                                          _lib.ptr(ws.sq), _lib.stream_handle()),
                           "pmc_maf_train_epoch_bf16")
            self.t = int(c.step)
            f.repack()                      # the float32 / fragment images follow once per epoch (validation, inference)
            return
        ts.ensure_sets(batch_size)
        # (the descriptor is built once per optimizer: an epoch of the Sampler's fits is ~100 us and this method runs once
        #  per epoch on the driver thread -- only what changes between epochs is written)
        c = getattr(self, "_desc", None)
        if c is None:
            sc_ptr, sc_dst = ts.scatter_maps(f)
            c = self._desc = _lib.pmc_adamw_t(
                params=f.params.data_ptr(), grad=ts.grad.data_ptr(), exp_avg=self.m.data_ptr(),
                exp_avg_sq=self.v.data_ptr(), n_params=f.params.numel(),
                pack_idx=f._pack_idx.data_ptr(), packed=f._packed.data_ptr(), n_packed=f._packed.numel(),
                packT_idx=ts.packT_idx.data_ptr(), packedT=ts.packedT.data_ptr(), n_packedT=ts.packedT.numel(),
                beta1=self.betas[0], beta2=self.betas[1], eps=self.eps, weight_decay=self.wd,
                scatter_ptr=sc_ptr.data_ptr(), scatter_dst=sc_dst.data_ptr())
            self._desc_refs = (C.byref(f._desc), C.byref(ts.desc), C.byref(c))
        c.lr, c.step, c.max_norm = self.lr, self.t, float(max_norm) if max_norm is not None else 0.0
        c.snapshot = snapshot.data_ptr() if snapshot is not None else None     # (the parameters behind the epoch's last step)
        d_maf, d_tr, d_opt = self._desc_refs

        def call():
            return f.lib.pmc_maf_train_epoch_gated(d_maf, d_tr, d_opt, x.data_ptr(), w.data_ptr() if w is not None else None,
                                                   perm.data_ptr() if perm is not None else None, x.shape[0], int(batch_size),
                                                   loss_acc.data_ptr(), gate.cuda_event if gate is not None else None,
                                                   stream if stream is not None else torch.cuda.current_stream(f.device).cuda_stream)
        if torch.cuda.current_device() == f.device.index:           # (the usual case: no device switch to pay for)
            rc = call()
        else:
            with torch.cuda.device(f.device):
                rc = call()
        if rc:
            _lib.check(rc, "pmc_maf_train_epoch")
        self.t = int(c.step)


class ReduceLROnPlateau:
    """``torch.optim.lr_scheduler.ReduceLROnPlateau(mode='min', factor=0.2, threshold=1e-4,
    threshold_mode='abs', min_lr=1e-6)`` as configured at ``flow.py:275-283``."""

    def __init__(self, opt, patience, factor=0.2, threshold=1e-4, min_lr=1e-6):
        self.opt, self.patience, self.factor, self.threshold, self.min_lr = opt, patience, factor, threshold, min_lr
        self.best, self.bad = float("inf"), 0

    def step(self, metric):
        if metric < self.best - self.threshold:
            self.best, self.bad = metric, 0
        else:
            self.bad += 1
        if self.bad > self.patience:
            new = max(self.opt.lr * self.factor, self.min_lr)
            if self.opt.lr - new > 1e-8:
                self.opt.lr = new
            self.bad = 0


def _dist_world(group):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group)
    return 1


def _weight_flags(flow):
    """uint8 [n_params]: 1 for the entries of the hyper-networks' weight matrices (what ``parameter_name.endswith('weight')``
    selects at ``flow.py:409-411`` -- masked-out entries included), 0 for biases."""
    ts = _train_state(flow)
    if getattr(ts, "weight_flags", None) is None:
        spec = flow.spec
        f = np.zeros(spec.n_params, dtype=np.uint8)
        for t in range(spec.n_transforms):
            for name in ("W0", "W1", "W2", "W3"):
                off, sz = spec.offsets[name]
                f[t * spec.params_per_transform + off: t * spec.params_per_transform + off + sz] = 1
        ts.weight_flags = torch.from_numpy(f).to(flow.device)
    return ts.weight_flags


def sharded_epoch(flow, opt, x, w, perm, batch_size, max_norm, loss_acc, group, penalty=None):
    """One epoch of data-parallel training: every rank holds a shard of the
    training rows; a global batch of ``batch_size`` rows is ``batch_size / world`` local rows per rank.
    Per batch: [all-reduce of the weight sum] -> local loss/gradient -> ONE all-reduce of
    (gradient, loss) -> the same clip + AdamW step on every rank (the clip needs the norm of the
    reduced gradient, flow.py:318)."""
    import torch.distributed as dist
    ts = _train_state(flow)
    world = dist.get_world_size(group)
    lb = max(1, int(batch_size) // world)
    n = x.shape[0]
    wide = _wide_state(flow)
    if wide is None:
        ts.ensure_sets(lb)
    st = _lib.stream_handle()
    for b0 in range(0, n, lb):
        nb = min(lb, n - b0)
        idx = perm[b0:b0 + nb] if perm is not None else None
        xb = x if idx is not None else x[b0:b0 + nb]
        wb = None if w is None else (w if idx is not None else w[b0:b0 + nb])
        with torch.cuda.device(flow.device):
            if w is not None:
                ts.wsum.zero_()
                wsel = w[idx] if idx is not None else wb
                _lib.check(flow.lib.pmc_sum_f32(_lib.ptr(wsel), _lib.ptr(ts.wsum), nb, st), "pmc_sum_f32")
                dist.all_reduce(ts.wsum, group=group)
                (wide or ts).desc.wsum = ts.wsum.data_ptr()
            ts.grad_ext[-1:].zero_()
            if wide is not None:
                _lib.check(flow.lib.pmc_maf_loss_grad_bf16(C.byref(flow._desc), C.byref(wide.desc), _lib.ptr(xb),
                                                           _lib.ptr(wb) if wb is not None else None,
                                                           _lib.ptr(idx) if idx is not None else None, 1000.0,
                                                           _lib.ptr(ts.grad),
                                                           C.c_void_p(ts.grad_ext.data_ptr() + 4 * ts.grad.numel()),
                                                           nb, st), "pmc_maf_loss_grad_bf16")
            else:
                _lib.check(flow.lib.pmc_maf_loss_grad(C.byref(flow._desc), C.byref(ts.desc), _lib.ptr(xb),
                                                      _lib.ptr(wb) if wb is not None else None,
                                                      _lib.ptr(idx) if idx is not None else None, 1000.0,
                                                      _lib.ptr(ts.grad), C.c_void_p(ts.grad_ext.data_ptr() + 4 * ts.grad.numel()),
                                                      nb, st), "pmc_maf_loss_grad")
            (wide or ts).desc.wsum = None
        dist.all_reduce(ts.grad_ext, group=group)
        loss_acc += ts.grad_ext[-1:]
        if penalty is not None:                   # the same penalty on every rank, once per global batch
            penalty(ts.grad, loss_acc)
        opt.step(max_norm)


def _batches(n, batch_size, shuffle):
    """``DataLoader(TensorDataset(...), batch_size, shuffle)``: a fresh permutation per epoch,
    last partial batch kept."""
    idx = torch.randperm(n) if shuffle else torch.arange(n)
    return [idx[i:i + batch_size] for i in range(0, n, batch_size)]


def fit_flow(flow, x, weights=None, validation_split=0.0, epochs=1000, batch_size=1000, patience=20,
             learning_rate=1e-3, weight_decay=0, laplace_scale=None, gaussian_scale=None, annealing=True,
             noise=None, shuffle=True, clip_grad_norm=1.0, verbose=0, group=None, sharded=None):
    """``sharded`` (default: a ``torch.distributed`` group with more than one rank exists): ``x`` /
    ``weights`` are THIS rank's shard of the training rows (equal shard sizes), ``batch_size`` is the
    global batch; gradients and losses are all-reduced (RCCL on the GPUs) so that every rank takes
    the same optimizer steps and the same early-stopping decisions."""
    from .flow import torch_double_to_float
    x = torch_double_to_float(torch.as_tensor(x))
    dev = flow.device
    n_samples, n_dim = x.shape
    if n_dim != flow.n_dim:
        raise ValueError("x has the wrong number of columns")
    w = None if weights is None else torch.as_tensor(weights).to(torch.float32)

    x = x.to(dev)
    w = None if w is None else w.to(dev)
    if shuffle:                                                     # flow.py:234-238 (the permutation still comes
        rand_indx = torch.randperm(n_samples).to(dev)               # from torch's CPU generator; rows move on the device)
        x = x[rand_indx]
        if w is not None:
            w = w[rand_indx]
    x = x.contiguous()
    w = None if w is None else w.contiguous()

    if validation_split > 0.0:                                      # flow.py:247-259
        cut = int(validation_split * n_samples)
        x_train, x_valid = x[:cut], x[cut:]
        w_train, w_valid = (None, None) if w is None else (w[:cut], w[cut:])
        validation = True
    else:
        x_train, w_train, x_valid, w_valid = x, w, None, None
        validation = False

    world = _dist_world(group)
    if sharded is None:
        sharded = world > 1
    if sharded:
        import torch.distributed as dist
    # ---- options (flow.py:240-245, :304-307, :314-315): see pmc_weight_penalty / pmc_add_noise_f32 in the header
    penalty = None
    if laplace_scale is not None or gaussian_scale is not None:
        penalty = (float(laplace_scale or 0.0), float(gaussian_scale or 0.0), _weight_flags(flow))
    noise_scale, noise_seed = None, 0
    if noise is not None:
        # flow.py:241-245: `mean_min_dist = torch.mean(min_dist)` -- the mean of the LAST row's distances to all rows
        # (the nearest-neighbour distances `min_dists` computed in the loop above it are never used); its loop raises
        # for a row without a positive distance, like torch.min of an empty tensor
        if n_samples < 2:
            raise RuntimeError("min(): Expected reduction dim to be specified for input.numel() == 0.")
        md = torch.zeros(1, dtype=torch.float32, device=dev)
        if sharded:
            # one scale for all ranks, the single-process value: the mean distance of the LAST row of the whole set (the
            # last rank's last row) to every row of every shard
            rank = dist.get_rank(group)
            last = x[-1:].clone()
            dist.broadcast(last, src=dist.get_global_rank(group, world - 1) if group is not None else world - 1, group=group)
            xc = torch.cat([x, last]).contiguous()
            with torch.cuda.device(dev):
                _lib.check(flow.lib.pmc_mean_distance_f32(_lib.ptr(xc), n_samples + 1, n_dim, n_samples, _lib.ptr(md),
                                                          _lib.stream_handle()), "pmc_mean_distance_f32")
            tot = md.double() * (n_samples + 1)            # (the appended row adds a zero distance)
            dist.all_reduce(tot, group=group)
            md = (tot / (n_samples * world)).float()
        else:
            with torch.cuda.device(dev):
                _lib.check(flow.lib.pmc_mean_distance_f32(_lib.ptr(x), n_samples, n_dim, n_samples - 1, _lib.ptr(md),
                                                          _lib.stream_handle()), "pmc_mean_distance_f32")
        noise_scale = float(noise) * float(md.item())
        noise_seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) * 2 ** 31 + int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        if sharded:
            sd = torch.tensor([noise_seed], dtype=torch.int64, device=dev)
            dist.broadcast(sd, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            noise_seed = int(sd.item())
    opt = AdamW(flow, learning_rate, weight_decay)
    sched = ReduceLROnPlateau(opt, patience) if annealing else None
    _train_state(flow).repack(flow)

    history = dict(loss=[], val_loss=[])
    monitor = "val_loss" if validation else "loss"
    best_epoch, best_loss = 0, np.inf
    best_model = flow.params.clone()
    start = time.time()

    n_train = x_train.shape[0]
    n_valid = x_valid.shape[0] if validation else 0
    # One epoch may be IN FLIGHT while the host looks at the previous epoch's losses (no scheduler, one GPU): the
    # early-stop test needs every epoch's loss on the host, and waiting for it before enqueuing the next epoch
    # leaves the GPU idle for the whole host turn-around (most of an epoch when the training set is one or two
    # batches, the Sampler's usual case).  The speculative epoch changes nothing observable: on an early stop the
    # best parameters are restored (flow.py:369-374), and no epoch is enqueued past `epochs`.
    pipelined = (sched is None) and not sharded
    slots = 2 if pipelined else 1
    acc_d = [torch.zeros(2, dtype=torch.float32, device=dev) for _ in range(slots)]   # [train loss, val loss]
    from .mcmc import _pinned_take, _pinned_give
    acc_h = [_pinned_take((2,), torch.float32).zero_() for _ in range(slots)]
    done = [torch.cuda.Event() for _ in range(slots)]
    after = [torch.empty_like(flow.params) for _ in range(slots)]                    # parameters after the epoch
    # (pinned staging from the process-wide free list: page-locking a buffer costs about a millisecond)
    ts = _train_state(flow)
    # Plain fits (one process, float32 engine, no penalty term, shuffled): ONE host-to-device copy per epoch carries both
    # permutations AND the zeros of the two loss accumulators (two int64 words in front of the permutations in one staging
    # buffer), and the parameters behind the epoch's last step are written by that step itself (pmc_adamw_t.snapshot) --
    # three ~5 us launches fewer per epoch, of the ~75-120 us an epoch of the Sampler's fits takes.
    fused = bool(shuffle) and not sharded and penalty is None and _wide_state(flow) is None
    if fused:
        n_stage = 2 + n_train + n_valid
        h_stage = [_pinned_take((n_stage,), torch.int64) for _ in range(slots)]
        for h_ in h_stage:
            h_[:2] = 0
        d_stage = [torch.zeros(n_stage, dtype=torch.int64, device=dev) for _ in range(slots)]
        acc_d = [d[:1].view(torch.float32) for d in d_stage]           # [train loss, val loss] in the first word
        view = lambda b, which: b[2:2 + n_train] if which == 0 else b[2 + n_train:n_stage]
        h_perm, d_perm = [[], []], [[], []]
    else:
        h_perm = [[_pinned_take((max(n_train, 1),), torch.int64) for _ in range(slots)],
                  [_pinned_take((max(n_valid, 1),), torch.int64) for _ in range(slots)]]
        d_perm = [[torch.empty(max(n_train, 1), dtype=torch.int64, device=dev) for _ in range(slots)],
                  [torch.empty(max(n_valid, 1), dtype=torch.int64, device=dev) for _ in range(slots)]]
    if validation and not sharded and (getattr(ts, "logp_scratch", None) is None or ts.logp_scratch.numel() < n_valid):
        ts.logp_scratch = torch.empty(int(n_valid), dtype=torch.float32, device=dev)
    # The validation pass of epoch e only READS the parameters, and so do the loss / gradient launches of epoch e + 1's
    # first batch; a batch's chain kernel occupies 32 of 256 compute units.  So the pass runs on a second stream, next to
    # that batch, and only the first optimizer step of epoch e + 1 waits for it (pmc_maf_train_epoch_gated).  Float32
    # engine, one process, no penalty term (those paths enqueue batch by batch from Python).
    # It pays where the pass (two cross-stream hand-overs of ~16 us, the forward launch, the reduction, the copies: ~65 us
    # at the Sampler's sizes) is SHORTER than the first batch's chain + weight-gradient launches it hides behind, and costs
    # the host ~15 us per epoch: deep or spline flows (chain >= ~80 us) with a validation set of at most two batches' worth
    # of rows -- the Sampler's regime (README example, nsf6: 152 -> 123 us per epoch).  Measured otherwise: maf3 at D = 10
    # 76 -> 91 us (the epoch is bound by its ~120 us of enqueue, not by the device), the bench's fit (maf3 at D = 32, ten
    # batches, 5000 validation rows whose 313 workgroups crowd the chain's 32) 0.95 -> 1.05 ms.
    side = None
    long_chain = flow.spec.n_transforms * (2 if flow.spec.univariate == "rqs" else 1) >= 6
    if (validation and not sharded and penalty is None and _wide_state(flow) is None and long_chain
            and n_valid <= 2 * int(batch_size)):
        side = ts.side_stream
    train_done = [torch.cuda.Event() for _ in range(slots)] if side is not None else None
    main_h = torch.cuda.current_stream(dev).cuda_stream               # (looked up once: the fit stays on this stream)
    gate = [None]                                                     # the event the next epoch's first update waits for

    def upload_both(sl):
        torch.randperm(n_train, out=view(h_stage[sl], 0))            # (the same draws in the same order as upload_perm)
        if n_valid:
            torch.randperm(n_valid, out=view(h_stage[sl], 1))
        d_stage[sl].copy_(h_stage[sl], non_blocking=True)
        return view(d_stage[sl], 0), (view(d_stage[sl], 1) if n_valid else None)

    def upload_perm(which, sl, n):
        # DataLoader(shuffle=...), flow.py:251-265: a fresh permutation per pass (pinned staging, no host sync)
        torch.randperm(n, out=h_perm[which][sl][:n])
        d_perm[which][sl][:n].copy_(h_perm[which][sl][:n], non_blocking=True)
        return d_perm[which][sl][:n]

    noisy = None
    if noise_scale is not None:
        noisy = [[torch.empty_like(x_train) for _ in range(slots)],
                 [torch.empty_like(x_valid) if validation else None for _ in range(slots)]]

    def with_noise(which, sl, epoch, src):
        """Fresh noise on every row of a pass (the reference draws it per batch of every epoch, flow.py:305 / :334)."""
        if noisy is None:
            return src
        dst = noisy[which][sl]
        with torch.cuda.device(dev):
            # (a rank's shard draws the noise of ITS rows of the whole set: keyed by the global row)
            row0 = (dist.get_rank(group) if sharded else 0) * src.shape[0]
            _lib.check(flow.lib.pmc_add_noise_rows_f32(_lib.ptr(src), src.shape[0], n_dim, noise_scale, noise_seed,
                                                       2 * epoch + which, row0, _lib.ptr(dst), _lib.stream_handle()),
                       "pmc_add_noise_rows_f32")
        return dst

    def add_penalty(grad, loss, mult=1.0):
        b, g, flags = penalty
        with torch.cuda.device(dev):
            _lib.check(flow.lib.pmc_weight_penalty(_lib.ptr(flow.params), _lib.ptr(flags), _lib.ptr(grad) if grad is not None else None,
                                                   flow.params.numel(), b, g, float(mult), _lib.ptr(loss),
                                                   _lib.ptr(ts.sq_partial), _lib.stream_handle()), "pmc_weight_penalty")

    def penalised_epoch(xs, ws, perm, acc):
        """Batch by batch (flow.py:301-321) with the penalty's gradient added before the clip (flow.py:314-318)."""
        n = xs.shape[0]
        for b0 in range(0, n, int(batch_size)):
            nb = min(int(batch_size), n - b0)
            idx = perm[b0:b0 + nb] if perm is not None else None
            xb = xs if idx is not None else xs[b0:b0 + nb]
            wb = None if ws is None else (ws if idx is not None else ws[b0:b0 + nb])
            acc += loss_and_grad(flow, xb, wb, idx, refresh=False)
            add_penalty(ts.grad, acc)
            opt.step(clip_grad_norm)

    def enqueue(epoch):
        sl = epoch % slots
        acc2 = acc_d[sl]
        vperm_f = None
        if fused:
            perm, vperm_f = upload_both(sl)                          # (also zeroes acc2)
        else:
            acc2.zero_()
            perm = upload_perm(0, sl, n_train) if shuffle else None
        acc = acc2[0:1]
        xs = with_noise(0, sl, epoch, x_train)
        if sharded:
            sharded_epoch(flow, opt, xs, w_train, perm, batch_size, clip_grad_norm, acc, group,
                          penalty=add_penalty if penalty is not None else None)
        elif penalty is not None:
            penalised_epoch(xs, w_train, perm, acc)
        elif side is not None:
            opt.epoch(xs, w_train, perm, batch_size, clip_grad_norm, acc, gate=gate[0], stream=main_h,
                      snapshot=after[sl] if fused else None)
            if not fused:
                after[sl].copy_(flow.params)                          # (the parameters after this epoch's last update)
            main = torch.cuda.current_stream(dev)
            train_done[sl].record(main)
            with torch.cuda.stream(side):
                side.wait_event(train_done[sl])
                x_valid_e = with_noise(1, sl, epoch, x_valid)
                vperm = vperm_f if fused else (upload_perm(1, sl, n_valid) if shuffle else None)
                with torch.cuda.device(dev):
                    _lib.check(flow.lib.pmc_maf_valid_epoch(C.byref(flow._desc), _lib.ptr(x_valid_e),
                                                            _lib.ptr(w_valid) if w_valid is not None else None,
                                                            _lib.ptr(vperm) if vperm is not None else None,
                                                            n_valid, int(batch_size), _lib.ptr(ts.logp_scratch),
                                                            _lib.ptr(acc2[1:2]), C.c_void_p(side.cuda_stream)), "pmc_maf_valid_epoch")
                acc_h[sl].copy_(acc2, non_blocking=True)
                done[sl].record(side)
            gate[0] = done[sl]
            return
        else:
            opt.epoch(xs, w_train, perm, batch_size, clip_grad_norm, acc, stream=main_h, snapshot=after[sl] if fused else None)
        vacc = acc2[1:2]
        x_valid_e = with_noise(1, sl, epoch, x_valid) if validation else None
        if validation and not sharded:
            # the whole validation pass in one library call (batches of the reference's DataLoader, flow.py:327-348)
            vperm = vperm_f if fused else (upload_perm(1, sl, n_valid) if shuffle else None)
            with torch.cuda.device(dev):
                _lib.check(flow.lib.pmc_maf_valid_epoch(C.byref(flow._desc), _lib.ptr(x_valid_e),
                                                        _lib.ptr(w_valid) if w_valid is not None else None,
                                                        _lib.ptr(vperm) if vperm is not None else None,
                                                        n_valid, int(batch_size), _lib.ptr(ts.logp_scratch),
                                                        _lib.ptr(vacc), _lib.stream_handle()), "pmc_maf_valid_epoch")
        elif validation:
            vb = max(1, batch_size // world)
            for idx in _batches(n_valid, vb, shuffle):
                idx = idx.to(dev)
                vacc += batch_loss(flow, x_valid_e[idx].contiguous(),
                                   None if w_valid is None else w_valid[idx].contiguous(), group, sharded)
        if sharded:
            # the validation loss is a sum over the ranks' shards (the training loss already is: it rode
            # along with the gradients)
            dist.all_reduce(acc2[1:2], group=group)
        if validation and penalty is not None:
            # flow.py:342-343: every validation batch's loss carries the penalty
            vb = int(batch_size) if not sharded else max(1, int(batch_size) // world)
            add_penalty(None, vacc, mult=-(-n_valid // vb))
        if not fused:
            after[sl].copy_(flow.params)
        acc_h[sl].copy_(acc2, non_blocking=True)
        done[sl].record()

    if epochs > 0:
        enqueue(0)
    for epoch in range(epochs):
        sl = epoch % slots
        if pipelined and epoch + 1 < epochs:
            enqueue(epoch + 1)                                        # speculative: runs while we read this epoch's losses
        done[sl].synchronize()                                        # the one wait of the epoch
        both = acc_h[sl].numpy()
        n_tr, n_va = (n_train * world, n_valid * world) if sharded else (n_train, n_valid)
        train_loss = float(both[0]) / max(n_tr, 1)                   # flow.py:323
        history["loss"].append(train_loss)
        if validation:
            val_loss = float(both[1]) / max(n_va, 1)                 # flow.py:348
            history["val_loss"].append(val_loss)
        if sched is not None:
            sched.step(val_loss if validation else train_loss)
        if verbose > 1:
            print("Epoch %3d/%3d, train loss: %5.2f" % (epoch + 1, epochs, train_loss)
                  + (", val loss: %5.2f" % val_loss if validation else ""))
        if history[monitor][-1] < best_loss:                          # flow.py:364-367
            best_loss, best_epoch = history[monitor][-1], epoch
            best_model.copy_(after[sl])
        if epoch - best_epoch >= int(1.5 * patience):                 # flow.py:369-374
            if side is not None:
                torch.cuda.current_stream(dev).wait_stream(side)      # (a speculative validation pass still reads the images)
            flow.params.copy_(best_model)
            flow.repack()
            if verbose > 0:
                print("Finished early after %3d epochs" % best_epoch)
                print("Best loss achieved %5.2f" % best_loss)
            break
        if not pipelined and epoch + 1 < epochs:
            enqueue(epoch + 1)
    if verbose > 0:
        total = time.time() - start
        print("\nTime total:     %5.2f sec" % total)
        print("Time per epoch: %5.2f sec" % (total / epochs))
    # a speculative epoch may still be copying its permutations / losses: wait before the staging goes back
    if side is not None:
        torch.cuda.current_stream(dev).wait_stream(side)
        side.synchronize()
    torch.cuda.current_stream().synchronize()
    for t in acc_h + h_perm[0] + h_perm[1] + (h_stage if fused else []):
        _pinned_give(t)
    if getattr(flow, "_bf16", None) is not None or getattr(flow, "_lane16", None) is not None:
        flow.repack()                              # the 16-bit images follow the trained float32 parameters
    if getattr(flow, "_lane16", None) is not None:
        # the 16-bit sweep's safety net: compared with the float32 sweep on the latent image of the training rows
        # (what mcmc.py:88 inverts), float32 from here on if it is not an inverse within the bounds
        guard = flow.check_inverse_precision(theta=flow.forward(x[:4096])[0], rows=4096)
        if guard is not None:
            if sharded:
                # every rank takes the same sweep: one rank's fallback is everybody's
                flag = torch.tensor([0.0 if guard["passed"] else 1.0], device=dev)
                dist.all_reduce(flag, group=group)
                if float(flag.item()) > 0 and guard["passed"]:
                    flow._desc.lane16 = None
                    guard["passed"] = False
            history["inverse_guard"] = guard
    return history
