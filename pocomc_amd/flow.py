"""``Flow`` -- host-side mirror of ``pocomc.flow.Flow`` (``pocomc/flow.py:12-384``)
whose arithmetic runs in the gfx950 kernels behind ``include/pocomc_amd.h``.

Same constructor and methods as the reference class: ``Flow(n_dim, flow='maf3')``,
``forward / inverse / log_prob / sample / fit``.  Tensors may live on the CPU or
on the GPU; results come back on the input's device with dtype float32, float64
input is cast with the reference's warning (``pocomc/tools.py:295-316``).

The predefined MAFs of the reference are supported (``maf3 | maf6 | maf12``,
``pocomc/flow.py:54-68``) and the neural spline flows (``nsf3 | nsf6 | nsf12``,
``flow.py:69-86``: the same masked hyper-network with an 8-bin monotonic
rational-quadratic spline per feature); a ``MAFSpec`` gives a custom
depth/width (the reference takes a ``zuko.flows.Flow`` object there,
``flow.py:87-88``).
"""
from __future__ import annotations

import ctypes as C
import warnings

import numpy as np
import torch

from . import _lib
from .maf_spec import MAFSpec, SPEC_BY_NAME, NSF_BY_NAME, spec_by_name


def torch_double_to_float(x: torch.Tensor, warn: bool = True) -> torch.Tensor:
    """``pocomc/tools.py:295-316``."""
    if x.dtype == torch.float64 and warn:
        warnings.warn("Float64 data is currently unsupported, casting to Float32. "
                      "Output will also have type Float32.")
        return x.float()
    elif x.dtype == torch.float32:
        return x
    raise ValueError(f"Unsupported datatype for input data: {x.dtype}")


LANE16_BOUND = 1e-2        # largest per-walker relative difference of x the 16-bit sweep may show against the float32 sweep
LANE16_LADJ_BOUND = 1e-1   # ... and of the log-determinant (it enters the Metropolis ratio of mcmc.py:124-134 as it is)
LANE16_LADJ_Q99_BOUND = 5e-2   # ... and of the 99th percentile of that difference over the compared points: the maximum alone says
                               # nothing about how many walkers' acceptance probabilities are perturbed at the percent level.
                               # Measured on proposal-distributed points of the trained flows (round 6): config 5 bf16 median
                               # 5.9e-3 / q99 2.4e-2 / max 3.6e-2 (passes); config 3 f16 2.0e-3 / 1.0e-2 / 2.1e-2 (passes);
                               # config 3 bf16 1.6e-2 / 7.7e-2 / 0.22 (refused: float32)


class Flow:
    """Masked autoregressive flow resident on one MI355X."""

    def __init__(self, n_dim, flow="nsf3", device=None, seed=None, precision="f32", train_engine=None,
                 inverse_precision=None, inverse_guard=True):                                                                 # default flow as pocomc/flow.py:46
        """``precision="bf16"`` (affine flows): ``forward`` / ``log_prob`` run on the bf16 matrix cores with fp32
        accumulation (``csrc/maf_forward_bf16.hip``; BASELINE config 5 names this precision), and ``fit`` takes the bf16
        gradient engine (``csrc/maf_train_bf16.hip``: bf16 weights / activations, fp32 master parameters, accumulation and
        optimizer) when the hidden layers are wide (>= ``train.WIDE_MIN_HIDDEN`` units: the config-5 flow);
        ``train_engine="f32" | "bf16"`` fixes the engine instead of the width rule (kept by ``save_state``).  The
        parameters, the chain of the inverse and the univariate maps stay float32.  ``inverse_precision`` ("f32" | "bf16" |
        "f16"; default "f32" whatever ``precision`` says -- an explicit opt-in): operand type of the LEFT-LOOKING products of the
        inverse sweep of the wide flows (``csrc/maf_inverse_tri6.hip``: everything left of the diagonal tile, multiplied by
        the helper wavefronts on ``v_mfma_f32_16x16x16_bf16 / _f16`` with float32 accumulation; the dependent chain stays
        float32) -- it halves the activations' LDS footprint, so that BASELINE config 5's 5000 walkers per GPU take one round
        of the sweep instead of two.  Narrow flows (fewer than 16 hidden tiles) keep the float32 sweeps whatever it says.
        The 16-bit sweep is GUARDED: whenever the parameters change (``set_params``, the end of ``fit``) the 16-bit and the
        float32 sweep are run on latent points of this flow (``check_inverse_precision``) and the flow goes back to the
        float32 sweep, with a warning, if a walker's x differs by more than ``LANE16_BOUND`` (relative) or its log-determinant
        by more than ``LANE16_LADJ_BOUND`` (99th percentile: ``LANE16_LADJ_Q99_BOUND``) -- ``Flow.inverse`` is the contract
        of ``flow.py:116-132``: an inverse (``inverse_guard=False`` switches the check off: measurements of the raw 16-bit
        sweep).  The guard is STATISTICAL, not a contract: it compares the two sweeps on a sample (latent images of the
        training rows at the end of ``fit``; at the head of every kernel call a strided sample of ALL walkers' proposals
        drawn from the step's own proposal law at the current sigma / mu -- ``mcmc._proposal_draws`` -- i.e. the
        heavy-tailed points ``mcmc.py:88`` actually inverts), so a rare outlier proposal can still see a 16-bit error
        beyond the bounds within a call.
        Default: float32 everywhere, like the reference."""
        self.n_dim = int(n_dim)
        if isinstance(flow, MAFSpec):
            spec = flow
            if spec.n_dim != self.n_dim:
                raise ValueError("MAFSpec.n_dim does not match n_dim")
        elif flow in SPEC_BY_NAME or flow in NSF_BY_NAME:
            spec = spec_by_name(self.n_dim, flow)
        else:
            raise ValueError("Invalid flow type. Choose from: maf3, maf6, maf12, nsf3, nsf6, nsf12, "
                             "or provide a MAFSpec object.")
        self.spec = spec
        self.lib = _lib.load()
        self.device = torch.device(device) if device is not None else _lib.require_gpu()
        if self.device.type != "cuda":
            raise _lib.PocomcAmdError("Flow lives on the GPU; there is no CPU fallback")
        if seed is None:
            seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())   # follows torch.manual_seed like zuko's init
        self.params = torch.from_numpy(spec.init_params(seed)).to(self.device)
        self._pack_idx = torch.from_numpy(spec.pack_index()).to(self.device)
        self._meta = torch.from_numpy(spec.device_meta()).to(self.device)
        self._packed = torch.zeros(spec.pk_size, dtype=torch.float32, device=self.device)
        self._desc = _lib.pmc_maf_t(
            packed=self._packed.data_ptr(), meta=self._meta.data_ptr(),
            D=spec.n_dim, H=spec.hidden, T=spec.n_transforms, Hp=spec.Hp, Dp=spec.Dp,
            nT=spec.nT, nXT=spec.nXT, nOT=spec.nOT, pk_per_transform=spec.pk_per_transform,
            tri_ok=int(spec.tri_ok), n_out=spec.n_out)
        self.inverse_algo = 0          # PMC_INVERSE_AUTO
        if precision not in ("f32", "bf16"):
            raise ValueError("precision must be 'f32' or 'bf16'")
        if precision == "bf16" and spec.univariate != "affine":
            raise NotImplementedError("the bf16 kernels are built for the affine flows")
        self.precision = precision
        if train_engine not in (None, "f32", "bf16"):
            raise ValueError("train_engine must be None, 'f32' or 'bf16'")
        self.train_engine = train_engine
        if inverse_precision is None:
            inverse_precision = "f32"          # (also what a checkpoint without the key reloads as)
        if inverse_precision not in ("f32", "bf16", "f16"):
            raise ValueError("inverse_precision must be 'f32', 'bf16' or 'f16'")
        if inverse_precision != "f32" and spec.univariate != "affine":
            raise NotImplementedError("the 16-bit helper products are built for the affine flows")
        self.inverse_precision = inverse_precision
        self._lane16 = None            # 16-bit image of the lane sweep's helper fragments (pmc_maf_pack_lane16)
        if inverse_precision != "f32":
            self._lane16 = torch.zeros(int(self.lib.pmc_maf_lane16_elems(C.byref(self._desc))), dtype=torch.int16,
                                       device=self.device)
            self._desc.lane16 = self._lane16.data_ptr()
            self._desc.lane16_fmt = 1 if inverse_precision == "bf16" else 2
        self._bf16 = None              # (gather map, image, elements per transform), built on first use
        self.inverse_guard = None      # last result of check_inverse_precision
        self.inverse_guard_enabled = bool(inverse_guard)
        self.repack()

    # ---------------------------------------------------------- checkpoints
    def __getstate__(self):
        """Plain tensors instead of device handles (the reference pickles the whole zuko module,
        sampler.py:1023-1049)."""
        return {"n_dim": self.n_dim,
                "spec": (self.spec.n_dim, self.spec.n_transforms, self.spec.hidden, self.spec.univariate, self.spec.bins),
                "params": self.params.detach().cpu().numpy(), "inverse_algo": self.inverse_algo,
                "precision": self.precision, "train_engine": self.train_engine,
                "inverse_precision": self.inverse_precision, "inverse_guard": self.inverse_guard_enabled,
                # the guard's last verdict travels with the parameters: a flow that fell back to float32 resumes in float32
                # (re-checking on reload would use other points than the run's and could decide otherwise)
                "inverse_fell_back": bool(self._lane16 is not None and not self._desc.lane16),
                "inverse_guard_result": self.inverse_guard}

    def __setstate__(self, st):
        spec = MAFSpec(*st["spec"])
        self.__init__(st["n_dim"], spec, precision=st.get("precision", "f32"), train_engine=st.get("train_engine"),
                      inverse_precision=st.get("inverse_precision"), inverse_guard=st.get("inverse_guard", True))
        self.set_params(st["params"], check=("inverse_fell_back" not in st))
        if "inverse_fell_back" in st and self._lane16 is not None:
            self.inverse_guard = st.get("inverse_guard_result")
            self._desc.lane16 = None if st["inverse_fell_back"] else self._lane16.data_ptr()
        self.inverse_algo = st.get("inverse_algo", 0)

    # ------------------------------------------------------------ parameters
    def repack(self):
        """Refresh the kernel-layout image after ``params`` changed."""
        with torch.cuda.device(self.device):
            _lib.check(self.lib.pmc_maf_pack(_lib.ptr(self.params), _lib.ptr(self._pack_idx),
                                             _lib.ptr(self._packed), self._packed.numel(),
                                             _lib.stream_handle()), "pmc_maf_pack")
            if getattr(self, "_lane16", None) is not None:
                _lib.check(self.lib.pmc_maf_pack_lane16(C.byref(self._desc), int(self._desc.lane16_fmt),
                                                        _lib.ptr(self._lane16), _lib.stream_handle()), "pmc_maf_pack_lane16")
            if getattr(self, "_bf16", None) is not None:
                idx, img, _ = self._bf16
                _lib.check(self.lib.pmc_maf_pack_bf16(_lib.ptr(self.params), _lib.ptr(idx), _lib.ptr(img), img.numel(),
                                                      _lib.stream_handle()), "pmc_maf_pack_bf16")

    def _bf16_image(self):
        """The bf16 fragment image of the current parameters (kept in step by ``repack``)."""
        if self._bf16 is None:
            idx = torch.from_numpy(self.spec.pack_index_bf16()).to(self.device)
            img = torch.zeros(idx.numel(), dtype=torch.int16, device=self.device)
            self._bf16 = (idx, img, self.spec.bf16_layout()["per_transform"])
            self.repack()
        return self._bf16

    def _forward_call(self, xd, z, ladj, lp, n):
        with torch.cuda.device(self.device):
            if self.precision == "bf16":
                _, img, per_t = self._bf16_image()
                _lib.check(self.lib.pmc_maf_forward_bf16(C.byref(self._desc), _lib.ptr(img), per_t, _lib.ptr(xd),
                                                         _lib.ptr(z) if z is not None else None,
                                                         _lib.ptr(ladj) if ladj is not None else None,
                                                         _lib.ptr(lp) if lp is not None else None, n, None,
                                                         _lib.stream_handle()), "pmc_maf_forward_bf16")
            else:
                _lib.check(self.lib.pmc_maf_forward(C.byref(self._desc), _lib.ptr(xd), _lib.ptr(z),
                                                    _lib.ptr(ladj) if ladj is not None else None,
                                                    _lib.ptr(lp) if lp is not None else None, n, _lib.stream_handle()),
                           "pmc_maf_forward")

    def set_params(self, flat, check=True):
        """New parameters; ``check``: run the 16-bit sweep's guard on them (a checkpoint reload restores the saved verdict
        instead)."""
        flat = torch.as_tensor(flat, dtype=torch.float32).reshape(-1)
        if flat.numel() != self.spec.n_params:
            raise ValueError("parameter vector has the wrong length")
        self.params.copy_(flat.to(self.device))
        self.repack()
        if check:
            self.check_inverse_precision()

    # ------------------------------------------------------------ the 16-bit sweep's safety net
    @property
    def inverse_precision_active(self):
        """What the sweep multiplies with right now: ``inverse_precision``, or "f32" after the guard fell back."""
        return self.inverse_precision if (self._lane16 is not None and self._desc.lane16) else "f32"

    @torch.no_grad()
    def check_inverse_precision(self, theta=None, rows=2048, bound=None, ladj_bound=None, ladj_q99_bound=None):
        """Run the 16-bit and the float32 lane sweep on ``theta`` (latent points of THIS flow: ``forward`` of its training
        rows at the end of ``fit``, standard-normal draws of a fixed generator otherwise -- what ``mcmc.py:88`` hands to
        ``flow.inverse`` once the flow fits) and compare walker by walker.  If the largest relative difference of x exceeds
        ``bound`` or the largest difference of the log-determinant ``ladj_bound`` (or the 16-bit sweep leaves rows
        non-finite that the float32 sweep does not), the flow falls back to the float32 sweep with a warning until its
        parameters change again.  Returns the comparison (also kept as ``inverse_guard``); None for float32 flows and for
        flows too narrow for the lane sweep."""
        if self._lane16 is None or not self.inverse_guard_enabled:
            return None
        bound = LANE16_BOUND if bound is None else float(bound)
        ladj_bound = LANE16_LADJ_BOUND if ladj_bound is None else float(ladj_bound)
        ladj_q99_bound = LANE16_LADJ_Q99_BOUND if ladj_q99_bound is None else float(ladj_q99_bound)
        self._desc.lane16 = self._lane16.data_ptr()              # (re-armed: new parameters get a new verdict)
        if not self.lib.pmc_maf_inverse_auto_is_lane(C.byref(self._desc)):
            self.inverse_guard = None                            # the narrow flows never take the 16-bit sweep
            return None
        if theta is None:
            g = torch.Generator().manual_seed(20240929)
            theta = torch.randn(int(rows), self.n_dim, generator=g, dtype=torch.float32)
        theta = theta[:max(int(rows), 1024)].to(self.device, torch.float32).contiguous()
        keep = self.inverse_algo
        try:
            self.inverse_algo = 9                                # PMC_INVERSE_TRIANGULAR_LANE16
            x16, l16 = self.inverse(theta)
            # the float32 side is what the flow would fall back to: AUTO without the 16-bit image (the float32 lane sweep,
            # or -- where its activations do not fit the LDS -- the two-wave sweep)
            self.inverse_algo = 0
            self._desc.lane16 = None
            x32, l32 = self.inverse(theta)
        finally:
            self.inverse_algo = keep
            self._desc.lane16 = self._lane16.data_ptr()
        ok32 = torch.isfinite(x32).all(dim=1) & torch.isfinite(l32)
        ok16 = torch.isfinite(x16).all(dim=1) & torch.isfinite(l16)
        lost = int((ok32 & ~ok16).sum().item())
        both = ok32 & ok16
        if bool(both.any()):
            ex = ((x16 - x32).abs().max(dim=1).values / x32.abs().max(dim=1).values.clamp_min(1e-30))[both]
            el = (l16 - l32).abs()[both]
            x_max, x_med, l_max, l_med = (float(ex.max().item()), float(ex.median().item()),
                                          float(el.max().item()), float(el.median().item()))
            l_q99 = float(torch.quantile(el.double(), 0.99).item())
            x_q99 = float(torch.quantile(ex.double(), 0.99).item())
        else:
            x_max = x_med = l_max = l_med = l_q99 = x_q99 = float("nan")
        passed = lost == 0 and x_max <= bound and l_max <= ladj_bound and l_q99 <= ladj_q99_bound
        self.inverse_guard = {"precision": self.inverse_precision, "rows": int(theta.shape[0]), "rows_compared": int(both.sum().item()),
                              "rows_lost_by_16bit": lost, "x_rel_err_max": x_max, "x_rel_err_median": x_med,
                              "ladj_abs_err_max": l_max, "ladj_abs_err_median": l_med, "ladj_abs_err_q99": l_q99,
                              "x_rel_err_q99": x_q99, "bound": bound, "ladj_bound": ladj_bound, "ladj_q99_bound": ladj_q99_bound,
                              "passed": bool(passed)}
        if not passed:
            self._desc.lane16 = None                             # AUTO takes the float32 helpers from here on
            warnings.warn(f"Flow: the {self.inverse_precision} inverse sweep is not an inverse of this flow within the bounds "
                          f"(max relative error on x {x_max:.3g} > {bound:g}, or on the log-determinant {l_max:.3g} > {ladj_bound:g} "
                          f"/ its 99th percentile {l_q99:.3g} > {ladj_q99_bound:g}, or {lost} rows lost); falling back to the float32 sweep.")
        return self.inverse_guard

    def state_dict(self):
        return {"params": self.params.detach().cpu().clone(), "n_dim": self.spec.n_dim,
                "n_transforms": self.spec.n_transforms, "hidden": self.spec.hidden}

    def load_state_dict(self, sd):
        self.set_params(sd["params"])

    # -------------------------------------------------------------- plumbing
    def _in(self, x):
        x = torch_double_to_float(x)
        if x.dim() != 2 or x.shape[1] != self.n_dim:
            raise ValueError(f"expected a (n, {self.n_dim}) tensor, got {tuple(x.shape)}")
        return x.to(self.device).contiguous(), x.device

    # -------------------------------------------------------------- contract
    @torch.no_grad()
    def forward(self, x):
        """``pocomc/flow.py:99-114``: data -> latent, returns ``(u, ladj)``."""
        xd, src = self._in(x)
        n = xd.shape[0]
        z = torch.empty_like(xd)
        ladj = torch.empty(n, dtype=torch.float32, device=self.device)
        self._forward_call(xd, z, ladj, None, n)
        return z.to(src), ladj.to(src)

    __call__ = forward

    @torch.no_grad()
    def inverse(self, u):
        """``pocomc/flow.py:116-132``: latent -> data, returns ``(x, ladj)``."""
        ud, src = self._in(u)
        n = ud.shape[0]
        x = torch.empty_like(ud)
        ladj = torch.empty(n, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.pmc_maf_inverse(C.byref(self._desc), _lib.ptr(ud), _lib.ptr(x), _lib.ptr(ladj),
                                                n, self.inverse_algo, _lib.stream_handle()), "pmc_maf_inverse")
        return x.to(src), ladj.to(src)

    @torch.no_grad()
    def log_prob(self, x):
        """``pocomc/flow.py:134-147``."""
        xd, src = self._in(x)
        n = xd.shape[0]
        z = torch.empty_like(xd)
        lp = torch.empty(n, dtype=torch.float32, device=self.device)
        self._forward_call(xd, z, None, lp, n)
        return lp.to(src)

    @torch.no_grad()
    def sample(self, size: int = 1, z=None):
        """``pocomc/flow.py:149-163``: ``(samples, log_prob)``.  ``z`` replays the
        base draw (tests)."""
        if z is None:
            # base draw from torch's global CPU generator, like zuko's rsample under torch.manual_seed
            z = torch.randn(int(size), self.n_dim, dtype=torch.float32)
        zd, src = self._in(z)
        x, ladj_inv = self.inverse(zd)
        base = -0.5 * (zd * zd).sum(dim=1) - 0.5 * self.n_dim * float(np.log(2 * np.pi))
        return x.to(src), (base - ladj_inv).to(src)

    def fit(self, x, weights=None, **kwargs):
        """``pocomc/flow.py:165-384``."""
        from .train import fit_flow
        return fit_flow(self, x, weights=weights, **kwargs)
