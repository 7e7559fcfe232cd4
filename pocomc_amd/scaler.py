"""``Reparameterize`` -- host-side mirror of ``pocomc.scaler.Reparameterize``
(``pocomc/scaler.py:8-489``) in the configuration the Sampler uses
(``pocomc/sampler.py:309-313``: diagonal affine, ``scale=True``).

``forward`` / ``inverse`` run on the GPU (``pmc_scaler_forward`` /
``pmc_scaler_inverse``); inside an MCMC step the engine calls the same kernels
on device-resident arrays without leaving HBM.  Only the fit statistics
(``np.mean`` / ``np.std`` of the transformed prior draws, ``scaler.py:170-173``)
are taken on the host.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib


def assert_array_within_interval(x, left, right):
    """``pocomc/input_validation.py:24-52`` (closed interval)."""
    left = left.copy()
    left[np.isnan(left)] = -np.inf
    right = right.copy()
    right[np.isnan(right)] = np.inf
    if not np.all((left <= x) & (x <= right)):
        raise ValueError(f"Expected input to be within interval [{left}, {right}], "
                         f"but got minimum = {np.min(x)} and maximum = {np.max(x)}")


class Reparameterize:
    def __init__(self, n_dim, bounds=None, periodic=None, reflective=None,
                 transform="probit", scale=True, diagonal=True, device=None):
        self.ndim = int(n_dim)
        if bounds is None:
            bounds = np.full((self.ndim, 2), np.inf)                         # scaler.py:56-57
        elif len(bounds) == 2 and not np.shape(bounds) == (2, 2):
            bounds = np.tile(np.array(bounds, dtype=np.float32).reshape(2, 1), self.ndim).T
        bounds = np.asarray(bounds)
        if not np.issubdtype(bounds.dtype, np.floating):
            raise ValueError(f"Expected input to have dtype float, but got {bounds.dtype}")
        self.low = bounds.T[0].astype(np.float64)
        self.high = bounds.T[1].astype(np.float64)
        self.periodic = periodic
        self.reflective = reflective
        if transform not in ["logit", "probit"]:
            raise ValueError("Please provide a valid transformation function (e.g. logit or probit)")
        self.transform = transform
        self.scale = scale
        self.diagonal = diagonal
        self.mu = None
        self.sigma = None
        self.cov = self.L = self.L_inv = self.log_det_L = None               # diagonal=False: scaler.py:175-178
        lo_f, hi_f = np.isfinite(self.low), np.isfinite(self.high)            # scaler.py:463-489
        self.mask_none = ~lo_f & ~hi_f
        self.mask_right = ~lo_f & hi_f
        self.mask_left = lo_f & ~hi_f
        self.mask_both = lo_f & hi_f
        self.lib = _lib.load()
        self.device = torch.device(device) if device is not None else None
        self._dev = None
        self._desc = None

    def __getstate__(self):
        st = self.__dict__.copy()
        for k in ("lib", "device", "_dev", "_desc"):
            st.pop(k, None)
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)
        self.lib = _lib.load()
        self.device, self._dev, self._desc = None, None, None

    # ---------------------------------------------------------- descriptor
    def _descriptor(self, scale=None):
        """(Re)build the device image; ``scale`` overrides ``self.scale``."""
        if self.device is None:
            self.device = _lib.require_gpu()
        use_scale = self.scale if scale is None else scale
        D = self.ndim
        kind = np.zeros(D, dtype=np.int32)
        kind[self.mask_left] = 1
        kind[self.mask_right] = 2
        kind[self.mask_both] = 3
        bc = None
        if self.periodic is not None or self.reflective is not None:
            bc = np.zeros(D, dtype=np.int32)
            for i in (self.periodic or []):
                bc[i] |= 1
            for i in (self.reflective or []):
                bc[i] |= 2
        with np.errstate(invalid="ignore", divide="ignore"):
            log_width = np.where(self.mask_both, np.log(self.high - self.low), 0.0)
        mu = self.mu if self.mu is not None else np.zeros(D)
        sigma = self.sigma if self.sigma is not None else np.ones(D)
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(self.device)
        dev = dict(low=t(np.nan_to_num(self.low, posinf=np.inf, neginf=-np.inf), np.float64),
                   high=t(self.high, np.float64), mu=t(mu, np.float64), sigma=t(sigma, np.float64),
                   kind=t(kind, np.int32), log_width=t(log_width, np.float64),
                   bc=t(bc, np.int32) if bc is not None else None)
        desc = _lib.pmc_scaler_t(
            low=dev["low"].data_ptr(), high=dev["high"].data_ptr(), mu=dev["mu"].data_ptr(),
            sigma=dev["sigma"].data_ptr(), kind=dev["kind"].data_ptr(),
            bc=dev["bc"].data_ptr() if bc is not None else None, log_width=dev["log_width"].data_ptr(),
            D=D, logit=int(self.transform == "logit"), scale=int(bool(use_scale and self.mu is not None)),
            reserved=0, sum_log_sigma=float(np.sum(np.log(sigma))))                     # scaler.py:306
        return desc, dev

    def device_descriptor(self):
        """Descriptor (and the tensors it points to) for the MCMC engine."""
        if not self.diagonal:
            raise NotImplementedError("the MCMC step kernels fuse the DIAGONAL affine map (what pocomc's Sampler "
                                      "configures, sampler.py:320-327); diagonal=False scalers transform arrays only")
        if self._desc is None:
            self._desc, self._dev = self._descriptor()
        return self._desc

    # ------------------------------------------------------------------ fit
    def fit(self, x):
        """``scaler.py:159-178``."""
        x = np.asarray(x, dtype=np.float64)
        assert_array_within_interval(x, self.low, self.high)
        self.mu, self.sigma = None, None
        u = self._forward_device(x, scale=False)
        self.mu = np.mean(u, axis=0)
        if self.diagonal:
            self.sigma = np.std(u, axis=0)
        else:                                                                # scaler.py:175-178
            self.cov = np.cov(u.T)
            self.L = np.linalg.cholesky(self.cov)
            self.L_inv = np.linalg.inv(self.L)
            self.log_det_L = np.linalg.slogdet(self.L)[1]
        self._desc = None

    def _affine_rows(self, M, a, mode):
        """``mu + L a`` (mode 0) or ``L^-1 (a - mu)`` (mode 1) row by row on the device (``pmc_affine_rows``)."""
        up = lambda v: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float64)).to(self.device)
        Md, mud, ad = up(M), up(self.mu), up(a)
        out = torch.empty_like(ad)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.pmc_affine_rows(_lib.ptr(Md), _lib.ptr(mud), _lib.ptr(ad), _lib.ptr(out), ad.shape[0],
                                                self.ndim, mode, _lib.stream_handle()), "pmc_affine_rows")
        return out.cpu().numpy()

    def _forward_device(self, x, scale=None):
        desc, dev = self._descriptor(scale)
        xd = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64)).to(self.device)
        ud = torch.empty_like(xd)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.pmc_scaler_forward(C.byref(desc), _lib.ptr(xd), _lib.ptr(ud), xd.shape[0],
                                                   _lib.stream_handle()), "pmc_scaler_forward")
        return ud.cpu().numpy()

    def forward(self, x, check_input=True):
        """``scaler.py:180-202``."""
        x = np.asarray(x, dtype=np.float64)
        if check_input:
            assert_array_within_interval(x, self.low, self.high)
        if self.diagonal or not self.scale or self.mu is None:
            return self._forward_device(x)
        return self._affine_rows(self.L_inv, self._forward_device(x, scale=False), 1)      # scaler.py:199-200, :291-292

    def inverse(self, u):
        """``scaler.py:204-226``: ``(x, log_det_J)`` (no boundary conditions here,
        exactly like the reference method)."""
        full = (not self.diagonal) and self.scale and self.mu is not None
        if self.device is None:
            self.device = _lib.require_gpu()
        if full:
            u = self._affine_rows(self.L, u, 0)                               # scaler.py:219, :311-313
        desc, dev = self._descriptor(scale=False if full else None)
        desc.bc = None
        ud = torch.from_numpy(np.ascontiguousarray(u, dtype=np.float64)).to(self.device)
        n = ud.shape[0]
        xo = torch.empty_like(ud)
        uo = torch.empty_like(ud)
        ldj = torch.empty(n, dtype=torch.float64, device=self.device)
        fin = torch.empty(n, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.pmc_scaler_inverse(C.byref(desc), None, _lib.ptr(ud), _lib.ptr(uo), _lib.ptr(xo),
                                                   None, _lib.ptr(ldj), _lib.ptr(fin), n, _lib.stream_handle()),
                       "pmc_scaler_inverse")
        ldj = ldj.cpu().numpy()
        if full:
            ldj = self.log_det_L * np.ones(len(ldj)) + ldj                    # scaler.py:220-221
        return xo.cpu().numpy(), ldj
