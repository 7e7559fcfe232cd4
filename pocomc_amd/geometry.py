"""``Geometry`` -- what ``pocomc/geometry.py:31-59`` (+ ``pocomc/student.py:5-85``) hands the MCMC step
(``t_mean, t_cov, t_nu, normal_mean, normal_cov``: SURVEY.md section 8(a) G1), fitted on the device.

``fit`` takes the sample where it lives -- theta is the float32 output of ``flow.forward`` on the GPU, the pool's ``u``
a float64 device array -- and runs three reductions there (``csrc/pool.hip``): weighted first / second moments
(``pmc_moments``), the systematic resampling of ``geometry.py:52`` (``pmc_resample_systematic``) and the per-column
median of ``student.py:45`` (``pmc_column_medians``: one segmented radix sort).  Only ``D`` and ``D x D`` numbers
come back.

What ``fit_mvstud`` (``student.py:5-85``) really computes
--------------------------------------------------------
Its EM loop starts from ``mu = median``, ``Sigma = cov * (n-1)/n + diag(var)/n``, ``nu = 20`` (``:45-48``) and first
updates ``nu`` through ``opt_nu`` (``:35-42``): if ``func0(1e300) >= 0`` it returns ``nu = inf`` -- and ``fit_mvstud``
returns the START values (``:59-60``) -- otherwise it calls ``scipy.optimize.bisect(func0, 1e-300, 1e300)`` with the
default ``maxiter=100``, which cannot converge (100 halvings of a 1e300-wide bracket leave 7.9e269) and raises
``RuntimeError``.  At ``nu = 1e300`` every EM weight ``(nu + dim) / (nu + delta)`` is exactly 1.0 in float64, so
``func0(1e300)`` does not depend on the data at all: it is the constant evaluated in :func:`_func0_at_1e300` (0.0 with
this scipy: psi and log agree to the last bit at 5e299).  The reference's t-fit is therefore ALWAYS
``(median, start Sigma, inf)`` with ``t_nu`` replaced by 1e6 (``geometry.py:58-59``); a build whose libm made the
constant negative would raise from ``bisect`` on every call.  This class reproduces exactly that: the start values from
the device, the same constant test, the same error.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib


def _func0_at_1e300(dim: int) -> float:
    """``func0(1e300)`` of ``student.py:36-38`` with its data terms at their exact values (weights == 1.0):
    ``sum(log w) / n = 0.0`` and ``sum(w) / n = 1.0``; same operation order."""
    from scipy import special
    nu = 1e300
    return float(-special.psi(nu / 2) + np.log(nu / 2) + 0.0 - 1.0 + 1 + special.psi((nu + dim) / 2)
                 - np.log((nu + dim) / 2))


def _as_device(a, keep32=True):
    """Device view of a sample: float32 stays float32 (theta), everything else becomes float64."""
    dev = _lib.require_gpu()
    if isinstance(a, torch.Tensor):
        t = a
    else:
        a = np.asarray(a)
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32 if (keep32 and a.dtype == np.float32) else np.float64))
    if t.dtype != torch.float32 or not keep32:
        t = t.to(torch.float64)
    return t.to(dev).contiguous()


def moments(x, idx=None, w=None):
    """``(mean [D], S [D, D], V1, V2)`` of the rows ``x[idx]`` with weights ``w`` on the device (``pmc_moments``);
    the results come back as float64 numpy arrays."""
    lib = _lib.load()
    n = int(idx.numel()) if idx is not None else int(x.shape[0])
    D = int(x.shape[1])
    dev = x.device
    mean = torch.empty(D, dtype=torch.float64, device=dev)
    S = torch.empty(D, D, dtype=torch.float64, device=dev)
    v = torch.empty(2, dtype=torch.float64, device=dev)
    nbytes = int(lib.pmc_moments_workspace_bytes(D))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    f32 = x.dtype == torch.float32
    with torch.cuda.device(dev):
        _lib.check(lib.pmc_moments(None if f32 else _lib.ptr(x), _lib.ptr(x) if f32 else None,
                                   _lib.ptr(idx) if idx is not None else None, _lib.ptr(w) if w is not None else None,
                                   n, D, _lib.ptr(mean), _lib.ptr(S), _lib.ptr(v), _lib.ptr(ws), nbytes,
                                   _lib.stream_handle()), "pmc_moments")
    out = torch.cat([mean, S.reshape(-1), v]).cpu().numpy()
    return out[:D], out[D:D + D * D].reshape(D, D), float(out[-2]), float(out[-1])


def column_medians(x, idx=None):
    """``np.median(x[idx], axis=0)`` on the device (``pmc_column_medians``), in the input's precision."""
    lib = _lib.load()
    n = int(idx.numel()) if idx is not None else int(x.shape[0])
    D = int(x.shape[1])
    dev = x.device
    f32 = x.dtype == torch.float32
    med = torch.empty(D, dtype=x.dtype, device=dev)
    nbytes = int(lib.pmc_column_medians_workspace_bytes(n, D, int(f32)))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.pmc_column_medians(None if f32 else _lib.ptr(x), _lib.ptr(x) if f32 else None,
                                          _lib.ptr(idx) if idx is not None else None, n, D,
                                          None if f32 else _lib.ptr(med), _lib.ptr(med) if f32 else None, _lib.ptr(ws),
                                          nbytes, _lib.stream_handle()), "pmc_column_medians")
    return med.cpu().numpy()


class Geometry:
    """``pocomc/geometry.py:5-59``."""

    def __init__(self):
        self.normal_mean = self.normal_cov = self.t_mean = self.t_cov = self.t_nu = None

    def fit(self, theta, weights=None):
        """``theta``: (n, D) numpy array or torch tensor (device tensors are used in place); ``weights``: (n,) or None."""
        from .tools import systematic_resample
        th = _as_device(theta)
        n, D = int(th.shape[0]), int(th.shape[1])
        if weights is None:
            mean, S, _, _ = moments(th)
            self.normal_mean = mean                                        # np.mean(theta, axis=0)
            self.normal_cov = S / (n - 1)                                  # np.cov(theta.T)
            idx = None
        else:
            w = _as_device(weights, keep32=False)
            mean, S, v1, v2 = moments(th, None, w)
            self.normal_mean = mean                                        # np.average(theta, axis=0, weights=weights)
            self.normal_cov = S / (v1 - v2 / v1)                           # np.cov(theta.T, aweights=weights): ddof = 1
            idx = systematic_resample(n, weights=w, device_indices=True)   # geometry.py:52 (one np.random.random())
        # ---- fit_mvstud(sample): its start values are its result (module docstring)
        Ss = S if idx is None else moments(th, idx)[1]                     # (unweighted: the scatter matrix just formed)
        med = column_medians(th, idx)                                      # student.py:45
        var = np.diag(Ss) / n
        if th.dtype == torch.float32:
            var = var.astype(np.float32)                                   # np.var of a float32 array is a float32
        sigma = Ss / n + (1 / n) * np.diag(var)                            # student.py:46-47: cov*(n-1)/n + diag(var)/n
        if _func0_at_1e300(D) >= 0:                                        # student.py:39-40 -> :59-60
            nu = np.inf
        else:
            raise RuntimeError("Failed to converge after 100 iterations (scipy.optimize.bisect(func0, 1e-300, 1e300), "
                               "pocomc/student.py:42)")
        # The reference fails loudly on such input (NaN / inf rows: scipy's bisect raises on a NaN bracket; a singular
        # scatter matrix: linalg.solve in the EM step raises LinAlgError) -- the shortcut above must not turn that into a
        # silent NaN geometry that surfaces later, or never
        if not (np.isfinite(med).all() and np.isfinite(sigma).all()):
            raise ValueError("Geometry.fit: non-finite values in theta (median / scatter matrix are not finite)")
        # (the reference's own failure condition: ``linalg.solve(cov, ...)`` of the EM step, student.py:70, raises on an
        #  exactly singular matrix only -- near-singular or slightly indefinite scatter matrices pass there and pass here)
        try:
            np.linalg.solve(sigma, np.eye(D))
        except np.linalg.LinAlgError:
            raise np.linalg.LinAlgError("Geometry.fit: the scatter matrix of theta is singular (student.py:70 solves with it)")
        self.t_mean, self.t_cov, self.t_nu = med, sigma, nu
        if not np.isfinite(self.t_nu):
            self.t_nu = 1e6                                                # geometry.py:58-59
