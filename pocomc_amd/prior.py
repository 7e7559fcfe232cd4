"""``Prior`` -- host-side mirror of ``pocomc/prior.py``: a product of frozen
``scipy.stats`` distributions.  It is a host black box like the likelihood
(SURVEY.md section 2 row 8); uniform / normal factors are evaluated with one vectorised numpy
expression instead of one scipy call per dimension (same values)."""
from __future__ import annotations

import numpy as np


class Prior:
    def __init__(self, dists=None):
        self.dists = dists
        self._fast = None
        try:
            kinds = [d.dist.name for d in dists]
            if all(k in ("uniform", "norm") for k in kinds):
                loc = np.array([d.kwds.get("loc", d.args[0] if len(d.args) > 0 else 0.0) for d in dists], float)
                scale = np.array([d.kwds.get("scale", d.args[1] if len(d.args) > 1 else 1.0) for d in dists], float)
                self._fast = (np.array([k == "uniform" for k in kinds]), loc, scale)
        except Exception:
            self._fast = None

    # ---------------------------------------------------------------- device
    def device_descriptor(self, device=None):
        """``pmc_prior_t`` for the MCMC engine, or ``None`` when a factor is not a family the device
        evaluates (then ``logpdf`` is called on the host like any black box)."""
        if self._fast is None:
            return None
        if getattr(self, "_ddesc", None) is None:
            import torch
            from . import _lib
            dev = device if device is not None else _lib.require_gpu()
            is_u, loc, scale = self._fast
            fam = np.where(is_u, 1, 2).astype(np.int32)
            self._dtensors = [torch.from_numpy(a).to(dev) for a in (fam, loc.copy(), scale.copy())]
            self._ddesc = _lib.pmc_prior_t(family=self._dtensors[0].data_ptr(), loc=self._dtensors[1].data_ptr(),
                                           scale=self._dtensors[2].data_ptr(), D=len(fam), reserved=0)
        return self._ddesc

    def __getstate__(self):
        st = self.__dict__.copy()
        st.pop("_ddesc", None); st.pop("_dtensors", None)
        return st

    def logpdf(self, x):
        """``pocomc/prior.py:70-100``."""
        if self._fast is not None:
            is_u, loc, scale = self._fast
            x = np.asarray(x, dtype=float)
            out = np.zeros(len(x))
            if is_u.any():
                xu = x[:, is_u]
                inside = np.all((xu >= loc[is_u]) & (xu <= loc[is_u] + scale[is_u]), axis=1)
                out += np.where(inside, -np.sum(np.log(scale[is_u])), -np.inf)
            if (~is_u).any():
                z = (x[:, ~is_u] - loc[~is_u]) / scale[~is_u]
                out += np.sum(-0.5 * z * z - np.log(scale[~is_u]) - 0.5 * np.log(2 * np.pi), axis=1)
            return out
        logp = np.zeros(len(x))
        for i, dist in enumerate(self.dists):
            logp += dist.logpdf(x[:, i])
        return logp

    def rvs(self, size=1):
        """``pocomc/prior.py:102-133``."""
        return np.transpose([dist.rvs(size=size) for dist in self.dists])

    @property
    def bounds(self):
        """``pocomc/prior.py:135-153``."""
        return np.array([dist.support() for dist in self.dists])

    @property
    def dim(self):
        return len(self.dists)
