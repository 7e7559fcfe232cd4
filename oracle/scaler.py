"""ORACLE (test infrastructure, never shipped, never measured as the product).

CPU restatement of ``pocomc/scaler.py`` (``Reparameterize``): the bounded <->
unbounded bijector that sits inside every MCMC step (``pocomc/mcmc.py:91-97``).

Pinned against the reference itself: ``tests/golden/make_golden.py`` runs the
reference class on the four bound types of ``tests/test_scaler.py:9-54`` and
``tests/test_oracle_golden.py`` compares.
"""
from __future__ import annotations

import numpy as np
from scipy.special import erf, erfinv


class Reparameterize:
    """``pocomc/scaler.py:8-489`` (diagonal and full affine, probit / logit)."""

    def __init__(self, n_dim, bounds=None, periodic=None, reflective=None,
                 transform="probit", scale=True, diagonal=True):
        self.ndim = n_dim
        if bounds is None:
            bounds = np.full((self.ndim, 2), np.inf)                       # scaler.py:56-57
        elif len(bounds) == 2 and not np.shape(bounds) == (2, 2):
            bounds = np.tile(np.array(bounds, dtype=np.float32).reshape(2, 1), self.ndim).T
        bounds = np.asarray(bounds)
        if not np.issubdtype(bounds.dtype, np.floating):
            raise ValueError(f"Expected input to have dtype float, but got {bounds.dtype}")
        self.low = bounds.T[0]
        self.high = bounds.T[1]
        self.periodic = periodic
        self.reflective = reflective
        if transform not in ["logit", "probit"]:
            raise ValueError("Please provide a valid transformation function (e.g. logit or probit)")
        self.transform = transform
        self.mu = None
        self.sigma = None
        self.cov = None
        self.L = None
        self.L_inv = None
        self.log_det_L = None
        self.scale = scale
        self.diagonal = diagonal
        lo_f, hi_f = np.isfinite(self.low), np.isfinite(self.high)           # scaler.py:463-489
        self.mask_none = ~lo_f & ~hi_f
        self.mask_right = ~lo_f & hi_f
        self.mask_left = lo_f & ~hi_f
        self.mask_both = lo_f & hi_f

    # ------------------------------------------------ boundary conditions
    def apply_boundary_conditions_x(self, x):
        """``scaler.py:84-107``."""
        if (self.periodic is None) and (self.reflective is None):
            return x
        elif self.periodic is None:
            return self._reflect(x)
        elif self.reflective is None:
            return self._wrap(x)
        return self._reflect(self._wrap(x))

    def _wrap(self, x):
        """``scaler.py:109-132``: the ``while`` loops, element by element."""
        x = x.copy()
        for i in self.periodic:
            for j in range(len(x)):
                while x[j, i] > self.high[i]:
                    x[j, i] = self.low[i] + x[j, i] - self.high[i]
                while x[j, i] < self.low[i]:
                    x[j, i] = self.high[i] + x[j, i] - self.low[i]
        return x

    def _reflect(self, x):
        """``scaler.py:134-157``."""
        x = x.copy()
        for i in self.reflective:
            for j in range(len(x)):
                while x[j, i] > self.high[i]:
                    x[j, i] = self.high[i] - x[j, i] + self.high[i]
                while x[j, i] < self.low[i]:
                    x[j, i] = self.low[i] + self.low[i] - x[j, i]
        return x

    # ---------------------------------------------------------------- fit
    def fit(self, x):
        """``scaler.py:159-178``."""
        _assert_within(x, self.low, self.high)
        u = self._forward(x)
        self.mu = np.mean(u, axis=0)
        if self.diagonal:
            self.sigma = np.std(u, axis=0)
        else:
            self.cov = np.cov(u.T)
            self.L = np.linalg.cholesky(self.cov)
            self.L_inv = np.linalg.inv(self.L)
            self.log_det_L = np.linalg.slogdet(self.L)[1]

    def forward(self, x, check_input=True):
        """``scaler.py:180-202``."""
        if check_input:
            _assert_within(x, self.low, self.high)
        u = self._forward(x)
        if self.scale:
            u = self._forward_affine(u)
        return u

    def inverse(self, u):
        """``scaler.py:204-226``."""
        if self.scale:
            x, log_det_J = self._inverse_affine(u)
            x, log_det_J_prime = self._inverse(x)
            log_det_J += log_det_J_prime
        else:
            x, log_det_J = self._inverse(u)
        return x, log_det_J

    # ------------------------------------------------------ bound transforms
    def _forward(self, x):
        """``scaler.py:228-247``."""
        u = np.empty(x.shape)
        m = self.mask_none
        u[:, m] = x[:, m]
        m = self.mask_left
        u[:, m] = np.log(x[:, m] - self.low[m])                               # :327
        m = self.mask_right
        u[:, m] = np.log(self.high[m] - x[:, m])                              # :358
        m = self.mask_both
        p = (x[:, m] - self.low[m]) / (self.high[m] - self.low[m])
        # scaler.py:393 calls np.clip and discards the result: no clipping
        if self.transform == "logit":
            u[:, m] = np.log(p / (1.0 - p))
        else:
            u[:, m] = np.sqrt(2.0) * erfinv(2.0 * p - 1.0)
        return u

    def _inverse(self, u):
        """``scaler.py:249-271``, ``:329-425``."""
        x = np.empty(u.shape)
        J = np.empty(u.shape)
        m = self.mask_none
        x[:, m], J[:, m] = u[:, m], 0.0
        m = self.mask_left
        x[:, m], J[:, m] = np.exp(u[:, m]) + self.low[m], u[:, m]
        m = self.mask_right
        x[:, m], J[:, m] = self.high[m] - np.exp(u[:, m]), u[:, m]
        m = self.mask_both
        w = self.high[m] - self.low[m]
        if self.transform == "logit":
            p = np.exp(-np.logaddexp(0, -u[:, m]))
            x[:, m] = p * w + self.low[m]
            J[:, m] = np.log(w) + np.log(p) + np.log(1.0 - p)
        else:
            p = (erf(u[:, m] / np.sqrt(2.0)) + 1.0) / 2.0
            x[:, m] = p * w + self.low[m]
            J[:, m] = np.log(w) + (-u[:, m] ** 2.0 / 2.0) - np.log(np.sqrt(2.0 * np.pi))
        return x, np.sum(J, axis=1)

    def _forward_affine(self, x):
        """``scaler.py:273-289``."""
        if self.diagonal:
            return (x - self.mu) / self.sigma
        return np.array([np.dot(self.L_inv, xi - self.mu) for xi in x])

    def _inverse_affine(self, u):
        """``scaler.py:291-313``."""
        if self.diagonal:
            log_det_J = np.sum(np.log(self.sigma))
            return self.mu + self.sigma * u, log_det_J * np.ones(len(u))
        x = self.mu + np.array([np.dot(self.L, ui) for ui in u])
        return x, self.log_det_L * np.ones(len(u))


def _assert_within(x, left, right):
    """``pocomc/input_validation.py:24-52`` (closed interval)."""
    left = left.copy()
    left[np.isnan(left)] = -np.inf
    right = right.copy()
    right[np.isnan(right)] = np.inf
    if not np.all((left <= x) & (x <= right)):
        raise ValueError(f"Expected input to be within interval [{left}, {right}], "
                         f"but got minimum = {np.min(x)} and maximum = {np.max(x)}")
