"""``Geometry`` -- host-side mirror of ``pocomc/geometry.py`` + ``pocomc/student.py``.

Fitted once per SMC iteration on a (n, D) sample; D x D linear algebra and a scalar
bisection: SURVEY.md section 2 row 6 keeps it on the host.  Its outputs
(``t_mean, t_cov, t_nu, normal_cov``) are inputs of the device step (section 8(a) G1).
"""
from __future__ import annotations

import numpy as np
from scipy import optimize, special


def fit_mvstud(data, tolerance=1e-6, max_iter=100):
    """EM fit of a multivariate Student-t (``pocomc/student.py:5-85``)."""
    data = np.asarray(data).T
    dim, n = data.shape

    def nu_root(delta):
        def f(nu):
            w = (nu + dim) / (nu + delta)
            return (-special.psi(nu / 2) + np.log(nu / 2) + np.sum(np.log(w)) / n - np.sum(w) / n + 1
                    + special.psi((nu + dim) / 2) - np.log((nu + dim) / 2))
        return np.inf if f(1e300) >= 0 else optimize.bisect(f, 1e-300, 1e300)

    mu = np.median(data, 1)[:, None]
    Sigma = np.cov(data) * (n - 1) / n + (1 / n) * np.diag(np.var(data, axis=1))
    nu, last_nu, it = 20, 0, 0
    while np.abs(last_nu - nu) > tolerance and it < max_iter:
        it += 1
        diffs = data - mu
        delta = np.sum(diffs * np.linalg.solve(Sigma, diffs), 0)
        last_nu, nu = nu, nu_root(delta)
        if nu == np.inf:
            return mu[:, 0], Sigma, nu
        w = (nu + dim) / (nu + delta)
        Sigma = np.dot(w * diffs, diffs.T) / n
        mu = (np.sum(w * data, 1) / np.sum(w))[:, None]
    return mu[:, 0], Sigma, nu


class Geometry:
    """``pocomc/geometry.py:5-59``."""

    def __init__(self):
        self.normal_mean = self.normal_cov = self.t_mean = self.t_cov = self.t_nu = None

    def fit(self, theta, weights=None):
        from .tools import systematic_resample
        if weights is None:
            self.normal_mean = np.mean(theta, axis=0)
            self.normal_cov = np.cov(theta.T)
            sample = theta
        else:
            self.normal_mean = np.average(theta, axis=0, weights=weights)
            self.normal_cov = np.cov(theta.T, aweights=weights)
            sample = theta[systematic_resample(len(theta), weights=weights)]
        self.t_mean, self.t_cov, self.t_nu = fit_mvstud(sample)
        if not np.isfinite(self.t_nu):
            self.t_nu = 1e6                                            # geometry.py:58-59
