"""ORACLE (test infrastructure, never shipped, never measured as the product).

CPU restatement of the particle math around the MCMC step:
``pocomc/tools.py`` (ESS / USS / trim / systematic resample),
``pocomc/particles.py:215-231`` (persistent-sampling log-weights and logZ),
``pocomc/geometry.py`` + ``pocomc/student.py`` (the once-per-iteration
Student-t fit whose outputs are step inputs) and the resampling draw of
``pocomc/sampler.py:680-715``.

Pinned by the reference's only known-answer test (``tests/test_tools.py:10-14``:
``compute_ess`` of a single particle == 1.0) and by golden vectors generated
from the reference's importable modules (``tests/golden/make_golden.py``).
"""
from __future__ import annotations

import math

import numpy as np

SQRTEPS = math.sqrt(float(np.finfo(np.float64).eps))


def trim_weights(samples, weights, ess=0.99, bins=1000):
    """``pocomc/tools.py:10-53``.  Mutates ``weights`` in place like the reference."""
    weights /= np.sum(weights)
    ess_total = 1.0 / np.sum(weights ** 2.0)
    percentiles = np.linspace(0, 99, bins)
    i = bins - 1
    while True:
        p = percentiles[i]
        threshold = np.percentile(weights, p)
        mask = weights >= threshold
        weights_trimmed = weights[mask]
        weights_trimmed /= np.sum(weights_trimmed)
        ess_trimmed = 1.0 / np.sum(weights_trimmed ** 2.0)
        if ess_trimmed / ess_total >= ess:
            break
        i -= 1
    return samples[mask], weights_trimmed


def effective_sample_size(weights):
    """``pocomc/tools.py:56-71`` (in-place normalisation, then ``1/sum w^2``)."""
    weights /= np.sum(weights)
    return 1.0 / np.sum(weights ** 2.0)


def unique_sample_size(weights, k=None):
    """``pocomc/tools.py:74-93``."""
    if k is None:
        k = len(weights)
    weights /= np.sum(weights)
    return np.sum(1.0 - (1.0 - weights) ** k)


def compute_ess(logw):
    """``pocomc/tools.py:96-114``."""
    logw_max = np.max(logw)
    logw_normed = logw - logw_max
    weights = np.exp(logw_normed) / np.sum(np.exp(logw_normed))
    return 1.0 / np.sum(weights * weights) / len(weights)


def increment_logz(logw):
    """``pocomc/tools.py:117-133``."""
    logw_max = np.max(logw)
    logw_normed = logw - logw_max
    return logw_max + np.logaddexp.reduce(logw_normed)


def systematic_resample(size, weights, random_state=None, offset=None):
    """``pocomc/tools.py:136-186``.  ``offset`` replays the single uniform of
    ``:175`` (test-only); otherwise it is drawn from the legacy stream."""
    if random_state is not None:
        np.random.seed(random_state)
    if abs(np.sum(weights) - 1.) > SQRTEPS:
        weights = np.array(weights) / np.sum(weights)
    if offset is None:
        offset = np.random.random()
    positions = (offset + np.arange(size)) / size
    j = 0
    cumulative_sum = weights[0]
    indeces = np.empty(size, dtype=int)
    for i in range(size):
        while positions[i] > cumulative_sum:
            j += 1
            cumulative_sum += weights[j]
        indeces[i] = j
    return indeces


def multinomial_resample(size, weights, uniforms=None):
    """``pocomc/sampler.py:703``: ``np.random.choice(len(w), size, p=w)``.

    numpy's legacy ``choice`` with ``p`` is ``cdf = p.cumsum(); cdf /= cdf[-1];
    idx = cdf.searchsorted(random_sample(size), side='right')``; ``uniforms``
    replays the draws (test-only)."""
    cdf = np.cumsum(weights)
    cdf /= cdf[-1]
    if uniforms is None:
        uniforms = np.random.random_sample(size)
    return cdf.searchsorted(uniforms, side="right")


def compute_logw_and_logz(logl, beta, logz, beta_final=1.0, normalize=True):
    """``pocomc/particles.py:215-231`` on plain arrays: ``logl`` is ``(T, N)``,
    ``beta`` and ``logz`` are ``(T,)``."""
    logl = np.asarray(logl)
    A = logl * beta_final
    b = np.array([logl * beta[i] - logz[i] for i in range(len(beta))])
    B = np.logaddexp.reduce(b, axis=0) - np.log(len(beta))
    logw = A - B
    logw = np.concatenate(logw)
    logz_new = np.logaddexp.reduce(logw) - np.log(len(logw))
    if normalize:
        logw -= np.logaddexp.reduce(logw)
    return logw, logz_new


# ------------------------------------------------------------------ geometry
def fit_mvstud(data, tolerance=1e-6, max_iter=100):
    """``pocomc/student.py:5-85``: EM fit of a multivariate Student-t."""
    from scipy import optimize, special

    def opt_nu(delta_iobs, nu):
        def func0(nu):
            w_iobs = (nu + dim) / (nu + delta_iobs)
            return (-special.psi(nu / 2) + np.log(nu / 2) + np.sum(np.log(w_iobs)) / n
                    - np.sum(w_iobs) / n + 1 + special.psi((nu + dim) / 2) - np.log((nu + dim) / 2))
        if func0(1e300) >= 0:
            return np.inf
        return optimize.bisect(func0, 1e-300, 1e300)

    data = data.T
    (dim, n) = data.shape
    mu = np.array([np.median(data, 1)]).T
    Sigma = np.cov(data) * (n - 1) / n + (1 / n) * np.diag(np.var(data, axis=1))
    nu = 20
    last_nu = 0
    i = 0
    while np.abs(last_nu - nu) > tolerance and i < max_iter:
        i += 1
        diffs = data - mu
        delta_iobs = np.sum(diffs * np.linalg.solve(Sigma, diffs), 0)
        last_nu = nu
        nu = opt_nu(delta_iobs, nu)
        if nu == np.inf:
            return mu.T[0], Sigma, nu
        w_iobs = (nu + dim) / (nu + delta_iobs)
        Sigma = np.dot(w_iobs * diffs, diffs.T) / n
        mu = np.sum(w_iobs * data, 1) / sum(w_iobs)
        mu = np.array([mu]).T
    return mu.T[0], Sigma, nu


class Geometry:
    """``pocomc/geometry.py:5-59``."""

    def __init__(self):
        self.normal_mean = None
        self.normal_cov = None
        self.t_mean = None
        self.t_cov = None
        self.t_nu = None

    def fit(self, theta, weights=None):
        if weights is None:
            self.normal_mean = np.mean(theta, axis=0)
            self.normal_cov = np.cov(theta.T)
            self.t_mean, self.t_cov, self.t_nu = fit_mvstud(theta)
        else:
            self.normal_mean = np.average(theta, axis=0, weights=weights)
            self.normal_cov = np.cov(theta.T, aweights=weights)
            idx = systematic_resample(len(theta), weights=weights)
            self.t_mean, self.t_cov, self.t_nu = fit_mvstud(theta[idx])
        if ~np.isfinite(self.t_nu):
            self.t_nu = 1e6
