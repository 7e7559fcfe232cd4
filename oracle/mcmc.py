"""ORACLE (test infrastructure, never shipped, never measured as the product).

CPU restatement of the four MCMC kernels of ``pocomc/mcmc.py``:
``preconditioned_pcn`` (``:8-183``), ``preconditioned_rwm`` (``:186-341``),
``pcn`` (``:344-506``), ``rwm`` (``:508-654``).

Same contract as the reference -- ``f(state_dict, function_dict, option_dict)
-> dict`` -- plus two test-only keyword arguments:

``rng``    where the random variates come from.  ``LegacyStream`` draws from the
           global legacy ``np.random`` stream in exactly the reference's call
           order (``mcmc.py:80`` N gamma draws, ``:85`` N x ``randn(D)``,
           ``:137`` ``rand(N)``), so after ``np.random.seed(s)`` the oracle walks
           the reference's trajectory; it also records what it drew.
           ``Replay`` feeds recorded variates back (what the HIP path's replay
           entry points consume).
``trace``  optional list; one dict of per-step intermediates is appended per
           step (``theta_prime, u_prime, x_prime, ..., alpha, accept, sigma, mu``).

``exact=True`` evaluates the quadratic forms with the reference's per-particle
``np.dot`` calls (bit-identical to the reference, slow); the default is the
vectorised form (agrees to a few ulp).

Pinned against the reference itself by ``tests/golden/make_golden.py`` (runs
``pocomc.mcmc`` from ``/root/reference`` in the build container) and
``tests/test_oracle_golden.py``.
"""
from __future__ import annotations

import numpy as np


# --------------------------------------------------------------------- RNG
class LegacyStream:
    """Draw from the global legacy ``np.random`` stream in reference order and
    record every variate."""

    def __init__(self):
        self.record = []      # list of per-step dicts

    def begin_step(self):
        self.record.append({})

    def std_gamma(self, shape, n):
        # np.random.gamma(shape, scale) == scale * standard_gamma(shape) in the
        # legacy generator, one element after the other (mcmc.py:79-80)
        g = np.random.standard_gamma(shape, size=n)
        self.record[-1]["gamma"] = g
        return g

    def normal(self, n, d):
        z = np.random.randn(n, d)                 # mcmc.py:84-85, row after row
        self.record[-1]["z"] = z
        return z

    def uniform(self, n):
        u = np.random.rand(n)                     # mcmc.py:137
        self.record[-1]["u"] = u
        return u


class Replay:
    """Feed back variates recorded by ``LegacyStream`` (or any arrays)."""

    def __init__(self, record):
        self.record = record
        self.i = -1

    def begin_step(self):
        self.i += 1

    def std_gamma(self, shape, n):
        return self.record[self.i]["gamma"]

    def normal(self, n, d):
        return self.record[self.i]["z"]

    def uniform(self, n):
        return self.record[self.i]["u"]


class flow_numpy_wrapper:
    """``pocomc/tools.py:318-349``: numpy f64 -> torch f32 -> numpy; ``forward``
    returns ``-ladj`` and ``inverse`` ``+ladj`` (both are log|du/dtheta|)."""

    def __init__(self, flow):
        self.flow = flow

    def forward(self, v):
        import torch
        theta, logdetj = self.flow.forward(torch.tensor(v, dtype=torch.float32))
        return theta.detach().numpy(), -logdetj.detach().numpy()

    def inverse(self, theta):
        import torch
        v, logdetj = self.flow.inverse(torch.tensor(theta, dtype=torch.float32))
        return v.detach().numpy(), logdetj.detach().numpy()


# ----------------------------------------------------------------- helpers
def _quad(diff, inv_cov, exact):
    """``diff_k^T inv_cov diff_k`` per particle (mcmc.py:80, :127-129)."""
    if exact:
        return np.array([np.dot(d, np.dot(inv_cov, d)) for d in diff])
    return np.einsum("ki,ki->k", diff @ inv_cov.T, diff)


def _chol_apply(chol, z, exact):
    """``np.dot(chol, z_k)`` per particle (mcmc.py:85, :253)."""
    if exact:
        return np.array([np.dot(chol, zk) for zk in z])
    return z @ chol.T


def _evaluate(x_prime, logdetj_prime, log_prior, log_like, have_blobs, blobs):
    """finite mask -> prior -> likelihood gating, ``mcmc.py:100-121``."""
    n = len(x_prime)
    finite = np.isfinite(logdetj_prime) & np.isfinite(x_prime).all(axis=1)
    logp_prime = np.empty(n)
    logp_prime[finite] = log_prior(x_prime[finite])
    logp_prime[~finite] = -np.inf
    finite = finite & np.isfinite(logp_prime)
    logl_prime = np.empty(n)
    blobs_prime = None
    if have_blobs:
        blobs_prime = np.empty(n, dtype=np.dtype((blobs[0].dtype, blobs[0].shape)))
        logl_prime[finite], blobs_prime[finite] = log_like(x_prime[finite])
    else:
        logl_prime[finite], _ = log_like(x_prime[finite])
    logl_prime[~finite] = -np.inf
    return logp_prime, logl_prime, blobs_prime, finite


def _scaler_step(scaler, u_prime):
    """``mcmc.py:91-97``: inverse, then the boundary-condition round trip."""
    x_prime, logdetj_prime = scaler.inverse(u_prime)
    if (scaler.periodic is not None) or (scaler.reflective is not None):
        x_prime = scaler.apply_boundary_conditions_x(x_prime)
        u_prime = scaler.forward(x_prime, check_input=False)
        x_prime, logdetj_prime = scaler.inverse(u_prime)
    return u_prime, x_prime, logdetj_prime


def _run(kind, state_dict, function_dict, option_dict, rng, trace, exact):
    pre = kind in ("preconditioned_pcn", "preconditioned_rwm")
    tpcn = kind in ("preconditioned_pcn", "pcn")
    n_calls = 0
    u = np.copy(state_dict.get("u"))
    x = np.copy(state_dict.get("x"))
    logdetj = np.copy(state_dict.get("logdetj"))
    logl = np.copy(state_dict.get("logl"))
    logp = np.copy(state_dict.get("logp"))
    beta = state_dict.get("beta")
    blobs = state_dict.get("blobs")
    have_blobs = blobs is not None

    log_like = function_dict.get("loglike")
    log_prior = function_dict.get("logprior")
    scaler = function_dict.get("scaler")
    flow = flow_numpy_wrapper(function_dict.get("flow")) if pre else None
    geometry = function_dict.get("theta_geometry" if pre else "u_geometry")

    n_max = option_dict.get("n_max")
    n_steps = option_dict.get("n_steps")
    progress_bar = option_dict.get("progress_bar")
    sigma = option_dict.get("proposal_scale")
    if tpcn:
        sigma = np.minimum(sigma, 0.99)                      # mcmc.py:54, :380
    if rng is None:
        rng = LegacyStream()

    n_walkers, n_dim = x.shape

    if pre and kind == "preconditioned_pcn":
        theta, logdetj_flow = flow.forward(u)                # mcmc.py:60
    if tpcn:
        mu = geometry.t_mean
        cov = geometry.t_cov
        nu = geometry.t_nu
        inv_cov = np.linalg.inv(cov)                         # mcmc.py:67
        chol_cov = np.linalg.cholesky(cov)                   # mcmc.py:68
    else:
        chol = np.linalg.cholesky(geometry.normal_cov)       # mcmc.py:237-238
    if kind == "preconditioned_rwm":
        theta, logdetj_flow = flow.forward(u)                # mcmc.py:241
    if not pre:
        theta = u                                            # pcn/rwm move u directly
        logdetj_flow = np.zeros(n_walkers)

    with_ldj = not tpcn                                      # stop metric, :243 / :550
    logp2_val = np.mean(logl + logp + logdetj) if with_ldj else np.mean(logl + logp)
    cnt = 0
    i = 0
    while True:
        i += 1
        rng.begin_step()
        if tpcn:
            diff = theta - mu
            quad = _quad(diff, inv_cov, exact)
            g = rng.std_gamma((n_dim + nu) / 2, n_walkers)
            s = 1.0 / ((2.0 / (nu + quad)) * g)              # mcmc.py:80
            z = rng.normal(n_walkers, n_dim)
            theta_prime = (mu + (1.0 - sigma ** 2.0) ** 0.5 * diff
                           + (sigma * np.sqrt(s))[:, None] * _chol_apply(chol_cov, z, exact))
        else:
            z = rng.normal(n_walkers, n_dim)
            theta_prime = theta + sigma * _chol_apply(chol, z, exact)   # :253, :563

        if pre:
            u_prime, logdetj_flow_prime = flow.inverse(theta_prime)     # :88
        else:
            u_prime, logdetj_flow_prime = theta_prime, np.zeros(n_walkers)

        u_prime, x_prime, logdetj_prime = _scaler_step(scaler, u_prime)
        logp_prime, logl_prime, blobs_prime, finite_mask = _evaluate(
            x_prime, logdetj_prime, log_prior, log_like, have_blobs, blobs)
        n_calls += np.sum(finite_mask)

        expo = (logl_prime * beta - logl * beta + logp_prime - logp
                + logdetj_prime - logdetj)
        if pre:
            expo = expo + logdetj_flow_prime - logdetj_flow
        if tpcn:
            diff_prime = theta_prime - mu
            A = -(n_dim + nu) / 2 * np.log(1 + _quad(diff_prime, inv_cov, exact) / nu)
            B = -(n_dim + nu) / 2 * np.log(1 + quad / nu)
            expo = expo - A + B
        with np.errstate(over="ignore", invalid="ignore"):
            alpha = np.minimum(np.ones(n_walkers), np.exp(expo))
        alpha[np.isnan(alpha)] = 0.0

        u_rand = rng.uniform(n_walkers)
        mask = u_rand < alpha

        if pre:
            theta[mask] = theta_prime[mask]
            logdetj_flow[mask] = logdetj_flow_prime[mask]
        u[mask] = u_prime[mask]
        x[mask] = x_prime[mask]
        logdetj[mask] = logdetj_prime[mask]
        logl[mask] = logl_prime[mask]
        logp[mask] = logp_prime[mask]
        if have_blobs:
            blobs[mask] = blobs_prime[mask]
        if not pre:
            theta = u

        cap = np.minimum(2.38 / n_dim ** 0.5, 0.99)
        if tpcn:                                                          # :152, :476
            sigma = np.abs(np.minimum(sigma + 1 / (i + 1) ** 0.75 * (np.mean(alpha) - 0.234), cap))
        elif kind == "preconditioned_rwm":                                # :314
            sigma = sigma + 1 / (i + 1) * (np.mean(alpha) - 0.234)
        else:                                                             # :627
            sigma = np.abs(sigma + 1 / (i + 1) * (np.mean(alpha) - 0.234))
        if kind == "preconditioned_pcn":                                  # :156
            mu = mu + 1.0 / (i + 1.0) * (np.mean(theta, axis=0) - mu)

        if progress_bar is not None:
            progress_bar.update_stats(dict(
                calls=progress_bar.info["calls"] + np.sum(finite_mask),
                acc=np.mean(alpha), steps=i, logP=np.mean(logl + logp),
                eff=sigma / (2.38 / np.sqrt(n_dim))))

        if trace is not None:
            trace.append(dict(
                theta_prime=theta_prime.copy(), u_prime=np.array(u_prime, dtype=float),
                x_prime=x_prime.copy(), logdetj_prime=logdetj_prime.copy(),
                logdetj_flow_prime=np.array(logdetj_flow_prime, dtype=float),
                logp_prime=logp_prime.copy(), logl_prime=logl_prime.copy(),
                finite=finite_mask.copy(), alpha=alpha.copy(), accept=mask.copy(),
                sigma=float(sigma), mu=np.array(mu, dtype=float).copy() if tpcn else None,
                theta=np.array(theta, dtype=float).copy(), u=u.copy(), x=x.copy(),
                logdetj=logdetj.copy(), logl=logl.copy(), logp=logp.copy(),
                logdetj_flow=np.array(logdetj_flow, dtype=float).copy()))

        logp2_val_new = np.mean(logl + logp + logdetj) if with_ldj else np.mean(logl + logp)
        if logp2_val_new > logp2_val:
            cnt = 0
            logp2_val = logp2_val_new
        else:
            cnt += 1
            ratio = (2.38 / n_dim ** 0.5) / sigma
            if kind == "preconditioned_rwm":                              # :333
                ratio = np.minimum(1.0, ratio)
            if cnt >= n_steps * ratio ** 2.0:
                break
        if i >= n_max:
            break

    return dict(u=u, x=x, logdetj=logdetj, logl=logl, logp=logp, blobs=blobs,
                efficiency=sigma, accept=np.mean(alpha), steps=i, calls=n_calls,
                proposal_scale=sigma)


def preconditioned_pcn(state_dict, function_dict, option_dict, rng=None, trace=None, exact=False):
    """``pocomc/mcmc.py:8-183``."""
    return _run("preconditioned_pcn", state_dict, function_dict, option_dict, rng, trace, exact)


def preconditioned_rwm(state_dict, function_dict, option_dict, rng=None, trace=None, exact=False):
    """``pocomc/mcmc.py:186-341``."""
    return _run("preconditioned_rwm", state_dict, function_dict, option_dict, rng, trace, exact)


def pcn(state_dict, function_dict, option_dict, rng=None, trace=None, exact=False):
    """``pocomc/mcmc.py:344-506``."""
    return _run("pcn", state_dict, function_dict, option_dict, rng, trace, exact)


def rwm(state_dict, function_dict, option_dict, rng=None, trace=None, exact=False):
    """``pocomc/mcmc.py:508-654``."""
    return _run("rwm", state_dict, function_dict, option_dict, rng, trace, exact)
