"""Seeded synthetic inputs shared by the golden generator and the tests.

Nothing here comes from the reference: targets, priors and initial particle
states are defined by SURVEY.md section 8(d) (Rosenbrock ``README.md:53-55``
formula, correlated Gaussian, Gaussian mixture) and seeded numpy generators.
"""
from __future__ import annotations

import numpy as np

from pocomc_amd.maf_spec import MAFSpec


# ------------------------------------------------------------------ targets
def rosenbrock(x):
    return -np.sum(10.0 * (x[:, ::2] ** 2.0 - x[:, 1::2]) ** 2.0 + (x[:, ::2] - 1.0) ** 2.0, axis=1)


def make_corr_gauss(D, rho=0.95):
    C = np.full((D, D), rho) + (1 - rho) * np.eye(D)
    Ci = np.linalg.inv(C)

    def f(x):
        return -0.5 * np.einsum("ki,ij,kj->k", x, Ci, x)
    return f


def std_gauss(x):
    return -0.5 * np.sum(x ** 2, axis=1)


def make_bimodal(D, sep=3.0):
    def f(x):
        a = -0.5 * np.sum((x - sep) ** 2, axis=1)
        b = -0.5 * np.sum((x + sep) ** 2, axis=1)
        return np.logaddexp(a, b) - np.log(2.0)
    return f


class UniformPrior:
    """Product of U(low, high): the host-side ``logprior`` black box."""

    def __init__(self, low, high, D):
        self.low, self.high, self.D = float(low), float(high), D
        self.bounds = np.tile(np.array([[self.low, self.high]]), (D, 1))

    def logpdf(self, x):
        inside = np.all((x >= self.low) & (x <= self.high), axis=1)
        return np.where(inside, -self.D * np.log(self.high - self.low), -np.inf)

    def rvs(self, n, rng):
        return rng.uniform(self.low, self.high, size=(n, self.D))


class NormalPrior:
    def __init__(self, scale, D):
        self.scale, self.D = float(scale), D
        self.bounds = np.tile(np.array([[-np.inf, np.inf]]), (D, 1))

    def logpdf(self, x):
        return np.sum(-0.5 * (x / self.scale) ** 2 - np.log(self.scale) - 0.5 * np.log(2 * np.pi), axis=1)

    def rvs(self, n, rng):
        return rng.normal(0.0, self.scale, size=(n, self.D))


class HalfBoundPrior:
    """Mixed bounds: dims cycle through none / left / right / both."""

    def __init__(self, D):
        self.D = D
        b = []
        for j in range(D):
            b.append([(-np.inf, np.inf), (0.0, np.inf), (-np.inf, 2.0), (-1.0, 3.0)][j % 4])
        self.bounds = np.array(b, dtype=float)

    def logpdf(self, x):
        inside = np.all((x >= self.bounds[:, 0]) & (x <= self.bounds[:, 1]), axis=1)
        return np.where(inside, -0.5 * np.sum((x - 0.7) ** 2, axis=1), -np.inf)

    def rvs(self, n, rng):
        x = np.empty((n, self.D))
        for j in range(self.D):
            k = j % 4
            if k == 0:
                x[:, j] = rng.normal(0.7, 1.0, n)
            elif k == 1:
                x[:, j] = rng.gamma(2.0, 0.7, n)
            elif k == 2:
                x[:, j] = 2.0 - rng.gamma(2.0, 0.7, n)
            else:
                x[:, j] = rng.uniform(-1.0, 3.0, n)
        return x


class CutPrior(HalfBoundPrior):
    """HalfBoundPrior whose density vanishes for x_0 > cut INSIDE the scaler's bounds: ``logpdf`` returns -inf on rows
    the bijector cannot exclude (the non-finite-prior branch of ``sampler.py:898-901`` / ``mcmc.py:104-109``)."""

    def __init__(self, D, cut=0.9):
        super().__init__(D)
        self.cut = cut

    def logpdf(self, x):
        return np.where(x[:, 0] > self.cut, -np.inf, super().logpdf(x))


class Geo:
    """Stand-in for ``pocomc.geometry.Geometry`` outputs (step inputs, G1)."""

    def __init__(self, t_mean, t_cov, t_nu, normal_cov):
        self.t_mean, self.t_cov, self.t_nu, self.normal_cov = t_mean, t_cov, t_nu, normal_cov


def flow_params(spec: MAFSpec, seed: int, gain: float = 1.2):
    """Deterministic non-trivial flow: the default init scaled by ``gain``."""
    return (spec.init_params(seed) * np.float32(gain)).astype(np.float32)


# -------------------------------------------------------------- MCMC cases
MCMC_CASES = {
    # name: kind, N, D, T, beta, nu, prior, target, seed, n_max, bc
    "tpcn_n64_d4_uniform":   dict(kind="preconditioned_pcn", N=64,  D=4,  T=3, beta=0.1, nu=5.0, prior="uniform", target="rosenbrock", seed=0, n_max=6),
    "tpcn_n256_d10_normal":  dict(kind="preconditioned_pcn", N=256, D=10, T=3, beta=1.0, nu=1e6, prior="normal",  target="gauss",      seed=1, n_max=5),
    "tpcn_n512_d32_uniform": dict(kind="preconditioned_pcn", N=512, D=32, T=3, beta=0.5, nu=5.0, prior="uniform", target="corr",       seed=0, n_max=3),
    "tpcn_n128_d6_mixed_bc": dict(kind="preconditioned_pcn", N=128, D=6,  T=3, beta=0.7, nu=7.5, prior="mixed",   target="gauss",      seed=2, n_max=4, periodic=[3], reflective=None),
    "tpcn_n96_d5_reflect":   dict(kind="preconditioned_pcn", N=96,  D=5,  T=3, beta=0.3, nu=4.0, prior="uniform", target="gauss",      seed=3, n_max=4, periodic=[0], reflective=[2]),
    "prwm_n128_d8_uniform":  dict(kind="preconditioned_rwm", N=128, D=8,  T=3, beta=0.6, nu=5.0, prior="uniform", target="rosenbrock", seed=4, n_max=4),
    "pcn_n128_d8_uniform":   dict(kind="pcn",                N=128, D=8,  T=3, beta=0.6, nu=5.0, prior="uniform", target="rosenbrock", seed=5, n_max=4),
    "rwm_n128_d8_normal":    dict(kind="rwm",                N=128, D=8,  T=3, beta=0.9, nu=5.0, prior="normal",  target="gauss",      seed=6, n_max=4),
    # round 2: three more cases per M2 kernel (N >= 256, D in {10, 32}) and BASELINE config 3 (50-D bimodal mixture, maf6)
    "prwm_n256_d10_normal":  dict(kind="preconditioned_rwm", N=256, D=10, T=3, beta=1.0, nu=5.0, prior="normal",  target="gauss",      seed=7, n_max=5),
    "prwm_n320_d32_uniform": dict(kind="preconditioned_rwm", N=320, D=32, T=3, beta=0.5, nu=5.0, prior="uniform", target="corr",       seed=8, n_max=3),
    "prwm_n256_d10_mixed":   dict(kind="preconditioned_rwm", N=256, D=10, T=3, beta=0.4, nu=5.0, prior="mixed",   target="gauss",      seed=9, n_max=4),
    "pcn_n256_d10_normal":   dict(kind="pcn",                N=256, D=10, T=3, beta=1.0, nu=1e6, prior="normal",  target="gauss",      seed=10, n_max=5),
    "pcn_n320_d32_uniform":  dict(kind="pcn",                N=320, D=32, T=3, beta=0.5, nu=5.0, prior="uniform", target="corr",       seed=11, n_max=3),
    "pcn_n256_d10_mixed":    dict(kind="pcn",                N=256, D=10, T=3, beta=0.4, nu=3.5, prior="mixed",   target="gauss",      seed=12, n_max=4),
    "rwm_n256_d10_uniform":  dict(kind="rwm",                N=256, D=10, T=3, beta=0.8, nu=5.0, prior="uniform", target="rosenbrock", seed=13, n_max=5),
    "rwm_n320_d32_uniform":  dict(kind="rwm",                N=320, D=32, T=3, beta=0.5, nu=5.0, prior="uniform", target="corr",       seed=14, n_max=3),
    "rwm_n256_d10_mixed":    dict(kind="rwm",                N=256, D=10, T=3, beta=0.4, nu=5.0, prior="mixed",   target="gauss",      seed=15, n_max=4),
    "tpcn_n256_d50_bimodal": dict(kind="preconditioned_pcn", N=256, D=50, T=6, beta=0.5, nu=5.0, prior="uniform", target="bimodal",    seed=16, n_max=2),
    # round 4: the spline flows (the reference's default is nsf6, sampler.py:169; Flow's own nsf3, flow.py:46) through the
    # reference's kernels -- "flow": the univariate map of MAFSpec ("rqs" = zuko NSF, 8 bins)
    "tpcn_n128_d8_nsf3":     dict(kind="preconditioned_pcn", N=128, D=8,  T=3, beta=0.6, nu=5.0, prior="uniform", target="rosenbrock", seed=17, n_max=4, flow="rqs"),
    "tpcn_n256_d10_nsf6":    dict(kind="preconditioned_pcn", N=256, D=10, T=6, beta=1.0, nu=1e6, prior="normal",  target="gauss",      seed=18, n_max=3, flow="rqs"),
    "prwm_n128_d8_nsf3":     dict(kind="preconditioned_rwm", N=128, D=8,  T=3, beta=0.6, nu=5.0, prior="mixed",   target="gauss",      seed=19, n_max=4, flow="rqs"),
    "tpcn_n512_d32_nsf3":    dict(kind="preconditioned_pcn", N=512, D=32, T=3, beta=0.5, nu=5.0, prior="uniform", target="corr",       seed=20, n_max=2, flow="rqs"),
}

# BASELINE-size cases WITH golden vectors from the reference: a one-step call (n_max = 1: every walker's step depends on
# its own row only, so a row subsample of the reference's output pins the call) -- the reference's per-walker Python
# loops and the oracle's D-pass spline inverse take about a minute here, once, in make_golden.py.  ``rows``: every
# ``stride``-th walker's output is stored.
BIG_GOLDEN_CASES = {
    "tpcn_n10000_d32_nsf3": dict(kind="preconditioned_pcn", N=10000, D=32, T=3, beta=0.5, nu=5.0, prior="uniform", target="rosenbrock", seed=24, n_max=1, flow="rqs", stride=16),
    "tpcn_n10000_d32_maf3": dict(kind="preconditioned_pcn", N=10000, D=32, T=3, beta=0.5, nu=5.0, prior="uniform", target="rosenbrock", seed=25, n_max=1, stride=16),
}


def make_funnel(D):
    """Neal's funnel as a likelihood (SURVEY.md 8(d) cfg 5): x0 ~ N(0, 3^2), x_i ~ N(0, e^{x0})."""
    def f(x):
        x0 = x[:, 0]
        return (-x0 ** 2 / 18.0 - 0.5 * np.sum(x[:, 1:] ** 2, axis=1) * np.exp(-x0) - 0.5 * (D - 1) * x0)
    return f


# BASELINE-size cases that the reference's per-walker Python loops / the D-pass oracle inverse do not finish in
# seconds: no golden vectors, the tests pin them through the oracle with a verified inverse (test_gpu_config.py)
BIG_CASES = {
    # config 5: 128-D funnel, 5000 walkers per GPU, 8-transform MAF, H = 512
    "tpcn_n5000_d128_funnel": dict(kind="preconditioned_pcn", N=5000, D=128, T=8, beta=0.5, nu=5.0, prior="uniform30", target="funnel", seed=21, n_max=2),
    # config 3 at its size: 50-D bimodal mixture, 1e4 walkers, maf6
    "tpcn_n10000_d50_bimodal": dict(kind="preconditioned_pcn", N=10000, D=50, T=6, beta=0.5, nu=5.0, prior="uniform", target="bimodal", seed=22, n_max=2),
    # config 2: 32-D correlated Gaussian, 1e4 walkers, maf3
    "tpcn_n10000_d32_corr": dict(kind="preconditioned_pcn", N=10000, D=32, T=3, beta=0.5, nu=5.0, prior="uniform", target="corr", seed=23, n_max=2),
}


# ---------------------------------------------------- orchestrator cases (pocomc/sampler.py:680-805)
def sampler_pool(seed, T=9, N=128, D=4):
    """A persistent-sampling history: T iterations of N particles (seeded; ``u[:, 0]`` is the pool row index, so the
    rows a method keeps can be read off its output)."""
    rs = np.random.RandomState(seed)
    betas = np.sort(rs.uniform(0, 0.6, T)); betas[0] = 0.0
    logzs = np.cumsum(rs.randn(T)) * 0.5; logzs[0] = 0.0
    rows = dict(u=rs.randn(T, N, D), x=rs.randn(T, N, D), logdetj=rs.randn(T, N), logp=rs.randn(T, N) - 3.0,
                logl=rs.randn(T, N) * 6.0 - 8.0)
    rows["u"][:, :, 0] = np.arange(T * N).reshape(T, N)
    return betas, logzs, rows


SAMPLER_CASES = {
    # name: pool seed, metric, dynamic, n_effective, n_active  (the three branches of sampler.py:747-777)
    "keep_beta_ess":   dict(seed=1, metric="ess", dynamic=False, n_effective=5000, n_active=128),
    "posterior_ess":   dict(seed=2, metric="ess", dynamic=True,  n_effective=8,    n_active=4),
    "bisect_ess":      dict(seed=3, metric="ess", dynamic=False, n_effective=256,  n_active=128),
    "bisect_ess_dyn":  dict(seed=4, metric="ess", dynamic=True,  n_effective=200,  n_active=128),
    "bisect_uss_dyn":  dict(seed=5, metric="uss", dynamic=True,  n_effective=300,  n_active=128),
    "bisect_uss":      dict(seed=6, metric="uss", dynamic=False, n_effective=150,  n_active=64),
    "bisect_ess_dyn2": dict(seed=7, metric="ess", dynamic=True,  n_effective=420,  n_active=128),
    "bisect_ess_dyn3": dict(seed=8, metric="ess", dynamic=True,  n_effective=130,  n_active=128),
}


def find_case(name):
    for table in (MCMC_CASES, BIG_CASES, BIG_GOLDEN_CASES):
        if name in table:
            return table[name]
    raise KeyError(name)


def build_case(name, scaler_cls):
    """Instantiate a case: returns ``(state_dict, function_dict, option_dict, aux)``.

    ``scaler_cls`` is the ``Reparameterize`` class to use (the reference's when
    generating goldens, the oracle's / the product's in tests)."""
    c = find_case(name)
    N, D = c["N"], c["D"]
    rng = np.random.default_rng(1000 + c["seed"])
    prior = {"uniform": lambda: UniformPrior(-10.0, 10.0, D),
             "uniform30": lambda: UniformPrior(-30.0, 30.0, D),
             "normal": lambda: NormalPrior(3.0, D),
             "mixed": lambda: HalfBoundPrior(D)}[c["prior"]]()
    target = {"rosenbrock": rosenbrock, "gauss": std_gauss,
              "corr": make_corr_gauss(D), "bimodal": make_bimodal(D), "funnel": make_funnel(D)}[c["target"]]
    scaler = scaler_cls(D, bounds=prior.bounds, periodic=c.get("periodic"),
                        reflective=c.get("reflective"))
    x_fit = prior.rvs(4 * N, rng)
    scaler.fit(x_fit)
    x = prior.rvs(N, rng)
    u = scaler.forward(x)
    logdetj = scaler.inverse(u)[1]
    logp = prior.logpdf(x)
    logl = target(x)
    spec = MAFSpec(D, c["T"], univariate=c.get("flow", "affine"))
    flat = flow_params(spec, c["seed"])
    # geometry (inputs of the step): a random SPD matrix around the particle scatter
    A = rng.normal(size=(D, D)) * 0.3
    cov = np.eye(D) * 1.3 + A @ A.T
    mean = rng.normal(size=D) * 0.2
    geo = Geo(mean, cov, c["nu"], cov * 0.8)
    state = dict(u=u, x=x, logdetj=logdetj, logl=logl, logp=logp, beta=c["beta"], blobs=None)
    funcs = dict(loglike=lambda xx: (target(xx), None), logprior=prior.logpdf, scaler=scaler,
                 flow=None, theta_geometry=geo, u_geometry=geo)
    opts = dict(n_max=c["n_max"], n_steps=max(D // 2, 1), progress_bar=None,
                proposal_scale=2.38 / D ** 0.5)
    aux = dict(spec=spec, flat=flat, prior=prior, target=target, case=c)
    return state, funcs, opts, aux
