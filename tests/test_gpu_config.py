"""BASELINE.json's configurations at their own sizes (the numpy oracle's D-pass inverse does not finish in seconds
there; see ``test_gpu_mcmc.VerifiedInverse``) and config 5's flow (D=128, 8 transforms, H=512) against the oracle.

config 1: 10-D Rosenbrock, 1000 particles, uniform prior       -> Sampler end to end (logZ over 3 seeds)
config 2: 32-D correlated Gaussian, 1e4 particles, maf3         -> teacher-forced tpCN step at 1e4 x 32
config 3: 50-D bimodal mixture, 1e4 particles, maf6             -> teacher-forced tpCN step at 1e4 x 50
config 4: 32-D Rosenbrock, 1e4 per GPU                          -> bench.py (+ test_gpu_fullsize, test_sharded_cpu)
config 5: 128-D funnel, 5000 per GPU, 8-transform MAF, H=512    -> flow parity + teacher-forced step at 5000 x 128
"""
import numpy as np
import pytest
import torch

import cases
from oracle.maf import OracleMAF
from parity import TOL, close_rel
from pocomc_amd.maf_spec import MAFSpec
from test_gpu_mcmc import teacher_forced

pytestmark = pytest.mark.gpu


def make(D, T, seed=3):
    from pocomc_amd import Flow
    spec = MAFSpec(D, T)
    flat = cases.flow_params(spec, seed)
    f = Flow(D, spec)
    f.set_params(flat)
    return f, OracleMAF(spec, flat)


@pytest.mark.parametrize("n", [1, 33, 500])
def test_config5_flow_forward_logprob_inverse_match_oracle(n):
    """(D, T, H) = (128, 8, 512): forward / log_prob / inverse (every algorithm the library offers for this width)
    against the oracle, 1e-5 relative per row."""
    f, o = make(128, 8)
    assert f.spec.hidden == 512
    rng = np.random.default_rng(n)
    x = (rng.normal(size=(n, 128)) * 1.5).astype(np.float32)
    z, ladj = f.forward(torch.from_numpy(x))
    zo, lo = o.forward(x)
    close_rel(z.numpy(), zo, TOL, "forward z")
    terms = o.ladj_abs_terms(x)                 # size of the terms the log-determinant sums (its conditioning)
    close_rel(ladj.numpy(), lo, TOL, "forward ladj", cancel=terms)
    close_rel(f.log_prob(torch.from_numpy(x)).numpy(), o.log_prob(x), TOL, "log_prob",
              cancel=terms + 0.5 * (zo.astype(np.float64) ** 2).sum(axis=1))
    zi = (rng.normal(size=(n, 128)) * 1.2).astype(np.float32)
    xo, lio = o.inverse(zi)                      # zuko's D-pass algorithm
    for algo in (0, 1, 2):                       # AUTO, triangular sweep, D-pass on the device
        f.inverse_algo = algo
        xi, li = f.inverse(torch.from_numpy(zi))
        close_rel(xi.numpy(), xo, TOL, f"inverse x (algo {algo})")
        close_rel(li.numpy(), lio, TOL, f"inverse ladj (algo {algo})", cancel=o.ladj_abs_terms(xo))
    f.inverse_algo = 0


@pytest.mark.parametrize("name", list(cases.BIG_CASES))
def test_teacher_forced_step_at_baseline_size(name):
    """One tpCN kernel call of BASELINE configs 2, 3 and 5 at their per-GPU size, step by step against the oracle
    (replayed variates; the flow inverse verified through the oracle's forward map)."""
    teacher_forced(name, verified_inverse=True)


def test_config1_sampler_rosenbrock_10d_1000_particles():
    """BASELINE configs[0] (README.md:43-66): 10-D Rosenbrock, U(-10,10)^10, n_active = 1000 through the Sampler;
    logZ over 3 seeds against the value committed from scripts/run_readme_example.py (same model, reference
    defaults) -- the estimator's own spread is ~0.1."""
    from scipy.stats import uniform
    import pocomc_amd as pc
    D = 10
    prior = pc.Prior(D * [uniform(-10.0, 20.0)])
    logzs = []
    for seed in (0, 1, 2):
        s = pc.Sampler(prior=prior, likelihood=cases.rosenbrock, vectorize=True, n_effective=2000, n_active=1000,
                       flow="maf3", random_state=seed)
        s.run()
        samples, weights, logl, logp = s.posterior()
        logz, logz_err = s.evidence()
        assert np.isfinite(samples).all() and samples.shape[1] == D
        assert abs(weights.sum() - 1) < 1e-9
        # posterior sits in the Rosenbrock valley: x_{2i} ~ 1 on average is too strong a statement; the pairs obey
        # x_{2i+1} ~ x_{2i}^2 within the likelihood's width
        w = weights / weights.sum()
        resid = np.sum(w[:, None] * (samples[:, ::2] ** 2 - samples[:, 1::2]) ** 2, axis=0)
        assert (resid < 1.0).all(), resid
        logzs.append(logz)
    logzs = np.array(logzs)
    print("config 1 logZ over seeds:", logzs)
    # the prior volume is 20^10: logZ = log int L dx - 10 log 20; int exp(-10 (x^2-y)^2 - (x-1)^2) dx dy = pi/sqrt(10) per pair
    exact = 5 * np.log(np.pi / np.sqrt(10.0)) - 10 * np.log(20.0)
    assert np.abs(logzs - exact).max() < 0.75, (logzs, exact)
    assert logzs.std() < 0.5
