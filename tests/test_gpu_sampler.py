"""The SMC orchestrator on the device kernels: the reference's own smoke tests
(``tests/test_sampler.py:19-44``, ``tests/test_state.py:16-63``) and statistical checks on an
analytic target (the flow is parity-unpinned, so end-to-end agreement is statistical:
SURVEY.md section 8(c))."""
import numpy as np
import pytest
from scipy.stats import norm, uniform

pytestmark = pytest.mark.gpu


def log_likelihood_single(x):
    return -0.5 * np.sum(x ** 2)


def log_likelihood_vectorized(x):
    return -0.5 * np.sum(x ** 2, axis=1)


def test_run_scalar_likelihood():
    """tests/test_sampler.py:19-30."""
    import pocomc_amd as pc
    prior = pc.Prior([norm(0, 1), norm(0, 1)])
    s = pc.Sampler(prior=prior, likelihood=log_likelihood_single, train_config={"epochs": 1}, random_state=0)
    s.run(progress=False)
    assert np.isfinite(s.evidence()[0])


def test_run_vectorized_likelihood():
    """tests/test_sampler.py:32-44."""
    import pocomc_amd as pc
    prior = pc.Prior([norm(0, 1), norm(0, 1)])
    s = pc.Sampler(prior=prior, likelihood=log_likelihood_vectorized, vectorize=True, train_config={"epochs": 1},
                   random_state=0)
    s.run(progress=False)
    x, w, logl, logp = s.posterior()
    assert x.shape[1] == 2 and len(w) == len(x) and np.isfinite(logl).all()


def test_save_load_resume(tmp_path):
    """tests/test_state.py:16-63."""
    import pocomc_amd as pc
    prior = pc.Prior([norm(0, 1), norm(0, 1)])
    s = pc.Sampler(prior=prior, likelihood=log_likelihood_vectorized, vectorize=True, train_config={"epochs": 1},
                   random_state=0, output_dir=tmp_path, output_label="t")
    s.run(progress=False, save_every=1, n_total=512, n_evidence=0)
    files = sorted(tmp_path.glob("t_*.state"))
    assert (tmp_path / "t_final.state").exists() and len(files) >= 2
    s2 = pc.Sampler(prior=prior, likelihood=log_likelihood_vectorized, vectorize=True, train_config={"epochs": 1},
                    random_state=0)
    s2.load_state(tmp_path / "t_final.state")
    assert s2.t == s.t and np.allclose(s2.flow.params.cpu().numpy(), s.flow.params.cpu().numpy())
    first = [f for f in files if "final" not in f.name][0]
    s3 = pc.Sampler(prior=prior, likelihood=log_likelihood_vectorized, vectorize=True, train_config={"epochs": 1},
                    random_state=0)
    s3.run(progress=False, resume_state_path=first, n_total=512, n_evidence=0)
    assert np.isfinite(s3.evidence()[0])


@pytest.mark.parametrize("sample,precondition,mcmc_options",
                         [("tpcn", True, None), ("rwm", True, None), ("tpcn", False, None),
                          ("tpcn", True, dict(x_order="F")), ("rwm", False, dict(x_order="F", lanes=2))])
def test_gaussian_posterior_and_evidence(sample, precondition, mcmc_options):
    """6-D Gaussian likelihood N(mu, 0.5^2) inside U(-5,5)^6: logZ = -6 log 10 analytically
    (likelihood normalised), posterior mean mu, std 0.5."""
    import pocomc_amd as pc
    D, mu, sd = 6, 0.7, 0.5
    prior = pc.Prior([uniform(-5, 10)] * D)

    def loglike(x):
        return np.sum(-0.5 * ((x - mu) / sd) ** 2 - np.log(sd) - 0.5 * np.log(2 * np.pi), axis=1)
    s = pc.Sampler(prior=prior, likelihood=loglike, vectorize=True, flow="maf3", sample=sample,
                   precondition=precondition, random_state=1, n_effective=512, n_active=256,
                   train_config={"epochs": 200}, mcmc_options=mcmc_options)   # x_order 'F': pipelined kernel calls
    s.run(progress=False, n_total=2048, n_evidence=2048 if precondition else 0)
    x, w, logl, logp = s.posterior()
    m = np.average(x, axis=0, weights=w)
    sdev = np.sqrt(np.average((x - m) ** 2, axis=0, weights=w))
    assert np.abs(m - mu).max() < 0.08, m
    assert np.abs(sdev - sd).max() < 0.08, sdev
    logz = s.evidence()[0]
    assert abs(logz - (-D * np.log(10.0))) < 0.25, logz


def test_default_flow_is_the_reference_default():
    """sampler.py:169: the Sampler's default flow is the spline flow nsf6; a short run on a
    correlated Gaussian recovers the analytic evidence with it."""
    import pocomc_amd as pc
    from scipy.stats import norm
    D = 4
    rng = np.random.default_rng(0)
    A = rng.normal(size=(D, D))
    cov = A @ A.T / D + 0.5 * np.eye(D)
    icov = np.linalg.inv(cov)
    norm_const = -0.5 * (D * np.log(2 * np.pi) + np.linalg.slogdet(cov)[1])

    def loglike(x):
        return norm_const - 0.5 * np.einsum("ni,ij,nj->n", x, icov, x)

    prior = pc.Prior([norm(0.0, 5.0)] * D)
    s = pc.Sampler(prior=prior, likelihood=loglike, vectorize=True, random_state=3)
    assert s.flow.spec.univariate == "rqs" and s.flow.spec.n_transforms == 6
    s.run(n_total=2048, n_evidence=2048, progress=False)
    logz, logz_err = s.evidence()
    # evidence of N(0, cov) likelihood under N(0, 25 I) prior: N(0; 0, cov + 25 I)
    true_logz = -0.5 * (D * np.log(2 * np.pi) + np.linalg.slogdet(cov + 25.0 * np.eye(D))[1])
    assert abs(logz - true_logz) < 0.35, (logz, true_logz)
    x, w, _, _ = s.posterior()
    m = np.average(x, weights=w, axis=0)
    assert np.abs(m).max() < 0.3


@pytest.mark.parametrize("flow_kind", ["maf3", "bf16"])
def test_a_run_is_reproducible_from_its_random_state(flow_kind):
    """``random_state`` fixes everything (``sampler.py:319-322``): numpy / torch streams on the host, Philox keys drawn
    from them on the device, fixed summation orders in every kernel (no floating-point atomics) -- two runs give the same
    particles and the same trained flow bit for bit.  The sharded Sampler relies on exactly this (every rank replicates
    the pool bookkeeping).  ``bf16``: a ready ``Flow`` on the bf16 matrix cores handed to the Sampler, trained by the
    bf16 engine."""
    import hashlib
    import pocomc_amd as pc
    from pocomc_amd.maf_spec import MAFSpec
    D = 6
    prior = pc.Prior([uniform(-5, 10)] * D)

    def loglike(x):
        return np.sum(-0.5 * ((x - 0.7) / 0.5) ** 2, axis=1)

    digests = []
    for _ in range(2):
        if flow_kind == "bf16":
            flow = pc.Flow(D, MAFSpec(D, 3, hidden=64), precision="bf16", seed=2)
            flow.train_engine = "bf16"
        else:
            flow = flow_kind
        s = pc.Sampler(prior=prior, likelihood=loglike, vectorize=True, flow=flow, random_state=5, n_effective=512,
                       n_active=256, train_config={"epochs": 30}, mcmc_options=dict(x_order="F", lanes=2))
        s.run(progress=False)
        x, w, _, _ = s.posterior()
        digests.append((hashlib.sha1(np.ascontiguousarray(x).tobytes()).hexdigest(),
                        hashlib.sha1(s.flow.params.cpu().numpy().tobytes()).hexdigest(), s.evidence()[0]))
        assert abs(s.evidence()[0] - (D * np.log(0.5 * np.sqrt(2 * np.pi)) - D * np.log(10.0))) < 0.3
    assert digests[0] == digests[1]
