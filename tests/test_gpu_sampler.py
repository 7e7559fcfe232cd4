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


@pytest.mark.parametrize("prec", ["f16", "bf16"])
def test_sampler_runs_on_a_guarded_16bit_flow(prec):
    """A whole SMC run on a flow wide enough for the lane sweep (16 hidden tiles) with ``inverse_precision`` opted in: every
    ``Flow.fit`` ends with the guard, every kernel call starts with it on the walkers' own theta, and whichever way the
    verdicts go (16-bit kept, or float32 with a warning) the run recovers the analytic evidence like the float32 run."""
    import warnings
    import pocomc_amd as pc
    from pocomc_amd.maf_spec import MAFSpec
    D = 20                                   # (hidden = 256 over 19 degrees: groups of <= 16 units, 16 hidden tiles: the lane sweep)
    prior = pc.Prior([uniform(-5, 10)] * D)

    def loglike(x):
        return np.sum(-0.5 * ((x - 0.7) / 0.5) ** 2, axis=1)
    flow = pc.Flow(D, MAFSpec(D, 3, hidden=256), seed=2, inverse_precision=prec)
    import ctypes
    assert flow._lane16 is not None and flow.inverse_precision == prec
    assert flow.lib.pmc_maf_inverse_auto_is_lane(ctypes.byref(flow._desc)) == 1
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        s = pc.Sampler(prior=prior, likelihood=loglike, vectorize=True, flow=flow, random_state=5, n_effective=1024,
                       n_active=512, train_config={"epochs": 40}, mcmc_options=dict(x_order="F"))
        s.run(progress=False)
    g = s.flow.inverse_guard
    fell = [str(m.message) for m in w if "falling back to the float32 sweep" in str(m.message)]
    print(f"{prec}: last guard {g}; fallbacks during the run: {len(fell)}; active now: {s.flow.inverse_precision_active}")
    assert g is not None and g["rows"] >= 256 and g["precision"] == prec
    assert (s.flow.inverse_precision_active == prec) == g["passed"]
    assert abs(s.evidence()[0] - (D * np.log(0.5 * np.sqrt(2 * np.pi)) - D * np.log(10.0))) < 0.5


@pytest.mark.parametrize("flow_name,n", [("maf3", 700), ("maf6", 1500)])
def test_compute_evidence_replayed_against_the_oracle(flow_name, n):
    """``Sampler._compute_evidence`` (``sampler.py:869-920``: flow.sample -> scaler.inverse -> prior -> likelihood ->
    importance-sampling logZ -> bootstrap spread) against ``oracle/evidence.py`` on the SAME base draw of the flow and
    the SAME bootstrap indices (the reference's ``np.random.choice`` calls under ``np.random.seed``, replayed through
    ``pmc_bootstrap_logz_replay``).  Half-bounded prior: part of the draws falls outside its support and is dropped
    (``:898-901``).  logZ to 1e-5 relative on its terms; the bootstrap's standard deviation -- a difference of
    replicates that agree to 1e-6 -- to 1e-3 relative."""
    import pocomc_amd as pc
    from oracle.evidence import compute_evidence
    from oracle.maf import OracleMAF
    from oracle.scaler import Reparameterize as OracleScaler
    from scipy.stats import halfnorm
    D = 5
    prior = pc.Prior([uniform(-4, 8)] * (D - 1) + [halfnorm(0, 2)])

    def loglike(x):
        return np.sum(-0.5 * ((x - 0.3) / 0.8) ** 2, axis=1) - 0.1 * x[:, 0] ** 4
    s = pc.Sampler(prior=prior, likelihood=loglike, vectorize=True, flow=flow_name, random_state=3, n_effective=256,
                   n_active=128, train_config={"epochs": 30})
    s.run(progress=False, n_total=512, n_evidence=0)
    rng = np.random.default_rng(n)
    z = rng.normal(size=(n, D)).astype(np.float32)
    osc = OracleScaler(D, bounds=prior.bounds)
    osc.mu, osc.sigma = np.array(s.scaler.mu, dtype=np.float64), np.array(s.scaler.sigma, dtype=np.float64)
    if hasattr(osc, "_refresh"):
        osc._refresh()
    maf = OracleMAF(s.flow.spec, s.flow.params.cpu().numpy())
    logz_o, dlogz_o, draws, logw_o = compute_evidence(maf, osc, prior.logpdf, loglike, z, seed=17, n_boot=300)
    assert 0.5 * n < len(logw_o) <= n
    calls0 = s.calls
    logz, dlogz = s._compute_evidence(replay=dict(z=z, draws=draws))
    assert s.calls - calls0 == len(logw_o)                     # the same rows survived the prior
    terms = np.abs(logw_o).max()
    assert abs(logz - logz_o) <= 1e-5 * max(abs(logz_o), terms), (logz, logz_o)
    assert abs(dlogz - dlogz_o) <= 1e-3 * dlogz_o, (dlogz, dlogz_o)


# ------------------------------------------------------------ orchestrator arithmetic vs the reference's own sampler.py
def _sampler_on_pool(c, N, D):
    import pocomc_amd as pc
    from pocomc_amd.particles import Particles
    from pocomc_amd.sampler import _Progress
    prior = pc.Prior([norm(0, 1)] * D)
    s = pc.Sampler(prior=prior, likelihood=log_likelihood_vectorized, vectorize=True, flow="maf3", random_state=0,
                   n_effective=c["n_effective"], n_active=c["n_active"], metric=c["metric"], dynamic=c["dynamic"])
    s.particles = Particles(N, D)
    s.pbar, s.walkers = _Progress(False), {}
    return s


@pytest.mark.parametrize("name", list(__import__("cases").SAMPLER_CASES))
def test_temper_and_draw_match_the_reference_sampler(name, golden_dir):
    """``Sampler._temper`` / ``_draw`` on a resident pool against ``Sampler._reweight`` (``sampler.py:717-805``) and
    ``_resample`` (``:680-715``) of the reference itself (called unbound by ``tests/golden/make_golden.py`` on the same
    history): the three branches of the temperature ladder, ESS and USS metrics, the dynamic ESS adjustment in both
    directions, trimming, both resampling schemes on numpy's legacy stream.  beta, the dynamic n_effective, the kept
    rows and the resampled rows are exact; weights / logZ / ESS to 1e-12 (device reductions sum in another order)."""
    import cases
    g = np.load(f"{golden_dir}/sampler_reference.npz")
    c = cases.SAMPLER_CASES[name]
    betas, logzs, rows = cases.sampler_pool(c["seed"])
    nT, N, D = rows["u"].shape
    s = _sampler_on_pool(c, N, D)
    for t in range(nT):
        s.particles.update(dict(u=rows["u"][t], x=rows["x"][t], logdetj=rows["logdetj"][t], logp=rows["logp"][t],
                                logl=rows["logl"][t], beta=betas[t], logz=logzs[t], iter=t, calls=0, steps=1,
                                efficiency=1.0, ess=1.0, accept=1.0))
    s.t = nT
    idx, w = s._temper()
    tag = f"sampler/{name}"
    assert s.walkers["beta"] == float(g[f"{tag}/beta"])
    np.testing.assert_allclose(s.walkers["logz"], g[f"{tag}/logz"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(s.walkers["ess"], g[f"{tag}/ess"], rtol=1e-11)
    assert s.n_effective == int(g[f"{tag}/n_effective_after"])
    np.testing.assert_array_equal(idx.cpu().numpy(), g[f"{tag}/idx"])
    np.testing.assert_allclose(w.cpu().numpy(), g[f"{tag}/weights"], rtol=1e-12)
    kept = s.particles.take(idx)
    for k in ("x", "logdetj", "logl", "logp"):
        np.testing.assert_array_equal(kept[k].cpu().numpy(), g[f"{tag}/kept_{k}"])
    for scheme in ("mult", "syst"):
        s.resample = scheme
        np.random.seed(100 + c["seed"])
        s._draw((idx, w))
        np.testing.assert_array_equal(s.walkers["u"][:, 0].cpu().numpy().astype(np.int64), g[f"{tag}/{scheme}/rows"])
        for k in ("x", "logdetj", "logl", "logp"):
            np.testing.assert_array_equal(s.walkers[k].cpu().numpy(), g[f"{tag}/{scheme}/{k}"])


@pytest.mark.parametrize("name", ["maf3_d5", "nsf3_d4"])
def test_compute_evidence_matches_the_reference_sampler(name, golden_dir):
    """``Sampler._compute_evidence`` against the reference's own method (``sampler.py:869-920`` called unbound by
    ``make_golden.py`` with the oracle flow behind ``.sample``): same base draw, the bootstrap indices of the same
    legacy-stream seed replayed through ``pmc_bootstrap_logz_replay``; a prior whose support cuts inside the scaler's
    bounds (rows dropped, ``:898-901``).  logZ to the flow's tolerance on its terms, the bootstrap spread to 1e-3."""
    import cases
    import pocomc_amd as pc
    from pocomc_amd.maf_spec import MAFSpec
    from pocomc_amd.sampler import _Progress
    g = np.load(f"{golden_dir}/sampler_reference.npz")
    tag = f"evidence/{name}"
    Dn, Tn, rqs, n, seed = (int(v) for v in g[f"{tag}/spec"])
    spec = MAFSpec(Dn, Tn, univariate="rqs" if rqs else "affine")
    flow = pc.Flow(Dn, spec)
    flow.set_params(cases.flow_params(spec, seed, gain=1.0))
    prior = cases.CutPrior(Dn)

    class P:                                   # the Sampler's view of a prior (prior.py): logpdf, rvs, bounds, dim
        logpdf, bounds, dim = staticmethod(prior.logpdf), prior.bounds, Dn
        rvs = staticmethod(lambda m: prior.rvs(m, np.random.default_rng(0)))
    like = lambda x: -0.5 * np.sum(((x - 0.3) / 0.8) ** 2, axis=1) - 0.1 * x[:, 0] ** 4
    s = pc.Sampler(prior=P, likelihood=like, vectorize=True, flow=flow, random_state=0, n_effective=256, n_active=128)
    s.scaler.fit(g[f"{tag}/x_fit"])
    s.pbar = _Progress(False)
    m = int(g[f"{tag}/calls"])
    np.random.seed(seed)
    draws = np.stack([np.random.choice(m, m) for _ in range(max(n, 1000))])
    logz, dlogz = s._compute_evidence(replay=dict(z=g[f"{tag}/z"], draws=draws))
    assert s.calls == m < n
    tol = 5e-5 if rqs else 1e-5
    assert abs(logz - float(g[f"{tag}/logz"])) <= tol * max(abs(float(g[f"{tag}/logz"])), 30.0), (logz, g[f"{tag}/logz"])
    assert abs(dlogz - float(g[f"{tag}/dlogz"])) <= 1e-3 * float(g[f"{tag}/dlogz"]), (dlogz, g[f"{tag}/dlogz"])
