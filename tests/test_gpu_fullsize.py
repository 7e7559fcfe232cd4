"""BASELINE.json's full sizes (1e4 walkers x 32-D maf3; 1e4 x 50-D maf6; 8e4-particle pools) are beyond what the
numpy oracle finishes in seconds, so parity at these sizes is asserted through size-independent properties:
round trips, algorithm-vs-algorithm agreement on the device, conservation of what a Metropolis step may
change, checksums of the reductions, invariance under sharding, linearity of the gradient."""
import ctypes as C

import numpy as np
import pytest
import torch

import cases

pytestmark = pytest.mark.gpu

N_FULL = 10000


def _flow(D, name, seed=0, epochs=0):
    import pocomc_amd as pc
    f = pc.Flow(D, name, seed=seed)
    # a non-trivial flow: scale the default init
    f.set_params(f.params.cpu().numpy() * np.float32(1.15))
    return f


@pytest.mark.parametrize("D,name", [(32, "maf3"), (50, "maf6"), (32, "nsf3")])
def test_flow_round_trip_and_ladj_antisymmetry_at_full_size(D, name):
    """configs 2-4: forward(inverse(z)) = z and ladj_inverse = -ladj_forward (tests/test_flow.py:88,:164) on 1e4 rows."""
    f = _flow(D, name)
    z = torch.randn(N_FULL, D, generator=torch.Generator().manual_seed(1)) * 1.3
    x, li = f.inverse(z)
    z2, lf = f.forward(x)
    assert torch.isfinite(x).all() and torch.isfinite(li).all()
    err = (z2 - z).abs().max().item()
    assert err < 5e-4, err
    assert (lf + li).abs().max().item() < 5e-3
    # log_prob is consistent with forward: base density of z2 plus the forward log-determinant
    lp = f.log_prob(x)
    ref = -0.5 * (z2 ** 2).sum(1) - 0.5 * D * np.log(2 * np.pi) + lf
    assert (lp - ref).abs().max().item() < 1e-3


@pytest.mark.parametrize("D,name", [(32, "maf3"), (32, "nsf3")])
def test_triangular_sweep_equals_the_reference_algorithm_at_full_size(D, name):
    """The single-sweep inverse against zuko's D-pass algorithm, both on the device, 1e4 rows."""
    f = _flow(D, name)
    z = torch.randn(N_FULL, D, generator=torch.Generator().manual_seed(2)) * 1.2
    f.inverse_algo = 1
    xa, la = f.inverse(z)
    f.inverse_algo = 2
    xb, lb = f.inverse(z)
    assert (xa - xb).abs().max().item() < 2e-4 * max(1.0, xb.abs().max().item())
    assert (la - lb).abs().max().item() < 2e-3


def _engine(N, D, seed=77, offset=0, flow=None, like=None):
    from scipy.stats import uniform
    import pocomc_amd as pc
    from pocomc_amd.geometry import Geometry
    from pocomc_amd.mcmc import StepEngine
    prior = pc.Prior([uniform(-10, 20)] * D)
    rng = np.random.default_rng(3)
    scaler = pc.Reparameterize(D, bounds=prior.bounds)
    scaler.fit(rng.uniform(-10, 10, size=(4000, D)))
    x = rng.uniform(-9, 9, size=(N_FULL, D))
    u = scaler.forward(x)
    flow = flow or _flow(D, "maf3")
    like = like or (lambda xx: (-0.5 * np.sum((xx / 3.0) ** 2, axis=1), None))
    geo = Geometry()
    geo.fit(flow.forward(torch.from_numpy(u).float())[0].numpy().astype(np.float64))
    sl = slice(offset, offset + N)
    eng = StepEngine("preconditioned_pcn", N, D, flow, scaler, seed=seed, shard_offset=offset, x_order="F")
    eng.set_device_prior(prior)
    eng.load_state(u[sl], x[sl], scaler.inverse(u[sl])[1], like(x[sl])[0], prior.logpdf(x[sl]))
    eng.set_geometry(mu=geo.t_mean, cov=geo.t_cov)
    eng.scaler = scaler
    return eng, prior, like, geo


@pytest.mark.parametrize("D,name,target", [(32, "maf3", "gauss"), (50, "maf6", "bimodal")])
def test_metropolis_step_conserves_and_checksums_at_full_size(D, name, target):
    """One tpCN step on 1e4 x 32 (config 4's per-GPU shard) and on 1e4 x 50 with maf6 and the bimodal mixture of
    BASELINE configs[2]: every walker is afterwards either exactly its old self or exactly its proposal, consistently
    across all state arrays; the sums the kernel reduces (what the adaptation consumes, mcmc.py:152-177) equal the sums
    of the downloaded state."""
    like = None
    if target == "bimodal":
        bm = cases.make_bimodal(D)
        like = lambda xx: (bm(xx), None)
    eng, prior, like, geo = _engine(N_FULL, D, flow=_flow(D, name), like=like)
    before = eng.download()
    theta_before = eng.theta32.cpu().numpy().copy()
    eng.propose(0.35, float(geo.t_nu))
    calls, _ = eng.evaluate(prior.logpdf, like)
    prop = dict(u=eng.p_u.cpu().numpy(), x=eng.p_x.cpu().numpy(), logdetj=eng.p_logdetj.cpu().numpy(),
                theta=eng.p_theta64.cpu().numpy())
    logl_p = eng._np_logl.copy()
    sums = eng.accept_reduce(0.5, float(geo.t_nu), want_mask=True).copy()
    acc = eng.h_accept.numpy().astype(bool)
    after = eng.download()
    assert calls == N_FULL and 0.02 < acc.mean() < 0.98
    for k in ("u", "x"):
        assert np.array_equal(after[k][acc], prop[k][acc]) and np.array_equal(after[k][~acc], before[k][~acc])
    assert np.array_equal(after["logdetj"][acc], prop["logdetj"][acc])
    assert np.array_equal(after["logl"][acc], logl_p[acc]) and np.array_equal(after["logl"][~acc], before["logl"][~acc])
    theta_after = eng.theta32.cpu().numpy()
    assert np.array_equal(theta_after[acc], prop["theta"][acc].astype(np.float32))
    assert np.array_equal(theta_after[~acc], theta_before[~acc])
    # checksums: [sum alpha, sum(logl+logp), sum(logl+logp+logdetj), n accepted, sum_k theta_k]
    assert sums[3] == acc.sum()
    np.testing.assert_allclose(sums[1], np.sum(after["logl"] + after["logp"]), rtol=1e-12)
    np.testing.assert_allclose(sums[2], np.sum(after["logl"] + after["logp"] + after["logdetj"]), rtol=1e-12)
    np.testing.assert_allclose(sums[4:], theta_after.astype(np.float64).sum(axis=0), rtol=1e-9, atol=1e-9)
    assert 0.0 < sums[0] <= N_FULL


def test_proposals_do_not_depend_on_the_sharding():
    """Philox variates are keyed by the GLOBAL walker index: the 1e4 walkers proposed by one engine or by two engines
    of 5e3 (two GPUs) are bit-identical."""
    D = 32
    flow = _flow(D, "maf3")
    whole, *_ , geo = _engine(N_FULL, D, flow=flow)
    whole.propose(0.35, float(geo.t_nu))
    torch.cuda.synchronize()
    ref_theta, ref_x = whole.p_theta64.cpu().numpy(), whole.p_x.cpu().numpy()
    for off in (0, N_FULL // 2):
        part, *_ = _engine(N_FULL // 2, D, offset=off, flow=flow)
        part.propose(0.35, float(geo.t_nu))
        torch.cuda.synchronize()
        assert np.array_equal(part.p_theta64.cpu().numpy(), ref_theta[off:off + N_FULL // 2])
        assert np.array_equal(part.p_x.cpu().numpy(), ref_x[off:off + N_FULL // 2])


def test_pipelined_two_lane_steps_equal_whole_set_steps_at_full_size():
    """bench.py's timed region -- 1e4 x 32 walkers stepped as two row ranges, sigma / mu adapted on the device, the
    next pre-steps enqueued behind the accepts -- against whole-set steps adapted on the host: the same Philox
    variates and (bit-identical) sweep kernels, so after 4 steps the same walkers moved to the same places; the
    sums, sigma and mu agree to the order of the additions."""
    from pocomc_amd.mcmc import Adaptation, LanedEngine
    D = 32
    flow = _flow(D, "maf3")
    whole, prior, like, geo = _engine(N_FULL, D, flow=flow)
    nu, beta, sigma0 = float(geo.t_nu), 0.5, 2.38 / D ** 0.5
    mk = lambda: Adaptation("preconditioned_pcn", D, N_FULL, n_steps=10 ** 9, n_max=10 ** 9, sigma0=sigma0,
                            mu0=geo.t_mean, logp2_0=-np.inf)
    start = whole.download()
    ad_w = mk()
    sums_w = []
    for _ in range(4):
        whole.propose(ad_w.sigma, nu)
        whole.evaluate(prior.logpdf, like)
        sums_w.append(whole.accept_reduce(beta, nu).copy())
        ad_w.update(sums_w[-1])
        whole.set_mu(ad_w.mu)
    end_w = whole.download()

    lanes = LanedEngine("preconditioned_pcn", N_FULL, D, flow, whole.scaler, lanes=2, seed=77, x_order="F",
                        streams=False)
    assert lanes.set_device_prior(prior) and lanes.can_pipeline()
    assert [e.n for e in lanes.lanes] == [5008, 4992]
    lanes.load_state(start["u"], start["x"], start["logdetj"], start["logl"], start["logp"])
    # (the whole-set engine started from the same arrays: _engine loads them before any step)
    lanes.set_geometry(mu=geo.t_mean, cov=geo.t_cov)
    ad_l = mk()
    lanes.start_pipeline(float(ad_l.sigma), ad_l.mu, nu)
    for k in range(4):
        calls, sums = lanes.step_pipelined(beta, nu, ad_l.coefficients(), N_FULL, prior.logpdf, like, more=k < 3)
        assert calls == N_FULL
        np.testing.assert_allclose(sums, sums_w[k], rtol=1e-9, atol=1e-9)
        ad_l.update(sums)
    lanes.finish_pipeline()
    np.testing.assert_allclose(ad_l.sigma, ad_w.sigma, rtol=1e-12)
    np.testing.assert_allclose(ad_l.mu, ad_w.mu, rtol=1e-12, atol=1e-14)
    end_l = lanes.download()
    same = np.isclose(end_l["u"], end_w["u"], rtol=1e-9, atol=1e-12).all(axis=1)
    assert same.mean() > 0.999, same.mean()
    moved = ~np.isclose(end_w["u"], start["u"]).all(axis=1)
    assert 0.3 < moved.mean() < 1.0


@pytest.mark.parametrize("name", ["maf3", "nsf3"])
def test_the_host_never_reads_behind_the_completion_word_at_full_size(name):
    """The pipelined kernel call hands x', the finite mask and logp' to the host through pinned memory behind completion
    words (every workgroup waits for the acknowledgement of its stores -- agent scope --, the last one publishes the word
    with a system-scope release: ``scaler_body.h``).  120 steps of 1e4 x 32 walkers as two row ranges against the same call
    with ``host_direct=False`` (stream-ordered copies, runtime synchronisation): one stale word of x' read by the likelihood
    changes a log-likelihood, then an accept decision, then everything after it -- the results are the same bit for bit."""
    from scipy.stats import uniform
    import pocomc_amd as pc
    from pocomc_amd import mcmc as pmcmc
    from pocomc_amd.geometry import Geometry
    D, N = 32, N_FULL
    prior = pc.Prior([uniform(-10, 20)] * D)
    rng = np.random.default_rng(5)
    scaler = pc.Reparameterize(D, bounds=prior.bounds)
    scaler.fit(rng.uniform(-10, 10, size=(4000, D)))
    x = rng.uniform(-9, 9, size=(N, D))
    u = scaler.forward(x)
    wts = np.linspace(0.5, 1.5, D)
    like = lambda xx: (-0.5 * np.sum(wts * (xx / 3.0) ** 2, axis=1), None)     # (every coordinate of every walker counts)
    flow = _flow(D, name)
    geo = Geometry()
    geo.fit(flow.forward(torch.from_numpy(u).float())[0].numpy().astype(np.float64))
    res = []
    for direct in (True, False):
        state = dict(u=u.copy(), x=x.copy(), logdetj=scaler.inverse(u)[1], logl=like(x)[0], logp=prior.logpdf(x),
                     beta=0.5, blobs=None)
        funcs = dict(loglike=like, logprior=prior.logpdf, scaler=scaler, flow=flow, theta_geometry=geo)
        opts = dict(n_max=120, n_steps=10 ** 6, progress_bar=None, proposal_scale=2.38 / D ** 0.5, seed=7,
                    device_prior=True, x_order="F", host_direct=direct, lanes=2)
        res.append(pmcmc.preconditioned_pcn(state, funcs, opts))
    assert res[0]["steps"] == res[1]["steps"] == 120 and res[0]["accept"] == res[1]["accept"]
    assert res[0]["calls"] == res[1]["calls"] == 120 * N
    for k in ("u", "x", "logl", "logp", "logdetj"):
        assert np.array_equal(res[0][k], res[1][k]), k


def test_pool_statistics_trim_and_resample_at_full_size():
    """An 8e4-particle persistent pool (config 4, 8 iterations x 1e4): the mixture log-weights normalise, ESS agrees
    with its definition, trimming keeps >= 99 % of the ESS with weights that sum to one, systematic resampling
    returns sorted indices whose multiplicities match n*w within one."""
    from pocomc_amd import tools
    T, N = 8, N_FULL
    rng = np.random.default_rng(6)
    logl = rng.normal(size=(T, N)) * 3.0 - 5.0
    beta = np.linspace(0.0, 0.7, T)
    logz = np.cumsum(rng.normal(size=T) * 0.1)
    logw, logz_new = tools.compute_logw_and_logz(logl, beta, logz, beta_final=0.8, normalize=True)
    np.testing.assert_allclose(np.exp(logw).sum(), 1.0, rtol=1e-10)
    w = np.exp(logw)
    np.testing.assert_allclose(tools.effective_sample_size(w.copy()), 1.0 / np.sum(w ** 2), rtol=1e-10)
    idx, wt = tools.trim_weights(np.arange(T * N), w.copy(), ess=0.99, bins=1000)
    np.testing.assert_allclose(wt.sum(), 1.0, rtol=1e-12)
    assert 1.0 / np.sum(wt ** 2) >= 0.99 * (1.0 / np.sum(w ** 2)) * (1 - 1e-9)
    assert len(idx) <= T * N and np.all(np.diff(idx) > 0)
    ridx = tools.systematic_resample(N, wt, offset=0.37)
    assert len(ridx) == N and np.all(np.diff(ridx) >= 0)
    counts = np.bincount(ridx, minlength=len(wt))
    assert np.all(np.abs(counts - N * wt) <= 1.0 + 1e-9)


def test_gradient_is_additive_over_the_batch():
    """The unweighted loss is a sum over rows (flow.py:309), so the gradient of a 512-row batch equals the sum of the
    gradients of its two halves (up to fp32 summation order) -- for the flows of configs 2/4 and 3."""
    from pocomc_amd.train import loss_and_grad, _train_state
    for D, name in ((32, "maf3"), (50, "maf6"), (32, "nsf3")):
        f = _flow(D, name)
        _train_state(f).repack(f)
        x = (torch.randn(512, D, generator=torch.Generator().manual_seed(5)) * 1.2).cuda()
        la = float(loss_and_grad(f, x))
        g = f._train.grad.clone()
        l1 = float(loss_and_grad(f, x[:256].contiguous()))
        g1 = f._train.grad.clone()
        l2 = float(loss_and_grad(f, x[256:].contiguous()))
        g2 = f._train.grad.clone()
        np.testing.assert_allclose(la, l1 + l2, rtol=2e-6)
        scale = g.abs().max().item()
        assert (g - (g1 + g2)).abs().max().item() < 2e-5 * scale, name
