"""Pin the oracle: every golden vector generated from the reference itself
(``tests/golden/make_golden.py``) and the reference's own known-answer /
property tests (``tests/test_tools.py:10-14``, ``tests/test_scaler.py``,
``tests/test_flow.py``)."""
import os

import numpy as np
import pytest

import cases
from oracle import mcmc as omcmc
from oracle import tools as otools
from oracle.maf import OracleMAF, TorchFlowAdapter
from oracle.scaler import Reparameterize
from pocomc_amd.maf_spec import MAFSpec


@pytest.fixture(scope="module")
def gold(golden_dir):
    return {k: np.load(os.path.join(golden_dir, f"{k}_reference.npz")) for k in ("mcmc", "scaler", "tools", "mcmc_big", "sampler")}


# ----------------------------------------------------------------- MCMC kernels
@pytest.mark.parametrize("name", list(cases.MCMC_CASES))
@pytest.mark.parametrize("exact", [True, False])
def test_mcmc_kernels_match_reference(gold, name, exact):
    g = gold["mcmc"]
    c = cases.MCMC_CASES[name]
    for n_max in sorted({1, c["n_max"]}):
        state, funcs, opts, aux = cases.build_case(name, Reparameterize)
        # the inputs are the ones the reference saw
        for k in ("u", "x", "logdetj", "logl", "logp"):
            np.testing.assert_allclose(state[k], g[f"mcmc/{name}/in/{k}"], rtol=1e-13, atol=1e-13)
        funcs["flow"] = TorchFlowAdapter(OracleMAF(aux["spec"], aux["flat"]))
        opts["n_max"] = n_max
        np.random.seed(c["seed"])
        res = getattr(omcmc, c["kind"])(state, funcs, opts, exact=exact)
        tag = f"mcmc/{name}/nmax{n_max}"
        assert res["steps"] == int(g[f"{tag}/steps"])
        assert res["calls"] == int(g[f"{tag}/calls"])
        tol = dict(rtol=0, atol=0) if exact else dict(rtol=1e-11, atol=1e-11)
        for k in ("u", "x", "logdetj", "logl", "logp"):
            np.testing.assert_allclose(res[k], g[f"{tag}/{k}"], err_msg=f"{tag}/{k}", **tol)
        for k in ("efficiency", "accept", "proposal_scale"):
            np.testing.assert_allclose(res[k], g[f"{tag}/{k}"], rtol=1e-12)


@pytest.mark.parametrize("name", list(cases.BIG_GOLDEN_CASES))
def test_one_step_calls_at_baseline_size_match_reference(gold, name):
    """BASELINE's size (1e4 x 32), one step, affine and spline flow: the oracle's vectorised kernel against the row
    subsample of the reference's output (every walker's step depends on its own row only)."""
    g = gold["mcmc_big"]
    c = cases.BIG_GOLDEN_CASES[name]
    state, funcs, opts, aux = cases.build_case(name, Reparameterize)
    sl = slice(None, None, c["stride"])
    tag = f"mcmc_big/{name}"
    for k in ("u", "x", "logdetj", "logl", "logp"):
        np.testing.assert_allclose(state[k][sl], g[f"{tag}/in/{k}"], rtol=1e-13, atol=1e-13)
    funcs["flow"] = TorchFlowAdapter(OracleMAF(aux["spec"], aux["flat"]))
    opts["n_max"] = 1
    np.random.seed(c["seed"])
    res = omcmc.preconditioned_pcn(state, funcs, opts, exact=False)
    assert res["steps"] == int(g[f"{tag}/steps"]) and res["calls"] == int(g[f"{tag}/calls"])
    for k in ("u", "x", "logdetj", "logl", "logp"):
        np.testing.assert_allclose(res[k][sl], g[f"{tag}/{k}"], rtol=1e-11, atol=1e-11, err_msg=k)
    for k in ("efficiency", "accept", "proposal_scale"):
        np.testing.assert_allclose(res[k], g[f"{tag}/{k}"], rtol=1e-12)


def test_replay_reproduces_stream(gold):
    """Replay of recorded variates == the legacy-stream run."""
    name = "tpcn_n64_d4_uniform"
    c = cases.MCMC_CASES[name]
    state, funcs, opts, aux = cases.build_case(name, Reparameterize)
    funcs["flow"] = TorchFlowAdapter(OracleMAF(aux["spec"], aux["flat"]))
    rng = omcmc.LegacyStream()
    np.random.seed(c["seed"])
    r1 = omcmc.preconditioned_pcn(state, funcs, opts, rng=rng)
    state, funcs, opts, aux = cases.build_case(name, Reparameterize)
    funcs["flow"] = TorchFlowAdapter(OracleMAF(aux["spec"], aux["flat"]))
    r2 = omcmc.preconditioned_pcn(state, funcs, opts, rng=omcmc.Replay(rng.record))
    for k in ("u", "x", "logl"):
        np.testing.assert_array_equal(r1[k], r2[k])


# ---------------------------------------------------------------------- scaler
@pytest.mark.parametrize("transform", ["probit", "logit"])
@pytest.mark.parametrize("bname", ["none", "left", "right", "both"])
def test_scaler_matches_reference(gold, transform, bname):
    g = gold["scaler"]
    tag = f"scaler/{transform}/{bname}"
    sc = Reparameterize(10, g[f"{tag}/bounds"], transform=transform)
    x = g[f"{tag}/x"]
    sc.fit(x)
    np.testing.assert_array_equal(sc.mu, g[f"{tag}/mu"])
    np.testing.assert_array_equal(sc.sigma, g[f"{tag}/sigma"])
    u = sc.forward(x)
    np.testing.assert_array_equal(u, g[f"{tag}/u"])
    xr, ldj = sc.inverse(u)
    np.testing.assert_array_equal(xr, g[f"{tag}/x_rt"])
    np.testing.assert_array_equal(ldj, g[f"{tag}/ldj"])
    # tests/test_scaler.py:73,92,111,130 round trip
    assert np.allclose(x, xr)
    xf, ldjf = sc.inverse(g[f"{tag}/u_far"])
    np.testing.assert_array_equal(xf, g[f"{tag}/x_far"])
    np.testing.assert_array_equal(ldjf, g[f"{tag}/ldj_far"])


def test_scaler_boundary_conditions(gold):
    g = gold["scaler"]
    sc = Reparameterize(4, g["scaler/bc/bounds"], periodic=[0, 1], reflective=[2])
    np.testing.assert_array_equal(sc.apply_boundary_conditions_x(g["scaler/bc/x"]), g["scaler/bc/x_bc"])


def test_scaler_out_of_bounds_raises():
    """tests/test_scaler.py:133-140."""
    sc = Reparameterize(3, np.tile(np.array([[0.0, 1.0]]), (3, 1)))
    with pytest.raises(ValueError):
        sc.fit(np.full((5, 3), 2.0))


# ----------------------------------------------------------------------- tools
def test_compute_ess_single_particle():
    """The reference's only KAT: tests/test_tools.py:10-14."""
    for v in (1.0, 251.0, -421.0, -421.125251, 0.0):
        assert otools.compute_ess(np.array([v])) == 1.0


@pytest.mark.parametrize("n", [1, 17, 1000, 5000])
def test_tools_match_reference(gold, n):
    g = gold["tools"]
    lw = g[f"tools/n{n}/logw"]
    w = np.exp(lw - lw.max())
    assert otools.effective_sample_size(w.copy()) == g[f"tools/n{n}/ess"]
    assert otools.unique_sample_size(w.copy()) == g[f"tools/n{n}/uss"]
    assert otools.unique_sample_size(w.copy(), k=64) == g[f"tools/n{n}/uss_k64"]
    assert otools.compute_ess(lw) == g[f"tools/n{n}/compute_ess"]
    assert otools.increment_logz(lw) == g[f"tools/n{n}/increment_logz"]
    if n >= 17:
        idx, wt = otools.trim_weights(np.arange(n), w.copy())
        np.testing.assert_array_equal(idx, g[f"tools/n{n}/trim_idx"])
        np.testing.assert_array_equal(wt, g[f"tools/n{n}/trim_w"])
        wn = w / w.sum()
        for s in (0, 1):
            got = otools.systematic_resample(min(n, 256), wn.copy(), offset=float(g[f"tools/n{n}/syst_{s}_offset"]))
            np.testing.assert_array_equal(got, g[f"tools/n{n}/syst_{s}"])
            got = otools.multinomial_resample(min(n, 256), wn, uniforms=g[f"tools/n{n}/mult_{s}_uniforms"])
            np.testing.assert_array_equal(got, g[f"tools/n{n}/mult_{s}"])


def test_logw_logz_match_reference(gold):
    g = gold["tools"]
    for bf in (0.3, 1.0):
        lw, lz = otools.compute_logw_and_logz(g["particles/logl"], g["particles/beta"], g["particles/logz"], bf)
        np.testing.assert_array_equal(lw, g[f"particles/logw_b{bf}"])
        assert lz == g[f"particles/logz_b{bf}"]
        lw2, _ = otools.compute_logw_and_logz(g["particles/logl"], g["particles/beta"], g["particles/logz"], bf, normalize=False)
        np.testing.assert_array_equal(lw2, g[f"particles/logw_raw_b{bf}"])


def test_geometry_matches_reference(gold):
    g = gold["tools"]
    G = otools.Geometry()
    G.fit(g["geometry/theta"])
    np.testing.assert_allclose(G.t_mean, g["geometry/t_mean"], rtol=1e-12)
    np.testing.assert_allclose(G.t_cov, g["geometry/t_cov"], rtol=1e-12)
    np.testing.assert_allclose(G.t_nu, g["geometry/t_nu"], rtol=1e-12)
    np.testing.assert_allclose(G.normal_cov, g["geometry/normal_cov"], rtol=1e-13)
    np.random.seed(11)
    G2 = otools.Geometry()
    G2.fit(g["geometry/theta"], weights=g["geometry/w"])
    np.testing.assert_allclose(G2.t_cov, g["geometry/w_t_cov"], rtol=1e-12)
    np.testing.assert_allclose(G2.t_nu, g["geometry/w_t_nu"], rtol=1e-12)
    np.testing.assert_allclose(G2.normal_cov, g["geometry/w_normal_cov"], rtol=1e-13)


# ------------------------------------------------- orchestrator: Sampler._compute_evidence
@pytest.mark.parametrize("name", ["maf3_d5", "nsf3_d4"])
def test_evidence_oracle_matches_reference(gold, name):
    """``oracle/evidence.py`` against ``Sampler._compute_evidence`` of the reference itself (``sampler.py:869-920`` called
    unbound by ``make_golden.py``; same base draw, same legacy-stream seed for the bootstrap)."""
    from oracle.evidence import compute_evidence
    g = gold["sampler"]
    tag = f"evidence/{name}"
    Dn, Tn, rqs, n, seed = (int(v) for v in g[f"{tag}/spec"])
    spec = MAFSpec(Dn, Tn, univariate="rqs" if rqs else "affine")
    maf = OracleMAF(spec, cases.flow_params(spec, seed, gain=1.0))
    prior = cases.CutPrior(Dn)
    sc = Reparameterize(Dn, prior.bounds)
    sc.fit(g[f"{tag}/x_fit"])
    like = lambda x: -0.5 * np.sum(((x - 0.3) / 0.8) ** 2, axis=1) - 0.1 * x[:, 0] ** 4
    logz, dlogz, draws, logw = compute_evidence(maf, sc, prior.logpdf, like, g[f"{tag}/z"], seed=seed)
    assert len(logw) == int(g[f"{tag}/calls"]) < n                 # part of the draws fell outside the prior's support
    np.testing.assert_allclose(logz, g[f"{tag}/logz"], rtol=1e-13)
    np.testing.assert_allclose(dlogz, g[f"{tag}/dlogz"], rtol=1e-12)


# ------------------------------------------------- flow: masks, the reference's properties
@pytest.mark.parametrize("D,T,H,uni", [(2, 2, None, "affine"), (4, 3, None, "affine"), (10, 3, None, "rqs"), (32, 3, None, "affine"),
                                       (50, 6, None, "affine"), (7, 2, 6, "rqs"), (9, 4, 19, "affine"), (128, 2, None, "affine")])
def test_oracle_masks_equal_the_product_masks(D, T, H, uni):
    """The oracle builds its masks from zuko's recipe (adjacency -> unique rows -> precedence, ``oracle.maf.zuko_masks``),
    the product from the closed form ``degree(k) = 1 + k mod (D - 1)`` (``MAFSpec.masks``): two constructions, one result."""
    from oracle.maf import zuko_masks
    spec = MAFSpec(D, T, hidden=H, univariate=uni)
    for t in range(T):
        order = np.arange(D) if t % 2 == 0 else np.arange(D)[::-1]
        assert np.array_equal(order, spec.orders[t])
        for a, b in zip(zuko_masks(order, spec.n_out, spec.hidden), spec.masks(t)):
            assert a.shape == b.shape and np.array_equal(a, b)


def test_float64_yardstick_of_the_oracle():
    """``OracleMAF(dtype=float64)``: the same float32 parameters, float64 arithmetic -- float32 and float64 evaluations
    agree to float32 rounding times the conditioning (affine: ~1e-6; spline: ~1e-5, the knot differences cancel)."""
    for uni, tol in (("affine", 5e-6), ("rqs", 1e-4)):
        spec = MAFSpec(6, 3, univariate=uni)
        flat = cases.flow_params(spec, 2)
        o32, o64 = OracleMAF(spec, flat), OracleMAF(spec, flat, dtype=np.float64)
        x = (np.random.default_rng(0).normal(size=(64, 6)) * 2.0).astype(np.float32)
        z32, l32 = o32.forward(x)
        z64, l64 = o64.forward(x)
        assert z32.dtype == np.float32 and z64.dtype == np.float64 and l64.dtype == np.float64
        assert (np.abs(z32 - z64).max(axis=1) / np.abs(z64).max(axis=1)).max() < tol
        xi64, li64 = o64.inverse(z64)
        assert np.abs(xi64 - x).max() < 1e-9 and np.abs(li64 + l64).max() < 1e-9      # the float64 round trip is exact to 1e-9


# ------------------------------------------------- flow: the reference's properties
def _flow_data():
    import torch
    torch.manual_seed(0)                       # tests/test_flow.py:8-13
    return (torch.randn(size=(100, 4)) * 1.5).numpy()


@pytest.mark.parametrize("D,T", [(4, 3), (10, 3), (7, 6), (32, 3)])
def test_flow_properties(D, T):
    """tests/test_flow.py: finite/shape/dtype (:16-72), round trip <= 1e-5 (:88),
    ladj antisymmetry (:164, :205)."""
    spec = MAFSpec(D, T)
    maf = OracleMAF(spec, cases.flow_params(spec, 3))
    x = _flow_data() if D == 4 else (np.random.default_rng(0).normal(size=(100, D)) * 1.5).astype(np.float32)
    z, ladj = maf.forward(x)
    assert z.shape == x.shape and z.dtype == np.float32 and np.isfinite(z).all()
    xr, ladj_inv = maf.inverse(z)
    assert np.allclose(x, xr, atol=1e-5)                      # tests/test_flow.py:88
    np.testing.assert_allclose(ladj, -ladj_inv, rtol=1e-5, atol=1e-5)
    lp = maf.log_prob(x)
    assert lp.shape == (100,) and lp.dtype == np.float32 and np.isfinite(lp).all()
    xs, lq = maf.sample_from(np.random.default_rng(1).normal(size=(50, D)).astype(np.float32))
    np.testing.assert_allclose(lq, maf.log_prob(xs), rtol=1e-4, atol=1e-4)


def test_flow_autoregressive_structure():
    """Jacobian of the hyper-network respects the masks: output of rank r
    depends only on inputs of rank < r (what makes the D-pass inverse exact)."""
    spec = MAFSpec(6, 2)
    maf = OracleMAF(spec, cases.flow_params(spec, 0))
    x = np.random.default_rng(0).normal(size=(1, 6)).astype(np.float32)
    for t in range(2):
        rank = spec.orders[t]
        s0, l0 = maf._hyper(t, x)
        for j in range(6):
            xp = x.copy(); xp[0, j] += 1.0
            s1, l1 = maf._hyper(t, xp)
            changed = (np.abs(s1 - s0) + np.abs(l1 - l0))[0] > 0
            assert not changed[rank <= rank[j]].any()
            assert changed[rank > rank[j]].all()


def test_torch_twin_matches_numpy():
    import torch
    from oracle.maf import torch_log_prob
    spec = MAFSpec(5, 3)
    flat = cases.flow_params(spec, 1)
    x = np.random.default_rng(0).normal(size=(40, 5)).astype(np.float32)
    a = OracleMAF(spec, flat).log_prob(x)
    b = torch_log_prob(spec, torch.from_numpy(flat), torch.from_numpy(x)).numpy()
    np.testing.assert_allclose(a, b, rtol=2e-5, atol=2e-5)
