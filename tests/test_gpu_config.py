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


# ------------------------------------------------------------------------------------------- bf16 matrix cores (N1)
@pytest.mark.parametrize("D,T,n", [(128, 8, 500), (128, 8, 33), (50, 6, 1000), (32, 3, 1000), (10, 3, 17), (4, 3, 5)])
def test_bf16_forward_and_logprob_against_the_fp32_oracle(D, T, n):
    """Flow(precision="bf16"): v_mfma_f32_16x16x32_bf16 with fp32 accumulation against the float32 oracle.
    Stated tolerance: weights and activations carry 8 mantissa bits (relative rounding 2^-9 each), accumulated over
    three hidden layers and T transforms -- z within 3e-2 of the row's scale, the log-determinant within
    3e-2 * (1 + sum |terms|) ** 0.5 ... measured maxima are printed; the float32 path stays the 1e-5 reference."""
    from pocomc_amd import Flow
    spec = MAFSpec(D, T)
    flat = cases.flow_params(spec, 3)
    f = Flow(D, spec, precision="bf16")
    f.set_params(flat)
    f32 = Flow(D, spec)
    f32.set_params(flat)
    o = OracleMAF(spec, flat)
    x = (np.random.default_rng(n).normal(size=(n, D)) * 1.5).astype(np.float32)
    z, ladj = f.forward(torch.from_numpy(x))
    zo, lo = o.forward(x)
    ez = np.abs(z.numpy() - zo).max(axis=1) / np.abs(zo).max(axis=1)
    terms = o.ladj_abs_terms(x)
    el = np.abs(ladj.numpy() - lo) / np.maximum(np.abs(lo), terms)
    lp = f.log_prob(torch.from_numpy(x)).numpy()
    lpo = o.log_prob(x)
    elp = np.abs(lp - lpo) / np.maximum(np.abs(lpo), terms + 0.5 * (zo.astype(np.float64) ** 2).sum(axis=1))
    print(f"bf16 D={D} T={T} n={n}: max rel err z {ez.max():.2e}, ladj {el.max():.2e}, log_prob {elp.max():.2e}")
    assert ez.max() < 3e-2 and el.max() < 3e-2 and elp.max() < 3e-2
    # the float32 kernel on the same parameters is the 1e-5 path
    z32, _ = f32.forward(torch.from_numpy(x))
    close_rel(z32.numpy(), zo, TOL, "fp32 forward")
    # the bf16 image follows the parameters
    f.set_params(flat * np.float32(0.5))
    z2, _ = f.forward(torch.from_numpy(x))
    o2 = OracleMAF(spec, flat * np.float32(0.5))
    assert (np.abs(z2.numpy() - o2.forward(x)[0]).max(axis=1) / np.abs(o2.forward(x)[0]).max(axis=1)).max() < 3e-2
    with pytest.raises(NotImplementedError):
        Flow(4, "nsf3", precision="bf16")


# ------------------------------------------------------------------ 16-bit helper operands of the lane sweep (config 5)
@pytest.mark.parametrize("D,T,n", [(128, 8, 33), (128, 8, 5000), (50, 6, 6496), (64, 3, 1000)])
@pytest.mark.parametrize("prec", ["bf16", "f16"])
def test_lane_sweep_with_16bit_helpers_against_the_float32_sweep(D, T, n, prec):
    """``Flow(inverse_precision="bf16" | "f16")``: the lane-per-walker sweep multiplies everything LEFT of the diagonal tile
    (the helper wavefronts' products: weights and activations rounded to 16 bits, float32 accumulation) on
    ``v_mfma_f32_16x16x32_bf16 / _f16``; the dependent chain stays float32.  Opt-in precision, stated tolerance (per walker,
    against the float32 sweep of the same flow, itself held to 1e-5 against the oracle above; n = 33 also against the
    oracle's D-pass inverse): bf16 -- x within 6e-2 of the row's scale, the log-determinant within 0.3 absolute (it sums
    T D log-scales each good to ~2^-9 of its hyper-network's terms); f16 (three more mantissa bits) -- 6e-3 and 4e-2.
    Measured on MI355X at (128, 8), worst of 5000 walkers (median): bf16 3.8e-2 (2.9e-3) / 0.107, f16 2.8e-3 (3.6e-4) / 0.012.  These are flows of the default initialisation
    x 1.2: a TRAINED flow can condition the sweep much worse (bench.py reports ``inverse_16bit_vs_f32`` on its trained flow)."""
    from pocomc_amd import Flow
    spec = MAFSpec(D, T)
    flat = cases.flow_params(spec, 3)
    f32 = Flow(D, spec)
    f32.set_params(flat)
    f16 = Flow(D, spec, inverse_precision=prec, inverse_guard=False)       # (the raw 16-bit sweep: the guard has its own test below)
    f16.set_params(flat)
    assert f16.inverse_precision == prec and f16._desc.lane16 and f16.inverse_guard is None
    z = (np.random.default_rng(n).normal(size=(n, D)) * 1.2).astype(np.float32)
    f32.inverse_algo = 8                                     # PMC_INVERSE_TRIANGULAR_LANE: float32 helpers
    xr, lr = (t.numpy() for t in f32.inverse(torch.from_numpy(z)))
    x, l = (t.numpy() for t in f16.inverse(torch.from_numpy(z)))            # AUTO: the 16-bit helpers
    fin = np.isfinite(xr).all(axis=1)
    assert fin.mean() > 0.995 and np.isfinite(x[fin]).all()
    ex = np.abs(x - xr)[fin].max(axis=1) / np.abs(xr[fin]).max(axis=1)
    el = np.abs(l - lr)[fin]
    print(f"{prec} helpers D={D} T={T} n={n}: x max {ex.max():.2e} median {np.median(ex):.2e}; ladj max {el.max():.2e} median {np.median(el):.2e}")
    tx, tl = (6e-2, 0.3) if prec == "bf16" else (6e-3, 4e-2)
    assert ex.max() < tx and el.max() < tl
    assert ex.max() > 1e-6                                   # (it IS the 16-bit path: the float32 sweeps agree to 1e-6)
    # explicit algorithm ids: 9 = 16-bit helpers (needs the image), 8 = float32 helpers whatever is attached
    f16.inverse_algo = 9
    x9, _ = f16.inverse(torch.from_numpy(z[:64]))
    np.testing.assert_array_equal(x9.numpy(), x[:64])
    f16.inverse_algo = 8
    x8, _ = f16.inverse(torch.from_numpy(z[:64]))
    np.testing.assert_array_equal(x8.numpy(), xr[:64])
    with pytest.raises(Exception):
        f32.inverse_algo = 9
        f32.inverse(torch.from_numpy(z[:16]))
    if n <= 64:
        xo, lo = OracleMAF(spec, flat).inverse(z)
        eo = (np.abs(x - xo).max(axis=1) / np.abs(xo).max(axis=1)).max()
        assert eo < tx
    # the image follows the parameters (set_params / the end of fit)
    f16.inverse_algo = 0
    f16.set_params(flat * np.float32(0.5))
    f32.set_params(flat * np.float32(0.5))
    f32.inverse_algo = 8
    xa, _ = f16.inverse(torch.from_numpy(z[:256]))
    xb, _ = f32.inverse(torch.from_numpy(z[:256]))
    assert (np.abs(xa.numpy() - xb.numpy()).max(axis=1) / np.abs(xb.numpy()).max(axis=1)).max() < tx


def test_the_16bit_sweep_is_guarded():
    """``Flow.inverse`` is an INVERSE (``pocomc/flow.py:116-132``): whenever the parameters of a flow with
    ``inverse_precision != "f32"`` change, the 16-bit and the float32 sweep are compared on latent points of the flow and the
    flow goes back to float32, with a warning, if a walker's x differs by more than ``LANE16_BOUND`` (1e-2 relative) or its
    log-determinant by more than ``LANE16_LADJ_BOUND`` (0.1; 99th percentile ``LANE16_LADJ_Q99_BOUND``, 5e-2).  Default
    construction and old checkpoints are float32."""
    import pickle
    import warnings
    from pocomc_amd import Flow
    from pocomc_amd.flow import LANE16_BOUND
    D, T = 128, 8
    spec = MAFSpec(D, T)
    flat = cases.flow_params(spec, 3)
    # precision="bf16" alone no longer switches the inverse to 16 bits
    assert Flow(D, spec, precision="bf16").inverse_precision == "f32"
    z = torch.from_numpy((np.random.default_rng(5).normal(size=(512, D)) * 1.2).astype(np.float32))
    ref = Flow(D, spec)
    ref.set_params(flat)
    ref.inverse_algo = 8
    xr, lr = ref.inverse(z)
    seen = {}
    for prec in ("bf16", "f16"):
        f = Flow(D, spec, inverse_precision=prec)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            f.set_params(flat * np.float32(1.2))              # a flow of 1.2 x the default initialisation: bf16 is 4e-2 off there
        g = f.inverse_guard
        assert g is not None and g["rows"] >= 1024 and g["precision"] == prec
        seen[prec] = (g["x_rel_err_max"], g["ladj_abs_err_max"], g["passed"])
        fired = any("falling back to the float32 sweep" in str(m.message) for m in w)
        assert fired == (not g["passed"])
        assert f.inverse_precision_active == (prec if g["passed"] else "f32")
        if not g["passed"]:
            # AUTO is the float32 sweep now, bit for bit
            r32 = Flow(D, spec)
            r32.set_params(flat * np.float32(1.2))
            r32.inverse_algo = 8
            xa, la = f.inverse(z)
            xb, lb = r32.inverse(z)
            np.testing.assert_array_equal(xa.numpy(), xb.numpy())
            np.testing.assert_array_equal(la.numpy(), lb.numpy())
            # the verdict travels with a checkpoint: a flow that fell back reloads on float32 (no re-check on other points)
            r = pickle.loads(pickle.dumps(f))
            assert r.inverse_precision == prec and r.inverse_precision_active == "f32" and r.inverse_guard["passed"] is False
            # new parameters get a new verdict: a tame flow passes again
            f.set_params(flat * np.float32(0.25))
            assert f.inverse_guard["passed"] and f.inverse_precision_active == prec
    print("guard at 1.2 x init:", seen)
    assert not seen["bf16"][2] and seen["bf16"][0] > LANE16_BOUND       # (measured 3.8e-2 in round 4)
    # checkpoints: the key travels; a state without it reloads as float32
    f = Flow(D, spec, inverse_precision="f16")
    f.set_params(flat * np.float32(0.25))
    g = pickle.loads(pickle.dumps(f))
    assert g.inverse_precision == "f16" and g.inverse_guard is not None
    st = f.__getstate__()
    st.pop("inverse_precision")
    h = Flow.__new__(Flow)
    h.__setstate__(st)
    assert h.inverse_precision == "f32" and h._lane16 is None


@pytest.mark.parametrize("prec", ["f16", "bf16"])
def test_the_guard_on_a_trained_config3_flow(prec):
    """BASELINE config 3's flow (maf6 at D = 50) trained as ``bench.py`` trains it (50 epochs on prior draws through the
    scaler): round 4 measured a maximum relative error of 33.8 on x for the f16 sweep on that flow.  ``Flow.fit`` ends with
    the guard on the latent image of its training rows: either the sweep is within the bounds or the flow is back on float32."""
    from pocomc_amd import Flow, Reparameterize
    D, n = 50, 10000
    rng = np.random.default_rng(7)
    x_fit = rng.uniform(-10.0, 10.0, size=(2 * n, D))
    scaler = Reparameterize(D, bounds=np.array([[-10.0, 10.0]] * D))
    scaler.fit(x_fit)
    torch.manual_seed(0)
    u_fit = torch.from_numpy(scaler.forward(x_fit[:n])).float()
    f = Flow(D, "maf6", seed=0, inverse_precision=prec)
    import warnings
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        hist = f.fit(u_fit, epochs=50, batch_size=512, validation_split=0.5, patience=D, annealing=False, verbose=0)
    g = hist["inverse_guard"]
    print(f"config-3 flow, {prec} sweep after {len(hist['loss'])} epochs: {g}")
    assert g is f.inverse_guard and g["rows"] >= 1024
    if g["passed"]:
        assert g["x_rel_err_max"] <= g["bound"] and g["ladj_abs_err_max"] <= g["ladj_bound"] and g["rows_lost_by_16bit"] == 0
        assert f.inverse_precision_active == prec
    else:
        assert f.inverse_precision_active == "f32" and any("falling back" in str(m.message) for m in w)
    # whatever the verdict: what the step calls now is an inverse of the flow's forward map within the guard's bound
    th = f.forward(u_fit[:2048])[0]
    back = f.inverse(th)[0]
    ok = torch.isfinite(back).all(dim=1)
    err = ((back - u_fit[:2048]).abs().max(dim=1).values / u_fit[:2048].abs().max(dim=1).values)[ok]
    print(f"round trip through forward / inverse ({f.inverse_precision_active}): max {float(err.max()):.3g} median {float(err.median()):.3g}, rows {int(ok.sum())}")
    assert float(err.median()) < 1e-2
