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
    f16 = Flow(D, spec, inverse_precision=prec)
    f16.set_params(flat)
    assert f16.inverse_precision == prec and f16._desc.lane16
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
