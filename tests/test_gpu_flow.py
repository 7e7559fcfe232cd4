"""Parity of the HIP flow kernels (through the C ABI) against the oracle, plus the
reference's own property tests (``tests/test_flow.py``) on the product ``Flow``.

Tolerance (``tests/parity.py``): the north star asks for 1e-5 relative in fp32.  The affine flows (MAF) are held to
exactly that, walker by walker and pure relative (``close_rel``: a walker's coordinates are one vector; a log-determinant
or log-density -- a sum of terms of either sign -- is measured against the size of its terms, ``cancel=``, stated per
call).  Measured maxima over the shapes below (printed at the end of a run, ``conftest.py``): z 2.5e-6, x 2.3e-6,
log-determinant 1.2e-6, log_prob 4.1e-6.  The spline flows (NSF) are held to 2e-5 (z) / 5e-5 (x) / 1e-4 (log-determinant),
same measure: a rational-quadratic bin amplifies an input's rounding by up to (bin height / bin width) x (derivative
ratio) -- the knots come out of a softmax whose smallest bin is 1e-3 of the box -- and its log-derivative is a difference of
logarithms of O(1) quantities, so two valid float32 evaluations of the same spline (numpy's IEEE division / exp / log in
the oracle, v_rcp / v_exp / v_log at 1 ulp in the kernels) differ by a few ulp x that factor: measured 1.1e-5 (z),
1.8e-5 (x), 2.4e-5 / 7.2e-5 (log-determinant forward / inverse), while the device's own forward and inverse agree with
each other to 8e-6 (round trip) / 5e-6 (antisymmetry).  ``close`` (absolute slack scaled by the array) is used only where
both sides are device results of different algorithms."""
import warnings

import numpy as np
import pytest
import torch

import cases
from oracle.maf import OracleMAF
from parity import close_rel, TOL
from pocomc_amd.maf_spec import MAFSpec

pytestmark = pytest.mark.gpu

SHAPES = [(2, 3), (3, 3), (4, 3), (5, 3), (7, 6), (10, 3), (17, 2), (32, 3), (50, 6)]


def close(a, b, tol=1e-5):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    scale = max(1.0, float(np.abs(b).max()) if b.size else 1.0)
    np.testing.assert_allclose(a, b, rtol=tol, atol=tol * scale)


def make(D, T, seed=3):
    from pocomc_amd import Flow
    spec = MAFSpec(D, T)
    flat = cases.flow_params(spec, seed)
    f = Flow(D, spec)
    f.set_params(flat)
    return f, OracleMAF(spec, flat)


@pytest.mark.parametrize("D,T", SHAPES)
@pytest.mark.parametrize("n", [1, 16, 100, 1000])
def test_forward_logprob_matches_oracle(D, T, n):
    f, o = make(D, T)
    x = (np.random.default_rng(n).normal(size=(n, D)) * 1.5).astype(np.float32)
    z, ladj = f.forward(torch.from_numpy(x))
    zo, lo = o.forward(x)
    terms = o.ladj_abs_terms(x)                                            # size of the log-determinant's terms, per walker
    close_rel(z.numpy(), zo, TOL, "z")
    close_rel(ladj.numpy(), lo, TOL, "ladj", cancel=terms)
    base = 0.5 * (zo.astype(np.float64) ** 2).sum(axis=1) + 0.5 * D * np.log(2 * np.pi)
    close_rel(f.log_prob(torch.from_numpy(x)).numpy(), o.log_prob(x), TOL, "log_prob", cancel=terms + base)


@pytest.mark.parametrize("D,T", SHAPES)
@pytest.mark.parametrize("n", [1, 33, 500])
def test_inverse_matches_oracle(D, T, n):
    f, o = make(D, T)
    z = (np.random.default_rng(7 + n).normal(size=(n, D)) * 1.2).astype(np.float32)
    xo, lo = o.inverse(z)                       # the reference's D-pass algorithm
    small = f.spec.nOT <= 8 and 2 * f.spec.Dp + 3 * f.spec.Hp + 176 <= 2560       # D <= 64, tiles fit the LDS (pmc_maf_inverse checks the 3-layer budget)
    # triangular (AUTO's pick), D-pass on the device, lane-per-walker sweep, one- / two-wave register-chain sweeps
    terms = o.ladj_abs_terms(xo)
    for algo in ([1, 2, 8] + ([6, 7] if small else []) if f.spec.tri_ok else [2]):
        f.inverse_algo = algo
        x, l = f.inverse(torch.from_numpy(z))
        close_rel(x.numpy(), xo, TOL, f"x, algorithm {algo}")
        close_rel(l.numpy(), lo, TOL, f"ladj, algorithm {algo}", cancel=terms)
    f.inverse_algo = 0
    x, l = f.inverse(torch.from_numpy(z))
    close_rel(x.numpy(), xo, TOL, "x, AUTO")
    close_rel(l.numpy(), lo, TOL, "ladj, AUTO", cancel=terms)


@pytest.mark.parametrize("n", [1, 17, 4096, 10000])
@pytest.mark.parametrize("D,T", [(32, 3), (10, 6), (64, 3), (64, 6)])
def test_the_two_wave_sweep_does_not_depend_on_the_launch_and_agrees_with_the_lone_wave(D, T, n):
    """PMC_INVERSE_TRIANGULAR_DUO (chain + burst wavefront, right-looking: the round-3 sweep) is what AUTO takes for
    every call size of the flows with fewer than 16 hidden tiles, so the size and the sharding of a walker set
    (lanes, ranks) cannot show in the results: a row's x and log-determinant are bit-identical whether it is swept
    alone, in a set of 16 or in the whole call.  PMC_INVERSE_TRIANGULAR_SOLO (one wavefront per 16 rows) forms the same
    sums in another order (left-looking): it agrees to float32 rounding, measured walker by walker."""
    f, _ = make(D, T)
    z = torch.randn(n, D, generator=torch.Generator().manual_seed(n))
    out = {}
    for algo in (6, 7, 0):
        f.inverse_algo = algo
        out[algo] = [t.numpy() for t in f.inverse(z)]
    f.inverse_algo = 7
    for lo, hi in ((0, 1), (n // 2, n // 2 + 16), (max(n - 19, 0), n)):
        hi = min(hi, n)
        part = [t.numpy() for t in f.inverse(z[lo:hi])]
        np.testing.assert_array_equal(part[0], out[7][0][lo:hi])
        np.testing.assert_array_equal(part[1], out[7][1][lo:hi])
    f.inverse_algo = 0
    fin = np.isfinite(out[6][0]).all(axis=1) & np.isfinite(out[6][1])
    assert (fin == (np.isfinite(out[7][0]).all(axis=1) & np.isfinite(out[7][1]))).all()
    assert fin.mean() > 0.99
    close_rel(out[7][0][fin], out[6][0][fin], 2e-6, "two-wave vs lone-wave sweep, x")
    if f.spec.nT < 16:             # (from 16 hidden tiles on AUTO is the lane-per-walker sweep: another order of additions)
        np.testing.assert_array_equal(out[0][0], out[7][0])
        np.testing.assert_array_equal(out[0][1], out[7][1])
    else:
        close(out[0][0], out[6][0])
        close(out[0][1], out[6][1])
    import ctypes as C
    from pocomc_amd import _lib
    if f.spec.nT < 16:
        assert _lib.load().pmc_maf_inverse_auto_is_duo(C.byref(f._desc), n) == 1


@pytest.mark.parametrize("n", [1, 33, 700])
def test_wide_sweep_with_four_and_five_wavefronts_agrees_bit_for_bit(n):
    """The lane-per-walker sweep of a wide flow (>= 20 hidden tiles: (D, T, H) = (128, 2, 512)) gives the layer-0
    partials to a fifth wavefront and assigns the roles by SIMD; PMC_MAF_VARIANT_LANE_FOUR keeps them on the output wavefront.  Same
    additions in the same order: identical results, and both within 1e-5 of the oracle."""
    from pocomc_amd.maf_spec import MAFSpec
    import pocomc_amd as pc
    spec = MAFSpec(128, 2)
    f = pc.Flow(128, spec, seed=5)
    z = torch.randn(n, 128, generator=torch.Generator().manual_seed(n))
    f.inverse_algo = 0
    x5, l5 = [t.cpu().numpy() for t in f.inverse(z.cuda())]
    f._desc.reserved = 2                                   # PMC_MAF_VARIANT_LANE_FOUR
    x4, l4 = [t.cpu().numpy() for t in f.inverse(z.cuda())]
    np.testing.assert_array_equal(x5, x4)
    np.testing.assert_array_equal(l5, l4)
    f.inverse_algo = 2                                     # D-pass cross-check kernel (no triangular structure used)
    xd, ld = [t.cpu().numpy() for t in f.inverse(z.cuda())]
    close(x5, xd, 1e-5)
    close(l5, ld, 1e-5)


def test_triangular_equals_naive_on_device():
    f, _ = make(32, 3)
    z = torch.randn(4096, 32, generator=torch.Generator().manual_seed(0))
    f.inverse_algo = 1
    x1, l1 = f.inverse(z)
    f.inverse_algo = 2
    x2, l2 = f.inverse(z)
    close(x1.numpy(), x2.numpy(), 2e-6)
    close(l1.numpy(), l2.numpy(), 2e-6)


def test_empty_input():
    f, _ = make(4, 3)
    z, l = f.forward(torch.empty(0, 4))
    assert z.shape == (0, 4) and l.shape == (0,)
    x, l = f.inverse(torch.empty(0, 4))
    assert x.shape == (0, 4)


def test_sample_matches_oracle():
    f, o = make(10, 3)
    z = np.random.default_rng(5).normal(size=(257, 10)).astype(np.float32)
    x, lq = f.sample(257, z=torch.from_numpy(z))
    xo, lqo = o.sample_from(z)
    close_rel(x.numpy(), xo, TOL, "x")
    base = 0.5 * (z.astype(np.float64) ** 2).sum(axis=1) + 5.0 * np.log(2 * np.pi)
    close_rel(lq.numpy(), lqo, TOL, "log q", cancel=o.ladj_abs_terms(xo) + base)


# ----------------------------------------------- the reference's tests/test_flow.py
def _data():
    torch.manual_seed(0)
    return torch.randn(size=(100, 4)) * 1.5           # tests/test_flow.py:8-13


def test_reference_flow_properties():
    from pocomc_amd import Flow
    torch.manual_seed(0)
    data = _data()
    flow = Flow(n_dim=4, flow="maf3")
    z, ladj = flow.forward(data)                       # :16-29
    assert not torch.any(torch.isnan(z)) and not torch.any(torch.isinf(z))
    assert z.shape == data.shape and z.dtype == data.dtype
    x, ladj_inv = flow.inverse(z)                      # :75-88
    assert x.shape == data.shape and x.dtype == data.dtype
    assert torch.allclose(data, x, atol=1e-5)
    torch.testing.assert_close(ladj, -ladj_inv, rtol=1e-5, atol=1e-5)   # :164, :205
    lp = flow.log_prob(data)                           # :46-58
    assert lp.shape == (100,) and lp.dtype == data.dtype and torch.isfinite(lp).all()
    xs, lq = flow.sample(100)                          # :61-72
    assert xs.shape == data.shape and xs.dtype == data.dtype and torch.isfinite(xs).all()


def test_reference_float64_warns_and_casts():
    from pocomc_amd import Flow
    flow = Flow(n_dim=4, flow="maf3")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        lp = flow.log_prob(_data().double())           # tests/test_flow.py:106-119
        assert any("Float64" in str(i.message) for i in w)
    assert lp.dtype == torch.float32
    with pytest.raises(ValueError):
        flow.forward(_data().to(torch.float16))        # tools.py:316
    z, _ = flow.forward(_data()[:1])                   # single row, :122-134
    assert z.shape == (1, 4)


def test_flow_names():
    from pocomc_amd import Flow
    assert Flow(6, "maf6").spec.n_transforms == 6
    assert Flow(6, "maf12").spec.n_transforms == 12
    with pytest.raises(ValueError):
        Flow(6, "bogus")
    f = Flow(6, "nsf6")                                # pocomc/flow.py:75-80
    assert f.spec.n_transforms == 6 and f.spec.univariate == "rqs" and f.spec.n_out == 23


# ----------------------------------------------------------------- neural spline flows
NSF_SHAPES = [(2, 3), (4, 3), (10, 3), (17, 2), (32, 3), (40, 2), (50, 6)]     # (40: the widest flow of the static burst tile; 50: the streamed path)
NSF_FWD, NSF_INV, NSF_LADJ = 2e-5, 5e-5, 1e-4        # per walker, pure relative (the header says why not 1e-5)


def make_nsf(D, T, seed=3, gain=1.0, H=None, bins=8):
    from pocomc_amd import Flow
    spec = MAFSpec(D, T, H, univariate="rqs", bins=bins)
    flat = cases.flow_params(spec, seed, gain=gain)
    f = Flow(D, spec)
    f.set_params(flat)
    return f, OracleMAF(spec, flat)


@pytest.mark.parametrize("D,T", NSF_SHAPES)
@pytest.mark.parametrize("n", [1, 16, 300])
def test_nsf_forward_logprob_matches_oracle(D, T, n):
    """Spline flows (pocomc/flow.py:69-86): values inside and outside the spline box [-5, 5]."""
    f, o = make_nsf(D, T)
    x = (np.random.default_rng(n).normal(size=(n, D)) * 2.5).astype(np.float32)
    z, ladj = f.forward(torch.from_numpy(x))
    zo, lo = o.forward(x)
    terms = o.ladj_abs_terms(x)
    close_rel(z.numpy(), zo, NSF_FWD, "nsf z")
    close_rel(ladj.numpy(), lo, NSF_LADJ, "nsf ladj", cancel=terms)
    base = 0.5 * (zo.astype(np.float64) ** 2).sum(axis=1) + 0.5 * D * np.log(2 * np.pi)
    close_rel(f.log_prob(torch.from_numpy(x)).numpy(), o.log_prob(x), NSF_LADJ, "nsf log_prob", cancel=terms + base)


@pytest.mark.parametrize("bins", [4, 16])
@pytest.mark.parametrize("D,T,H,n", [(4, 3, None, 37), (10, 3, 100, 200), (17, 2, None, 16)])
def test_spline_flows_with_other_bin_counts_match_the_oracle(bins, D, T, H, n):
    """What ``pocomc/flow.py:87-88`` accepts beyond its named flows, as far as a ``MAFSpec`` expresses it: spline flows of 4 and
    16 bins (11 / 47 hyper-network outputs per feature).  Forward / log_prob, ``sample``'s direction (the inverse: zuko's
    own D-pass algorithm on the device -- the triangular sweeps are built for the default 8 bins and say so) and the round
    trip, against the oracle's arithmetic for that bin count."""
    from pocomc_amd import _lib
    f, o = make_nsf(D, T, H=H, bins=bins)
    assert f.spec.n_out == 3 * bins - 1
    x = (np.random.default_rng(n + bins).normal(size=(n, D)) * 2.5).astype(np.float32)
    z, ladj = f.forward(torch.from_numpy(x))
    zo, lo = o.forward(x)
    terms = o.ladj_abs_terms(x)
    close_rel(z.numpy(), zo, NSF_FWD, f"nsf z, {bins} bins")
    close_rel(ladj.numpy(), lo, NSF_LADJ, f"nsf ladj, {bins} bins", cancel=terms)
    base = 0.5 * (zo.astype(np.float64) ** 2).sum(axis=1) + 0.5 * D * np.log(2 * np.pi)
    close_rel(f.log_prob(torch.from_numpy(x)).numpy(), o.log_prob(x), NSF_LADJ, f"nsf log_prob, {bins} bins", cancel=terms + base)
    zz = (np.random.default_rng(7 + n).normal(size=(n, D)) * 1.5).astype(np.float32)
    xo, lio = o.inverse(zz)
    xi, li = f.inverse(torch.from_numpy(zz))                    # AUTO
    close_rel(xi.numpy(), xo, NSF_INV, f"nsf x, {bins} bins")
    close_rel(li.numpy(), lio, NSF_LADJ, f"nsf ladj inverse, {bins} bins", cancel=o.ladj_abs_terms(xo))
    back, lb = f.forward(xi)
    close_rel(back.numpy(), zz, 10 * NSF_INV, f"nsf round trip, {bins} bins")
    f.inverse_algo = 1
    with pytest.raises(_lib.PocomcAmdError, match="8 bins"):
        f.inverse(torch.from_numpy(zz))
    f.inverse_algo = 0
    with pytest.raises(NotImplementedError):
        MAFSpec(D, T, univariate="rqs", bins=5)


@pytest.mark.parametrize("D,T,H,uni", [(10, 3, 100, "affine"), (7, 3, 48, "affine"), (10, 3, 100, "rqs"), (12, 2, 11, "affine"),
                                       (33, 2, 200, "affine"), (9, 3, 72, "rqs")])
def test_hidden_widths_that_are_not_powers_of_two(D, T, H, uni):
    """``hidden_features`` of any width >= D - 1 (``flow.py:46-68`` takes a power of two; a zuko flow object need not):
    forward, every inverse sweep AUTO can pick, and the round trip against the oracle."""
    from pocomc_amd import Flow
    spec = MAFSpec(D, T, H, univariate=uni)
    flat = cases.flow_params(spec, 5)
    f = Flow(D, spec)
    f.set_params(flat)
    o = OracleMAF(spec, flat)
    tol_x, tol_l = (NSF_INV, NSF_LADJ) if uni == "rqs" else (TOL, TOL)
    x = (np.random.default_rng(H).normal(size=(150, D)) * 1.5).astype(np.float32)
    z, ladj = f.forward(torch.from_numpy(x))
    zo, lo = o.forward(x)
    close_rel(z.numpy(), zo, tol_x, f"z, H={H}")
    close_rel(ladj.numpy(), lo, tol_l, f"ladj, H={H}", cancel=o.ladj_abs_terms(x))
    xo, lio = o.inverse(x)
    for algo in ([0, 1, 2] if spec.tri_ok else [0, 2]):
        f.inverse_algo = algo
        xi, li = f.inverse(torch.from_numpy(x))
        close_rel(xi.numpy(), xo, tol_x, f"x, H={H}, algorithm {algo}")
        close_rel(li.numpy(), lio, tol_l, f"ladj inverse, H={H}, algorithm {algo}", cancel=o.ladj_abs_terms(xo))
    with pytest.raises(NotImplementedError):
        MAFSpec(D, T, D - 2)


@pytest.mark.parametrize("D,T", NSF_SHAPES)
@pytest.mark.parametrize("n", [1, 33, 200])
def test_nsf_inverse_matches_oracle(D, T, n):
    f, o = make_nsf(D, T)
    z = (np.random.default_rng(7 + n).normal(size=(n, D)) * 1.5).astype(np.float32)
    xo, lo = o.inverse(z)                       # the reference's D-pass algorithm
    terms = o.ladj_abs_terms(xo)
    for algo in ([1, 6, 7, 2] if f.spec.tri_ok else [2]):     # triangular sweeps (AUTO's choice, lone wave, two waves), D-pass on the device
        f.inverse_algo = algo
        x, l = f.inverse(torch.from_numpy(z))
        close_rel(x.numpy(), xo, NSF_INV, f"nsf x, algorithm {algo}")
        close_rel(l.numpy(), lo, NSF_LADJ, f"nsf ladj inverse, algorithm {algo}", cancel=terms)
    f.inverse_algo = 0
    x, l = f.inverse(torch.from_numpy(z))
    close_rel(x.numpy(), xo, NSF_INV, "nsf x, AUTO")
    close_rel(l.numpy(), lo, NSF_LADJ, "nsf ladj inverse, AUTO", cancel=terms)
    # tests/test_flow.py:88 and :164 on the product: round trip and ladj antisymmetry
    z2, l2 = f.forward(x)
    close_rel(z2.numpy(), z, NSF_INV, "nsf round trip")
    close_rel(l2.numpy(), -l.numpy(), NSF_LADJ, "nsf antisymmetry", cancel=terms)


@pytest.mark.parametrize("D,T", NSF_SHAPES)
def test_nsf_float32_evaluations_against_the_float64_yardstick(D, T):
    """Why the spline flows are not held to 1e-5 against the float32 oracle: the rational-quadratic spline differences
    knots of size <= 5 that are cumulative sums of a softmax (bin widths ~0.1-1), so ANY float32 evaluation -- zuko's,
    the oracle's, a kernel's -- sits eps * cond away from the exact map of the same float32 parameters, with cond ~ 10-100.
    The yardstick is that exact map: ``OracleMAF(dtype=float64)``.  Asserted per batch: the kernels' distance to it is
    within the north star's 1e-5 for D >= 4 (measured on MI355X: <= 6.1e-6 over these shapes, forward and every inverse sweep;
    the float32 oracle's own distance: <= 5.3e-6; at D = 2 the float32 ORACLE is 1.1e-5 away and the kernel 1.3e-5) AND not larger than YARD x the float32 oracle's own distance to it or 1e-5,
    whichever is larger -- i.e. the device is as good a float32 evaluation of the reference's flow as float32 numpy is;
    both distances are printed."""
    f, o = make_nsf(D, T)
    o64 = OracleMAF(o.spec, o.flat, dtype=np.float64)
    from parity import rel_rows
    YARD = 2.0
    rng = np.random.default_rng(100 + D)
    n = 300
    x = (rng.normal(size=(n, D)) * 2.5).astype(np.float32)
    z64, l64 = o64.forward(x)
    z32, l32 = o.forward(x)
    zk, lk = f.forward(torch.from_numpy(x))
    terms = o.ladj_abs_terms(x)
    e_o, e_k = rel_rows(z32, z64).max(), rel_rows(zk.numpy(), z64).max()
    tiny = np.finfo(np.float64).tiny                 # (a row outside the spline box in every transform: ladj = 0 exactly)
    el_o = (np.abs(l32 - l64) / np.maximum(np.maximum(np.abs(l64), terms), tiny)).max()
    el_k = (np.abs(lk.numpy() - l64) / np.maximum(np.maximum(np.abs(l64), terms), tiny)).max()
    print(f"nsf D={D} T={T} forward: z oracle32 {e_o:.2e} kernel {e_k:.2e}; ladj oracle32 {el_o:.2e} kernel {el_k:.2e}")
    assert e_k <= max(YARD * e_o, 1e-5) and el_k <= max(YARD * el_o, 1e-5)
    assert (e_k <= TOL and el_k <= TOL) or D == 2
    z = (rng.normal(size=(n, D)) * 1.5).astype(np.float32)
    x64, li64 = o64.inverse(z)
    x32, li32 = o.inverse(z)
    terms = o.ladj_abs_terms(x32)
    for algo in ([0, 6, 7] if f.spec.tri_ok else [2]):
        f.inverse_algo = algo
        xk, lik = f.inverse(torch.from_numpy(z))
        e_o, e_k = rel_rows(x32, x64).max(), rel_rows(xk.numpy(), x64).max()
        el_o = (np.abs(li32 - li64) / np.maximum(np.maximum(np.abs(li64), terms), tiny)).max()
        el_k = (np.abs(lik.numpy() - li64) / np.maximum(np.maximum(np.abs(li64), terms), tiny)).max()
        print(f"nsf D={D} T={T} inverse algorithm {algo}: x oracle32 {e_o:.2e} kernel {e_k:.2e}; ladj oracle32 {el_o:.2e} kernel {el_k:.2e}")
        assert e_k <= max(YARD * e_o, 1e-5) and el_k <= max(YARD * el_o, 1e-5), (algo, e_k, e_o, el_k, el_o)
        assert (e_k <= TOL and el_k <= TOL) or D == 2, (algo, e_k, el_k)     # (D = 2: the float32 oracle itself is 1.1e-5 away)
    f.inverse_algo = 0


@pytest.mark.parametrize("D,T,n", [(32, 3, 300), (32, 6, 33), (33, 2, 100), (30, 2, 64), (40, 2, 17)])    # 9, 9, 8, 11, 10 hidden tiles
def test_eager_partials_of_the_spline_sweep_change_no_bit(D, T, n):
    """The burst wave of the two-wave spline sweep forms the output partials of the last two hidden tiles early (steps 2-4,
    csrc/maf_inverse_nsf2.hip: NSF2_EAGER_OK -- flows of >= 8 hidden tiles); PMC_MAF_VARIANT_LEFT_LOOKING keeps the left-looking schedule.
    Every partial receives its K tiles in ascending order either way: identical results, plain and fused launches alike."""
    f, o = make_nsf(D, T)
    assert f.spec.device_meta()[7] >= 8, "the shape does not reach the eager path"
    z = (np.random.default_rng(3 * n + D).normal(size=(n, D)) * 1.5).astype(np.float32)
    f.inverse_algo = 7
    xe, le = [t.numpy() for t in f.inverse(torch.from_numpy(z))]
    f._desc.reserved = 1                                   # PMC_MAF_VARIANT_LEFT_LOOKING
    xl, ll = [t.numpy() for t in f.inverse(torch.from_numpy(z))]
    np.testing.assert_array_equal(xe, xl)
    np.testing.assert_array_equal(le, ll)
    xo, lo = o.inverse(z)
    close_rel(xe, xo, NSF_INV, "nsf x, eager partials")


def test_spline_sweeps_on_random_flow_shapes():
    """Random spline flows (D <= 64, T, hidden, n): the lone-wave and the two-wave sweep (static burst tiles up to 11 live
    hidden tiles, streamed above) agree to float32 rounding and follow the D-pass inverse on the device."""
    from pocomc_amd import Flow
    rng = np.random.default_rng(11)
    done, worst64 = 0, 0.0
    for case in range(24):
        D = int(rng.integers(2, 65))
        T = int(rng.integers(1, 5))
        H = max(int(rng.choice([max(D - 1, 4), D + 3, 2 * D, 3 * D + 1, 128, 4 * (D - 1) + 5])), D - 1)
        n = int(rng.choice([1, 15, 16, 17, 100, 1000]))
        spec = MAFSpec(D, T, hidden=H, univariate="rqs")
        if not spec.tri_ok:
            continue
        f = Flow(D, spec, seed=case)
        f.set_params(cases.flow_params(spec, case))
        z = torch.randn(n, D, generator=torch.Generator().manual_seed(case)) * 1.5
        out = {}
        for algo in (2, 6, 7):
            f.inverse_algo = algo
            out[algo] = [t.numpy() for t in f.inverse(z)]
        sc = np.maximum(1.0, np.abs(out[2][0]).max(axis=1, keepdims=True))
        assert np.isfinite(out[7][0]).all() and np.isfinite(out[6][0]).all(), (D, T, H, n)
        assert (np.abs(out[7][0] - out[6][0]) / sc).max() < 2e-4, (D, T, H, n)
        assert (np.abs(out[7][0] - out[2][0]) / sc).max() < 5e-4, (D, T, H, n)
        assert np.abs(out[7][1] - out[6][1]).max() < 1e-3 * max(1.0, float(np.abs(out[6][1]).max())), (D, T, H, n)
        # ... and, on up to 32 rows, the float64 evaluation of the same parameters by the oracle (zuko's D-pass algorithm in
        # numpy): not a device-against-device statement
        k = min(n, 32)
        x64, l64 = OracleMAF(spec, cases.flow_params(spec, case), dtype=np.float64).inverse(z[:k].numpy())
        ok = np.isfinite(x64).all(axis=1)
        e64 = float((np.abs(out[7][0][:k][ok] - x64[ok]) / np.maximum(1.0, np.abs(x64[ok]).max(axis=1, keepdims=True))).max()) if ok.any() else 0.0
        worst64 = max(worst64, e64)
        assert e64 < 5e-5, (D, T, H, n, e64)                 # (the spline's stated bound; measured 8.3e-6)
        done += 1
    print(f"spline sweeps on random shapes: two-wave sweep against the float64 oracle, worst row {worst64:.2e}")
    assert done >= 15


def test_sweeps_on_random_flow_shapes():
    """Random (D <= 64, T, hidden, n): the lone-wave (left-looking) and the two-wave (right-looking) sweep agree to
    float32 rounding and follow the D-pass inverse on the device (``scripts/fuzz_inverse.py`` is the long version)."""
    from pocomc_amd import Flow
    rng = np.random.default_rng(7)
    done, worst64 = 0, 0.0
    for case in range(40):
        D = int(rng.integers(2, 65))
        T = int(rng.integers(1, 8))
        H = max(int(rng.choice([max(D - 1, 4), D + 3, 2 * D, 3 * D + 1, 128, 4 * (D - 1) + 5])), D - 1)
        n = int(rng.choice([1, 15, 16, 17, 100, 1000, 5000]))
        spec = MAFSpec(D, T, hidden=H)
        if not spec.tri_ok:
            continue
        f = Flow(D, spec, seed=case)
        f.set_params(cases.flow_params(spec, case))
        z = torch.randn(n, D, generator=torch.Generator().manual_seed(case)) * 1.1
        out = {}
        for algo in (2, 6, 7):
            f.inverse_algo = algo
            out[algo] = [t.numpy() for t in f.inverse(z)]
        fin = np.isfinite(out[2][0]).all(axis=1) & np.isfinite(out[7][0]).all(axis=1) & np.isfinite(out[6][0]).all(axis=1)
        if fin.any():
            sc = np.maximum(1.0, np.abs(out[2][0][fin]).max(axis=1, keepdims=True))
            assert (np.abs(out[7][0][fin] - out[6][0][fin]) / sc).max() < 1e-4, (D, T, H, n)
            assert np.abs(out[7][1][fin] - out[6][1][fin]).max() < 1e-3 * max(1.0, float(np.abs(out[6][1][fin]).max())), (D, T, H, n)
            # (two float32 algorithms with different summation orders; up to seven transforms amplify the rounding of a
            #  stretched row: measured <= 6e-5 over these shapes)
            assert (np.abs(out[7][0][fin] - out[2][0][fin]) / sc).max() < 5e-4, (D, T, H, n)
        # ... and, on up to 32 rows, the float64 evaluation of the same parameters by the oracle: not device against device
        k = min(n, 32)
        x64, l64 = OracleMAF(spec, cases.flow_params(spec, case), dtype=np.float64).inverse(z[:k].numpy())
        ok = np.isfinite(x64).all(axis=1) & fin[:k] & (np.abs(x64).max(axis=1) < 1e30)
        if ok.any():
            e64 = float((np.abs(out[7][0][:k][ok] - x64[ok]) / np.maximum(1.0, np.abs(x64[ok]).max(axis=1, keepdims=True))).max())
            worst64 = max(worst64, e64)
            assert e64 < 1e-5, (D, T, H, n, e64)             # (the north star's 1e-5; measured 1.2e-6)
        done += 1
    print(f"sweeps on random shapes: two-wave sweep against the float64 oracle, worst row {worst64:.2e}")
    assert done >= 25
