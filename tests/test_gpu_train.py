"""Training kernels (``Flow.fit``, pocomc/flow.py:165-384) against the torch-autograd twin of the
oracle MAF, and the reference's own fit test (``tests/test_flow.py:168-193``)."""
import numpy as np
import pytest
import torch

import cases
from oracle.maf import torch_loss
from pocomc_amd.maf_spec import MAFSpec

pytestmark = pytest.mark.gpu


def make(D, T, seed=1, H=None):
    from pocomc_amd import Flow
    spec = MAFSpec(D, T, H)
    flat = cases.flow_params(spec, seed)
    f = Flow(D, spec)
    f.set_params(flat)
    return f, spec, flat


@pytest.mark.parametrize("D,T,H", [(2, 3, None), (4, 3, None), (10, 3, None), (32, 3, None), (7, 6, None), (10, 3, 100), (12, 2, 11)])
@pytest.mark.parametrize("n", [5, 16, 100])
@pytest.mark.parametrize("weighted", [False, True])
def test_loss_and_gradient_match_autograd(D, T, H, n, weighted):
    from pocomc_amd.train import loss_and_grad, _train_state
    f, spec, flat = make(D, T, H=H)
    rng = np.random.default_rng(n + D)
    x = (rng.normal(size=(n, D)) * 1.3).astype(np.float32)
    w = rng.uniform(0.1, 1.0, size=n).astype(np.float32) if weighted else None
    ft = torch.tensor(flat, requires_grad=True)
    lo = torch_loss(spec, ft, torch.from_numpy(x), None if w is None else torch.from_numpy(w))
    lo.backward()
    g_ref = ft.grad.numpy()
    _train_state(f).repack(f)
    loss = loss_and_grad(f, torch.from_numpy(x).cuda(), None if w is None else torch.from_numpy(w).cuda())
    g = f._train.grad.cpu().numpy()
    np.testing.assert_allclose(float(loss), float(lo.detach()), rtol=2e-5)
    scale = np.abs(g_ref).max()
    np.testing.assert_allclose(g, g_ref, rtol=2e-4, atol=2e-5 * scale)
    # masked weights never receive gradient
    assert not g[spec.mask_flat() == 0].any()


def test_adamw_with_clipping_matches_torch():
    """Optimizer arithmetic (clip_grad_norm_ + AdamW, flow.py:268,:318-319) on identical gradients:
    Adam normalises every coordinate by its own gradient scale, so rounding-noise gradients would
    otherwise be amplified to O(lr) differences that say nothing about the kernels."""
    from pocomc_amd.train import AdamW, _train_state
    f, spec, flat = make(6, 3)
    rng = np.random.default_rng(0)
    x = torch.from_numpy((rng.normal(size=(64, 6)) * 1.5).astype(np.float32))
    ft = torch.tensor(flat, requires_grad=True)
    ref = torch.optim.AdamW([ft], 1e-2, weight_decay=0.01)
    opt = AdamW(f, 1e-2, weight_decay=0.01)
    ts = _train_state(f)
    for _ in range(4):
        ref.zero_grad()
        torch_loss(spec, ft, x).backward()
        ts.grad.copy_(ft.grad)                       # same gradient into both optimizers
        torch.nn.utils.clip_grad_norm_([ft], 1.0)
        ref.step()
        opt.step(1.0)
        np.testing.assert_allclose(f.params.cpu().numpy(), ft.detach().numpy(), rtol=1e-5, atol=1e-6)


def test_training_trajectory_tracks_autograd():
    """A few full steps (device gradients + device optimizer) against torch: the loss curves agree."""
    from pocomc_amd.train import loss_and_grad, AdamW, _train_state
    f, spec, flat = make(6, 3)
    rng = np.random.default_rng(0)
    x = torch.from_numpy((rng.normal(size=(256, 6)) * 1.5).astype(np.float32))
    ft = torch.tensor(flat, requires_grad=True)
    ref = torch.optim.AdamW([ft], 1e-3)
    opt = AdamW(f, 1e-3)
    _train_state(f).repack(f)
    for _ in range(10):
        ref.zero_grad()
        lo = torch_loss(spec, ft, x)
        lo.backward()
        torch.nn.utils.clip_grad_norm_([ft], 1.0)
        ref.step()
        loss = loss_and_grad(f, x.cuda())
        opt.step(1.0)
        np.testing.assert_allclose(float(loss), float(lo.detach()), rtol=1e-4)


def test_reference_fit_smoke():
    """tests/test_flow.py:168-193: fit for a few epochs, then everything is finite."""
    from pocomc_amd import Flow
    torch.manual_seed(0)
    data = torch.randn(size=(100, 4)) * 1.5
    flow = Flow(n_dim=4, flow="maf3")
    hist = flow.fit(data, epochs=5)
    assert len(hist["loss"]) == 5 and np.isfinite(hist["loss"]).all()
    z, l = flow.forward(data)
    x, li = flow.inverse(z)
    assert torch.isfinite(z).all() and torch.isfinite(x).all() and torch.isfinite(flow.log_prob(data)).all()
    xs, lq = flow.sample(100)
    assert torch.isfinite(xs).all() and torch.isfinite(lq).all()


def test_fit_learns_a_scaled_gaussian():
    """Sampler-style call (sampler.py:655-669): weighted, validation_split=0.5, early stopping."""
    from pocomc_amd import Flow
    torch.manual_seed(1)
    D = 5
    data = torch.randn(2000, D) * torch.tensor([0.5, 1.0, 2.0, 3.0, 0.2]) + 1.0
    w = torch.ones(2000) / 2000
    flow = Flow(D, "maf3", seed=3)
    lp0 = flow.log_prob(data).mean().item()
    hist = flow.fit(data, weights=w, validation_split=0.5, epochs=60, batch_size=512, patience=D, annealing=False,
                    clip_grad_norm=1.0)
    lp1 = flow.log_prob(data).mean().item()
    assert hist["val_loss"][-1] < hist["val_loss"][0]
    assert lp1 > lp0 + 0.5
    # analytic optimum: mean log-density of the data under the true Gaussian
    true_lp = (-0.5 * D * np.log(2 * np.pi) - np.log([0.5, 1.0, 2.0, 3.0, 0.2]).sum() - 0.5 * D)
    assert lp1 > true_lp - 1.0


@pytest.mark.parametrize("D,T,H,n", [(50, 6, 256, 40), (128, 8, 512, 24), (64, 3, 256, 33)])
def test_wide_flows_fit_in_lds(D, T, H, n):
    """BASELINE configs 3 and 5 (D=50 maf6 H=256; D=128, 8 transforms, H=512): the aliased LDS plan
    of the loss/gradient kernel holds them; gradient against autograd."""
    from pocomc_amd import Flow
    from pocomc_amd.train import loss_and_grad, _train_state
    spec = MAFSpec(D, T, H)
    flat = cases.flow_params(spec, 2, gain=1.0)
    f = Flow(D, spec)
    f.set_params(flat)
    rng = np.random.default_rng(D)
    x = rng.normal(size=(n, D)).astype(np.float32)
    ft = torch.tensor(flat, requires_grad=True)
    lo = torch_loss(spec, ft, torch.from_numpy(x))
    lo.backward()
    g_ref = ft.grad.numpy()
    _train_state(f).repack(f)
    loss = loss_and_grad(f, torch.from_numpy(x).cuda())
    g = f._train.grad.cpu().numpy()
    np.testing.assert_allclose(float(loss), float(lo.detach()), rtol=5e-5)
    scale = np.abs(g_ref).max()
    np.testing.assert_allclose(g, g_ref, rtol=1e-3, atol=5e-5 * scale)


@pytest.mark.parametrize("cap", [None, 40])
def test_large_batch_loops_over_row_sets(cap):
    """A batch of many row sets; ``cap = 40``: more row sets (131) than the scratch arrays hold, so the batch comes in four
    chunks -- chain kernel + weight-gradient kernel per chunk, the later chunks adding to the gradient in place.  Indexed and
    contiguous batches agree bit for bit either way."""
    from pocomc_amd.train import loss_and_grad, _train_state
    f, spec, flat = make(10, 3)
    if cap is not None:
        ts = _train_state(f)
        ts.set_cap = cap
        assert ts.n_sets == 0
    n = 64 * 16 * 2 + 37
    rng = np.random.default_rng(5)
    x = (rng.normal(size=(n, 10)) * 1.2).astype(np.float32)
    w = rng.uniform(0.1, 1.0, size=n).astype(np.float32)
    ft = torch.tensor(flat, requires_grad=True)
    lo = torch_loss(spec, ft, torch.from_numpy(x), torch.from_numpy(w))
    lo.backward()
    g_ref = ft.grad.numpy()
    _train_state(f).repack(f)
    xd, wd = torch.from_numpy(x).cuda(), torch.from_numpy(w).cuda()
    loss = float(loss_and_grad(f, xd, wd))
    g = f._train.grad.cpu().numpy().copy()
    np.testing.assert_allclose(loss, float(lo.detach()), rtol=5e-5)
    np.testing.assert_allclose(g, g_ref, rtol=5e-4, atol=5e-5 * np.abs(g_ref).max())
    # the same batch through a row-index list into a shuffled copy of the data
    perm = torch.randperm(n)
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(n)
    loss2 = float(loss_and_grad(f, xd[perm.cuda()].contiguous(), wd[perm.cuda()].contiguous(), idx=inv.cuda()))
    assert loss2 == loss
    assert np.array_equal(f._train.grad.cpu().numpy(), g)
    assert f._train.desc.max_sets == (cap or (n + 15) // 16)


@pytest.mark.parametrize("weighted", [False, True])
def test_epoch_call_equals_batch_by_batch(weighted):
    """pmc_maf_train_epoch enqueues exactly the per-batch sequence (gradient, clip, AdamW, repack):
    bitwise the same parameters, optimizer state and accumulated loss; a second run reproduces
    the first bit for bit (no atomics anywhere in the training path)."""
    from pocomc_amd.train import loss_and_grad, AdamW, _train_state
    rng = np.random.default_rng(3)
    n, D, bs = 1000, 6, 256
    x = torch.from_numpy((rng.normal(size=(n, D)) * 1.5).astype(np.float32)).cuda()
    w = torch.from_numpy(rng.uniform(0.1, 1.0, size=n).astype(np.float32)).cuda() if weighted else None
    perm = torch.from_numpy(rng.permutation(n)).cuda()

    def run(epoch_call):
        f, spec, flat = make(D, 3)
        opt = AdamW(f, 2e-3, weight_decay=0.01)
        _train_state(f).repack(f)
        acc = torch.zeros(1, dtype=torch.float32, device="cuda")
        if epoch_call:
            opt.epoch(x, w, perm, bs, 1.0, acc)
        else:
            for b0 in range(0, n, bs):
                idx = perm[b0:b0 + bs].contiguous()
                acc += loss_and_grad(f, x, w, idx=idx)
                opt.step(1.0)
        return f.params.cpu().numpy(), opt.m.cpu().numpy(), opt.v.cpu().numpy(), float(acc), opt.t

    a, b, c = run(True), run(False), run(True)
    assert a[4] == b[4] == 4
    for u, v in zip(a[:3], c[:3]):
        assert np.array_equal(u, v)
    assert a[3] == c[3]
    # per-batch path: pmc_adamw_step recomputes the squared norm with a different (fixed) blocking
    for u, v in zip(a[:3], b[:3]):
        np.testing.assert_allclose(u, v, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(a[3], b[3], rtol=1e-6)


# ----------------------------------------------------------------- neural spline flows
@pytest.mark.parametrize("D,T,H,bins", [(2, 3, None, 8), (4, 3, None, 8), (10, 3, None, 8), (17, 2, None, 8), (32, 3, None, 8),
                                        (50, 6, None, 8), (4, 3, None, 4), (10, 3, 100, 4), (20, 2, None, 4), (4, 3, None, 16),
                                        (10, 2, 100, 16), (20, 2, None, 16), (10, 3, 100, 8)])
@pytest.mark.parametrize("n,weighted", [(5, False), (40, True)])
def test_nsf_loss_and_gradient_match_autograd(D, T, H, bins, n, weighted):
    """Spline flows: hyper-network gradients through the rational-quadratic spline (knots via
    softmax + cumsum, bin selection, end-knot derivatives) against torch autograd on the oracle
    twin; data spread beyond the spline box so that the identity tails are exercised too."""
    from pocomc_amd import Flow
    from pocomc_amd.train import loss_and_grad, _train_state
    spec = MAFSpec(D, T, H, univariate="rqs", bins=bins)          # (bins 4 / 16, odd hidden widths: flow.py:87-88)
    flat = cases.flow_params(spec, 4, gain=1.0)
    f = Flow(D, spec)
    f.set_params(flat)
    rng = np.random.default_rng(n + D)
    x = (rng.normal(size=(n, D)) * 2.5).astype(np.float32)
    w = rng.uniform(0.1, 1.0, size=n).astype(np.float32) if weighted else None
    ft = torch.tensor(flat, requires_grad=True)
    lo = torch_loss(spec, ft, torch.from_numpy(x), None if w is None else torch.from_numpy(w))
    lo.backward()
    g_ref = ft.grad.numpy()
    _train_state(f).repack(f)
    loss = loss_and_grad(f, torch.from_numpy(x).cuda(), None if w is None else torch.from_numpy(w).cuda())
    g = f._train.grad.cpu().numpy()
    np.testing.assert_allclose(float(loss), float(lo.detach()), rtol=5e-5)
    scale = np.abs(g_ref).max()
    np.testing.assert_allclose(g, g_ref, rtol=2e-3, atol=1e-4 * scale)
    assert not g[spec.mask_flat() == 0].any()


def test_nsf_fit_learns_a_bimodal_density():
    """The reference's default flow family (sampler.py:169 'nsf6'): fit nsf3 to a two-component
    mixture the affine flow cannot represent well; the spline flow's log-likelihood must beat the
    best single Gaussian by a clear margin."""
    from pocomc_amd import Flow
    torch.manual_seed(2)
    n, D = 4000, 2
    comp = torch.randint(0, 2, (n, 1)).float()
    data = torch.randn(n, D) * 0.5 + (2.0 * comp - 1.0) * 2.0
    flow = Flow(D, "nsf3", seed=1)
    hist = flow.fit(data, validation_split=0.8, epochs=150, batch_size=512, patience=30, learning_rate=3e-3)
    assert np.isfinite(hist["loss"]).all()
    lp = flow.log_prob(data).mean().item()
    cov = np.cov(data.numpy().T)
    gauss_lp = -0.5 * D * np.log(2 * np.pi) - 0.5 * np.linalg.slogdet(cov)[1] - 0.5 * D
    true_lp = -(np.log(2.0) + D * (0.5 * np.log(2 * np.pi) + np.log(0.5)) + 0.5 * D)
    assert lp > gauss_lp + 0.3, (lp, gauss_lp, true_lp)
    assert lp > true_lp - 0.25, (lp, true_lp)
    x, lq = flow.sample(2000)
    assert torch.isfinite(x).all() and torch.isfinite(lq).all()
    # both modes are populated by the samples
    frac = (x[:, 0] > 0).float().mean().item()
    assert 0.3 < frac < 0.7


@pytest.mark.parametrize("weighted", [False, True])
@pytest.mark.parametrize("name", ["maf3", "nsf3"])
def test_validation_epoch_call_equals_batch_by_batch(name, weighted):
    """pmc_maf_valid_epoch (one call per validation pass, rows gathered through the permutation inside the forward
    kernel) against the per-batch path (torch gather + forward + weighted sum) on the same batches."""
    import ctypes as C
    from pocomc_amd import Flow, _lib
    from pocomc_amd.train import batch_loss, _train_state
    rng = np.random.default_rng(8)
    n, D, bs = 1000, 6, 256
    f = Flow(D, name, seed=2)
    x = torch.from_numpy((rng.normal(size=(n, D)) * 1.5).astype(np.float32)).cuda()
    w = torch.from_numpy(rng.uniform(0.1, 1.0, size=n).astype(np.float32)).cuda() if weighted else None
    perm = torch.from_numpy(rng.permutation(n)).cuda()
    ref = 0.0
    for b0 in range(0, n, bs):
        idx = perm[b0:b0 + bs]
        ref += float(batch_loss(f, x[idx].contiguous(), None if w is None else w[idx].contiguous()))
    acc = torch.zeros(1, dtype=torch.float32, device="cuda")
    scratch = torch.empty(n, dtype=torch.float32, device="cuda")
    _lib.check(f.lib.pmc_maf_valid_epoch(C.byref(f._desc), _lib.ptr(x), _lib.ptr(w) if w is not None else None,
                                         _lib.ptr(perm), n, bs, _lib.ptr(scratch), _lib.ptr(acc), _lib.stream_handle()))
    np.testing.assert_allclose(float(acc), ref, rtol=2e-6)
    # consecutive rows (no permutation)
    acc.zero_()
    _lib.check(f.lib.pmc_maf_valid_epoch(C.byref(f._desc), _lib.ptr(x), _lib.ptr(w) if w is not None else None, None,
                                         n, bs, _lib.ptr(scratch), _lib.ptr(acc), _lib.stream_handle()))
    ref2 = sum(float(batch_loss(f, x[b0:b0 + bs].contiguous(), None if w is None else w[b0:b0 + bs].contiguous()))
               for b0 in range(0, n, bs))
    np.testing.assert_allclose(float(acc), ref2, rtol=2e-6)


# ---------------------------------------------------------------------------------------------- Flow.fit options
def test_weight_penalty_value_and_gradient():
    """pmc_weight_penalty: R = sum |W| / b + W^2 / (2 s^2) over the weight matrices (flow.py:387-421 as documented) and
    dR/dparams added to a gradient; biases untouched."""
    import ctypes as C
    from pocomc_amd import Flow, _lib
    from pocomc_amd.train import _weight_flags, _train_state
    f = Flow(5, "maf3", seed=2)
    flags = _weight_flags(f)
    p = f.params
    lib = f.lib
    ts = _train_state(f)
    grad = torch.full_like(p, 0.25)
    loss = torch.zeros(1, dtype=torch.float32, device="cuda")
    b, s = 0.7, 1.3
    _lib.check(lib.pmc_weight_penalty(_lib.ptr(p), _lib.ptr(flags), _lib.ptr(grad), p.numel(), b, s, 2.0, _lib.ptr(loss),
                                      _lib.ptr(ts.sq_partial), _lib.stream_handle()))
    pw = p.cpu().numpy().astype(np.float64)
    fl = flags.cpu().numpy().astype(bool)
    R = np.sum(np.abs(pw[fl]) / b + pw[fl] ** 2 / (2 * s * s))
    np.testing.assert_allclose(loss.item(), 2.0 * R, rtol=2e-6)
    want = np.full(pw.shape, 0.25)
    want[fl] += np.sign(pw[fl]) / b + pw[fl] / (s * s)
    np.testing.assert_allclose(grad.cpu().numpy(), want, rtol=1e-6, atol=1e-7)
    # every weight matrix entry is flagged, no bias is
    spec = f.spec
    n_w = spec.n_transforms * sum(spec.offsets[k][1] for k in ("W0", "W1", "W2", "W3"))
    assert fl.sum() == n_w


def test_fit_with_regularisation_shrinks_the_weights_and_reports_the_penalised_loss():
    from pocomc_amd import Flow
    torch.manual_seed(0)
    x = torch.randn(600, 4) * torch.tensor([1.0, 0.3, 2.0, 0.7]) + 0.5
    norms = {}
    for key, kw in (("plain", {}), ("l1", dict(laplace_scale=0.05)), ("l2", dict(gaussian_scale=0.2)),
                    ("both", dict(laplace_scale=0.05, gaussian_scale=0.2))):
        f = Flow(4, "maf3", seed=1)
        torch.manual_seed(1)
        h = f.fit(x, epochs=30, batch_size=200, validation_split=0.8, patience=1000, annealing=False, **kw)
        assert np.isfinite(h["loss"]).all() and np.isfinite(h["val_loss"]).all() and len(h["loss"]) == 30
        from pocomc_amd.train import _weight_flags
        fl = _weight_flags(f).cpu().numpy().astype(bool)
        norms[key] = (float(np.abs(f.params.cpu().numpy()[fl]).sum()), h["loss"][-1])
    assert norms["l1"][0] < 0.8 * norms["plain"][0] and norms["both"][0] < norms["l2"][0] < norms["plain"][0]
    assert norms["l1"][1] > norms["plain"][1]                     # the reported loss carries the penalty (flow.py:314-321)


def test_fit_with_noise_augmentation():
    """flow.py:240-245, :304-307: Gaussian noise of scale noise * mean_j |x_last - x_j| on every pass; the fit still learns
    (a broadened density), and the noise kernel has the stated law."""
    import ctypes as C
    from pocomc_amd import Flow, _lib
    from scipy import stats
    lib = _lib.load()
    x = torch.randn(4000, 6, device="cuda")
    out = torch.empty_like(x)
    _lib.check(lib.pmc_add_noise_f32(_lib.ptr(x), 4000, 6, 0.5, 77, 3, _lib.ptr(out), _lib.stream_handle()))
    d = ((out - x) / 0.5).cpu().numpy().ravel()
    assert stats.kstest(d, "norm").pvalue > 1e-3
    out2 = torch.empty_like(x)
    _lib.check(lib.pmc_add_noise_f32(_lib.ptr(x), 4000, 6, 0.5, 77, 4, _lib.ptr(out2), _lib.stream_handle()))
    assert not torch.equal(out, out2)                              # a fresh draw per pass
    md = torch.zeros(1, device="cuda")
    _lib.check(lib.pmc_mean_distance_f32(_lib.ptr(x), 4000, 6, 3999, _lib.ptr(md), _lib.stream_handle()))
    ref = torch.linalg.norm(x[-1] - x, dim=1).mean().item()        # flow.py:243-245: mean of the LAST row's distances
    np.testing.assert_allclose(md.item(), ref, rtol=1e-5)
    torch.manual_seed(0)
    data = torch.randn(800, 3) * 0.5 + 1.0
    f = Flow(3, "maf3", seed=0)
    h = f.fit(data, epochs=40, batch_size=200, validation_split=0.8, noise=0.2, annealing=False, patience=1000)
    assert np.isfinite(h["loss"]).all() and h["loss"][-1] < h["loss"][0]
    with pytest.raises(RuntimeError):
        Flow(3, "maf3", seed=0).fit(torch.randn(1, 3), epochs=1, noise=0.1)
