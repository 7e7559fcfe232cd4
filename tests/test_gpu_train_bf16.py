"""bf16 matrix-core training of the wide affine flows (``csrc/maf_train_bf16.hip``, BASELINE config 5) against autograd
on the float32 oracle twin of the flow (``oracle.maf.torch_loss``) and against the float32 training kernels.

Tolerance: weights, hidden activations and their gradients carry 8 mantissa bits (relative rounding 2^-9 = 2e-3 per
value); sums are float32.  The batch loss is compared at 1e-3 relative, the gradient at 2e-2 in the L2 norm overall
(1e-1 for any single tensor: the first layer of a small flow sees few rows to average the rounding over) with a cosine >= 0.9995 -- the float32 kernels hold 1e-3 / 5e-5 on the same inputs
(``test_gpu_train.py``)."""
import os
import socket

import numpy as np
import pytest
import torch

import cases
from oracle.maf import torch_loss
from pocomc_amd.maf_spec import MAFSpec

pytestmark = pytest.mark.gpu


def make(D, T, H, seed=2, gain=1.0):
    from pocomc_amd import Flow
    spec = MAFSpec(D, T, hidden=H)
    flat = cases.flow_params(spec, seed, gain=gain)
    f = Flow(D, spec, precision="bf16")
    f.train_engine = "bf16"
    f.set_params(flat)
    return f, spec, flat


@pytest.mark.parametrize("D,T,H,n,weighted", [(16, 2, 64, 100, True), (16, 2, 64, 1, False), (16, 2, 64, 31, True), (5, 3, 32, 33, False), (50, 6, 256, 300, False),
                                               (128, 8, 512, 512, False), (128, 8, 512, 700, True)])
def test_bf16_loss_and_gradient_match_autograd(D, T, H, n, weighted):
    from pocomc_amd.train import loss_and_grad, _train_state
    f, spec, flat = make(D, T, H)
    rng = np.random.default_rng(D + n)
    x = (rng.normal(size=(n, D)) * 1.2).astype(np.float32)
    w = rng.uniform(0.1, 1.0, size=n).astype(np.float32) if weighted else None
    ft = torch.tensor(flat, requires_grad=True)
    lo = torch_loss(spec, ft, torch.from_numpy(x), None if w is None else torch.from_numpy(w))
    lo.backward()
    g_ref = ft.grad.numpy().astype(np.float64)
    loss = float(loss_and_grad(f, torch.from_numpy(x).cuda(), None if w is None else torch.from_numpy(w).cuda()))
    g = _train_state(f).grad.cpu().numpy().astype(np.float64)
    assert abs(loss - float(lo.detach())) <= 1e-3 * abs(float(lo.detach()))
    rel = np.linalg.norm(g - g_ref) / np.linalg.norm(g_ref)
    cos = g @ g_ref / np.linalg.norm(g) / np.linalg.norm(g_ref)
    worst = 0.0
    for t in range(T):
        for name in spec.offsets:
            a, b = spec.view(g_ref, t, name), spec.view(g, t, name)
            worst = max(worst, np.linalg.norm(a - b) / np.linalg.norm(a))
    print(f"D={D} T={T} H={H} n={n}: loss rel {abs(loss - float(lo.detach())) / abs(float(lo.detach())):.1e}, gradient rel "
          f"{rel:.1e} (worst tensor {worst:.1e}), cosine {cos:.6f}")
    assert rel <= 2e-2 and worst <= 1e-1 and cos >= 0.9995
    # masked entries are never written
    assert np.all(g[spec.mask_flat() == 0] == 0.0)


def test_bf16_indexed_batch_is_the_contiguous_batch_bit_for_bit():
    from pocomc_amd.train import loss_and_grad, _train_state
    f, spec, flat = make(16, 2, 64)
    n = 700                                            # two chunks of <= 512 rows
    rng = np.random.default_rng(5)
    x = torch.from_numpy((rng.normal(size=(n, 16)) * 1.2).astype(np.float32)).cuda()
    w = torch.from_numpy(rng.uniform(0.1, 1.0, size=n).astype(np.float32)).cuda()
    loss = float(loss_and_grad(f, x, w))
    g = _train_state(f).grad.cpu().numpy().copy()
    perm = torch.randperm(n)
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(n)
    loss2 = float(loss_and_grad(f, x[perm.cuda()].contiguous(), w[perm.cuda()].contiguous(), idx=inv.cuda()))
    assert loss2 == loss
    assert np.array_equal(_train_state(f).grad.cpu().numpy(), g)
    # and a second call reproduces the first (fixed summation orders, no atomics)
    assert float(loss_and_grad(f, x, w)) == loss
    assert np.array_equal(_train_state(f).grad.cpu().numpy(), g)


@pytest.mark.parametrize("weighted", [False, True])
def test_bf16_epoch_call_equals_batch_by_batch(weighted):
    from pocomc_amd.train import loss_and_grad, AdamW
    rng = np.random.default_rng(3)
    n, D, bs = 1000, 16, 256
    x = torch.from_numpy((rng.normal(size=(n, D)) * 1.5).astype(np.float32)).cuda()
    w = torch.from_numpy(rng.uniform(0.1, 1.0, size=n).astype(np.float32)).cuda() if weighted else None
    perm = torch.from_numpy(rng.permutation(n)).cuda()

    def run(epoch_call):
        f, spec, flat = make(D, 3, 64)
        opt = AdamW(f, 2e-3, weight_decay=0.01)
        acc = torch.zeros(1, dtype=torch.float32, device="cuda")
        if epoch_call:
            opt.epoch(x, w, perm, bs, 1.0, acc)
        else:
            for b0 in range(0, n, bs):
                idx = perm[b0:b0 + bs].contiguous()
                acc += loss_and_grad(f, x, w, idx=idx, refresh=False)
                opt.step(1.0)
        return f.params.cpu().numpy(), opt.m.cpu().numpy(), opt.v.cpu().numpy(), float(acc), opt.t

    a, b, c = run(True), run(False), run(True)
    for u, v in zip(a, b):
        assert np.array_equal(u, v)
    for u, v in zip(a, c):
        assert np.array_equal(u, v)


def test_bf16_fit_follows_the_float32_fit():
    """Same data, same initial parameters, same batches: the bf16 engine's loss curve stays within 1 % of the float32
    engine's, and the fitted density is the same to a few 1e-2 nats per sample."""
    from pocomc_amd import Flow
    D = 24
    rng = np.random.default_rng(8)
    A = rng.normal(size=(D, D)) / np.sqrt(D) + np.eye(D)
    x = torch.from_numpy((rng.normal(size=(6000, D)) @ A.T + 0.5).astype(np.float32))
    spec = MAFSpec(D, 3, hidden=256)
    flat = spec.init_params(4)
    hist, lp = {}, {}
    for prec in ("f32", "bf16"):
        f = Flow(D, spec, precision=prec)
        f.set_params(flat)
        hist[prec] = f.fit(x, epochs=25, batch_size=512, validation_split=0.8, shuffle=False, annealing=False,
                           patience=100)
        lp[prec] = f.log_prob(x[:2000]).double().mean().item()
    l32, l16 = np.array(hist["f32"]["loss"]), np.array(hist["bf16"]["loss"])
    v32, v16 = np.array(hist["f32"]["val_loss"]), np.array(hist["bf16"]["val_loss"])
    print("train loss f32 / bf16:", l32[[0, 5, -1]], l16[[0, 5, -1]], " log_prob", lp)
    assert l32[-1] < l32[0] - 1.0                      # (the fit does learn something)
    np.testing.assert_allclose(l16, l32, rtol=1e-2)
    np.testing.assert_allclose(v16, v32, rtol=1e-2)
    assert abs(lp["bf16"] - lp["f32"]) < 0.05 * abs(lp["f32"]) / D + 0.1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data(n=1024, D=16):
    rng = np.random.default_rng(11)
    x = (rng.normal(size=(n, D)) * np.linspace(0.5, 2.0, D) + 0.3).astype(np.float32)
    w = rng.uniform(0.2, 1.0, size=n).astype(np.float32)
    return x, w


def _fit(x, w, sharded):
    from pocomc_amd import Flow
    spec = MAFSpec(x.shape[1], 3, hidden=64)
    f = Flow(x.shape[1], spec, precision="bf16")
    f.train_engine = "bf16"
    f.set_params(spec.init_params(5))
    hist = f.fit(torch.from_numpy(x), weights=torch.from_numpy(w), validation_split=0.75, epochs=4, batch_size=256,
                 shuffle=False, annealing=False, patience=100, **({} if sharded else {"sharded": False}))
    return f, hist


def _worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x, w = _data()
    B, lb = 256, 256 // world
    rows = np.concatenate([np.arange(b0 + rank * lb, b0 + (rank + 1) * lb) for b0 in range(0, len(x), B)])
    f, hist = _fit(x[rows], w[rows], True)
    if rank == 0:
        np.savez(out, params=f.params.cpu().numpy(), loss=np.array(hist["loss"]), val=np.array(hist["val_loss"]))
    dist.barrier()
    dist.destroy_process_group()


def test_bf16_two_ranks_equal_one_process(tmp_path):
    """Data-parallel bf16 training (weight-sum and gradient all-reduce as in ``test_gpu_sharded_train.py``): a row's
    activations do not depend on which rank holds it, so only the float32 order of the gradient sums differs."""
    import torch.multiprocessing as mp
    x, w = _data()
    out = str(tmp_path / "r0.npz")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    f, hist = _fit(x, w, False)
    np.testing.assert_allclose(got["loss"], hist["loss"], rtol=5e-5)
    np.testing.assert_allclose(got["val"], hist["val_loss"], rtol=5e-5)
    np.testing.assert_allclose(got["params"], f.params.cpu().numpy(), rtol=5e-3, atol=5e-5)
