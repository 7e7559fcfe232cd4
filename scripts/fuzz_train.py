"""Random shapes through the training kernels (chain kernel + weight-gradient kernel) against torch autograd on the oracle
twin: (D, T, hidden, univariate, bins, rows, weighted, row gather, scratch cap).   python scripts/fuzz_train.py [cases] [seed]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np
import torch
import cases
from oracle.maf import torch_loss
from pocomc_amd import Flow
from pocomc_amd.maf_spec import MAFSpec
from pocomc_amd.train import loss_and_grad, _train_state

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = 0.0
for c in range(n_cases):
    uni = "rqs" if rng.random() < 0.4 else "affine"
    D = int(rng.integers(2, 40))
    T = int(rng.integers(1, 5))
    H = None if rng.random() < 0.4 else int(rng.integers(max(D - 1, 2), 3 * D + 40))
    bins = int(rng.choice([4, 8, 16])) if uni == "rqs" else 8
    n = int(rng.choice([1, 3, 16, 17, 100, 513, 700]))
    weighted, gather = rng.random() < 0.5, rng.random() < 0.3
    cap = int(rng.choice([0, 0, 4, 9]))
    spec = MAFSpec(D, T, H, univariate=uni, bins=bins)
    flat = cases.flow_params(spec, int(rng.integers(0, 100)), gain=1.0)
    f = Flow(D, spec)
    f.set_params(flat)
    ts = _train_state(f)
    if cap:
        ts.set_cap = cap
    ts.repack(f)
    x = (rng.normal(size=(n, D)) * 1.8).astype(np.float32)
    w = rng.uniform(0.1, 1.0, size=n).astype(np.float32) if weighted else None
    ft = torch.tensor(flat, requires_grad=True)
    lo = torch_loss(spec, ft, torch.from_numpy(x), None if w is None else torch.from_numpy(w))
    lo.backward()
    g_ref = ft.grad.numpy()
    xd = torch.from_numpy(x).cuda()
    wd = None if w is None else torch.from_numpy(w).cuda()
    if gather:
        perm = torch.randperm(n)
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(n)
        loss = float(loss_and_grad(f, xd[perm.cuda()].contiguous(), None if wd is None else wd[perm.cuda()].contiguous(), idx=inv.cuda()))
    else:
        loss = float(loss_and_grad(f, xd, wd))
    g = ts.grad.cpu().numpy()
    scale = np.abs(g_ref).max()
    err = np.abs(g - g_ref).max() / max(scale, 1e-30)
    lerr = abs(loss - float(lo.detach())) / max(abs(float(lo.detach())), 1e-30)
    worst = max(worst, err)
    ok = err < 2e-3 and lerr < 1e-4 and not g[spec.mask_flat() == 0].any() and np.isfinite(g).all()
    print(f"{'ok ' if ok else 'BAD'} D={D} T={T} H={spec.hidden} {uni}{bins if uni == 'rqs' else ''} n={n} w={int(weighted)} gather={int(gather)} cap={cap} "
          f"tri_ok={spec.tri_ok} jobs={ts.n_jobs}: grad err {err:.2e} loss err {lerr:.2e}")
    if not ok:
        sys.exit(1)
print("worst relative gradient error", worst)
