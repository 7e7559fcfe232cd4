"""bf16 training engine against the float32 one on the same flow and batch:  python scripts/check_wide.py D T H n [weighted]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from pocomc_amd import Flow
from pocomc_amd.maf_spec import MAFSpec
from pocomc_amd.train import loss_and_grad, _train_state, _wide_state

D, T, H, n = (int(a) for a in sys.argv[1:5])
weighted = len(sys.argv) > 5
spec = MAFSpec(D, T, hidden=H)
rng = np.random.default_rng(1)
flat = spec.init_params(3)
x = torch.from_numpy(rng.normal(size=(n, D)).astype(np.float32) * 1.3).cuda()
w = torch.from_numpy(rng.uniform(0.1, 1.0, size=n).astype(np.float32)).cuda() if weighted else None
out = {}
for prec in ("f32", "bf16"):
    f = Flow(D, spec, precision=prec)
    f.set_params(flat)
    f.train_engine = prec
    if prec == 'f32':
        _train_state(f).repack(f)
    loss = float(loss_and_grad(f, x, w))
    g = _train_state(f).grad.cpu().numpy().astype(np.float64)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        loss_and_grad(f, x, w, refresh=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    out[prec] = (loss, g)
    print(f"{prec}: loss {loss:.6f}  |grad| {np.linalg.norm(g):.6f}  {dt * 1e6:.0f} us per call")
l0, g0 = out["f32"]
l1, g1 = out["bf16"]
print(f"loss rel {abs(l1 - l0) / abs(l0):.2e}   grad rel (L2) {np.linalg.norm(g1 - g0) / np.linalg.norm(g0):.2e}   "
      f"cos {g0 @ g1 / np.linalg.norm(g0) / np.linalg.norm(g1):.6f}   nonzero {np.count_nonzero(g0)} / {np.count_nonzero(g1)}")
m = spec.mask_flat()
for t in range(T):
    for name in spec.offsets:
        a, b = spec.view(g0, t, name), spec.view(g1, t, name)
        mk = spec.view(m, t, name)
        den = np.linalg.norm(a) + 1e-30
        print(f"t={t} {name}: unmasked {int(mk.sum())}  nonzero f32 {np.count_nonzero(a)} bf16 {np.count_nonzero(b)}  "
              f"rel {np.linalg.norm(a - b) / den:.2e}  |f32| {den:.3e} |bf16| {np.linalg.norm(b):.3e}")
