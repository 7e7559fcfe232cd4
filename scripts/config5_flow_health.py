"""Health of the config-5 flow the step benchmark trains (128-D, 8 transforms, H = 512, 5000 logit-transformed prior draws):
finite fraction and error of inverse(forward(u)) as a function of the epochs / rows / weight decay of the fit.

    python scripts/config5_flow_health.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pocomc_amd import Flow, Reparameterize
from pocomc_amd.maf_spec import MAFSpec
D, n = 128, 5000
lo, hi = -30.0, 30.0
bounds = np.array([[lo, hi]] * D)
fit_rng = np.random.default_rng(7)
x_fit = fit_rng.uniform(lo, hi, size=(2 * n, D))
scaler = Reparameterize(D, bounds=bounds)
scaler.fit(x_fit)
x = np.random.default_rng(1000).uniform(lo, hi, size=(n, D))
u = torch.from_numpy(scaler.forward(x)).float().cuda()
for label, rows, kw in (("50 epochs, 5000 rows (bench)", n, dict(epochs=50)), ("20 epochs", n, dict(epochs=20)), ("10 epochs", n, dict(epochs=10)),
                        ("5 epochs", n, dict(epochs=5)), ("50 epochs, 10000 rows", 2 * n, dict(epochs=50)),
                        ("50 epochs, l2 1e-3", n, dict(epochs=50, weight_decay=1e-3)), ("untrained", n, None)):
    torch.manual_seed(0)
    flow = Flow(D, MAFSpec(D, 8), seed=0)
    if kw is not None:
        u_fit = torch.from_numpy(scaler.forward(x_fit[:rows])).float().cuda()
        try:
            hist = flow.fit(u_fit, batch_size=512, validation_split=0.5, patience=D, annealing=False, verbose=0, **kw)
        except TypeError as e:
            print(label, "skipped:", e); continue
        tail = f"loss {hist['loss'][-1]:.2f} val {hist['val_loss'][-1]:.2f} epochs run {len(hist['loss'])}"
    else:
        tail = ""
    th, _ = flow.forward(u)
    ub, _ = flow.inverse(th)
    ok = torch.isfinite(ub).all(dim=1)
    err = ((ub - u).abs().max(dim=1).values / u.abs().max(dim=1).values)[ok]
    print(f"{label:32s} finite rows {int(ok.sum())}/{n}  round trip rel err median {float(err.median()) if ok.any() else float('nan'):.2e} "
          f"p99 {float(err.quantile(0.99)) if ok.any() else float('nan'):.2e}  |theta| max {float(th.abs().max()):.1f}  {tail}")
