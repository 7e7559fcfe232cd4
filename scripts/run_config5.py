"""BASELINE configs[4] in the small: 128-D funnel, 8-transform MAF (H = 512) on the bf16 matrix cores, one GPU.
    python scripts/run_config5.py [n_active] [precision] [seed] [sigma0]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy.stats import uniform

import pocomc_amd as pc
from pocomc_amd.maf_spec import MAFSpec

n_active = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
s0 = float(sys.argv[4]) if len(sys.argv) > 4 else 1.5
D = 128


def funnel(x):
    x0 = x[:, 0]
    return -x0 ** 2 / (2 * s0 ** 2) - 0.5 * np.sum(x[:, 1:] ** 2, axis=1) * np.exp(-x0) - 0.5 * (D - 1) * x0


target = os.environ.get("TARGET", "funnel")
if target == "gauss":
    rng = np.random.default_rng(5)
    A = rng.normal(size=(D, D)) / np.sqrt(D) * 0.6 + np.eye(D)
    cov = A @ A.T
    icov = np.linalg.inv(cov)
    mu = rng.normal(size=D)
    _, logdet = np.linalg.slogdet(cov)

    def funnel(x):                                        # (a correlated Gaussian instead)
        d = x - mu
        return -0.5 * np.einsum("ni,ij,nj->n", d, icov, d)

prior = pc.Prior(D * [uniform(-30.0, 60.0)])
flow = pc.Flow(D, MAFSpec(D, 8), precision=prec)
t0 = time.time()
s = pc.Sampler(prior=prior, likelihood=funnel, vectorize=True, n_effective=2 * n_active, n_active=n_active, flow=flow,
               random_state=seed, train_config=dict(epochs=int(os.environ.get("EPOCHS", 50))))
s.run()
dt = time.time() - t0
logz, err = s.evidence()
x, w, _, _ = s.posterior()
w = w / w.sum()
exact = np.log(s0) + 0.5 * D * np.log(2 * np.pi) - D * np.log(60.0)
if target == "gauss":
    exact = 0.5 * D * np.log(2 * np.pi) + 0.5 * logdet - D * np.log(60.0)
print(f"precision={prec} n_active={n_active} wall={dt:.1f}s iterations={len(s.particles.scalars['beta'])} "
      f"calls={s.results['calls'][-1] if hasattr(s, 'results') else -1} logZ={logz:.2f}+-{err:.2f} (analytic {exact:.2f}) "
      f"x0 mean {np.sum(w * x[:, 0]):.2f} std {np.sqrt(np.sum(w * x[:, 0] ** 2) - np.sum(w * x[:, 0]) ** 2):.2f}")
