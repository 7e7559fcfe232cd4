"""Same seed, separate processes: prints a digest of a short Sampler run (python scripts/repro_check.py precision [hidden])"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hashlib
import numpy as np
from scipy.stats import uniform
import pocomc_amd as pc
from pocomc_amd.maf_spec import MAFSpec

prec = sys.argv[1]
H = int(sys.argv[2]) if len(sys.argv) > 2 else 256
D = 16
rng = np.random.default_rng(5)
A = rng.normal(size=(D, D)) / np.sqrt(D) * 0.6 + np.eye(D)
icov = np.linalg.inv(A @ A.T)


def like(x):
    return -0.5 * np.einsum("ni,ij,nj->n", x, icov, x)


flow = pc.Flow(D, MAFSpec(D, 3, hidden=H), precision=prec, seed=1)
s = pc.Sampler(prior=pc.Prior(D * [uniform(-10.0, 20.0)]), likelihood=like, vectorize=True, n_effective=512, n_active=256,
               flow=flow, random_state=3, train_config=dict(epochs=30))
s.run()
x, w, _, _ = s.posterior()
print(prec, "logZ", s.evidence(), "iterations", len(s.particles.scalars["beta"]), "betas", np.round(s.particles.scalars["beta"][:6], 6),
      hashlib.sha1(np.ascontiguousarray(x).tobytes()).hexdigest()[:12], hashlib.sha1(flow.params.cpu().numpy().tobytes()).hexdigest()[:12])
