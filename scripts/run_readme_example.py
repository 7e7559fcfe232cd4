"""BASELINE config 1: the reference's README example (10-D Rosenbrock, uniform prior, default Sampler) end to end."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy.stats import uniform
import pocomc_amd as pc

n_dim = 10
prior = pc.Prior(n_dim * [uniform(-10.0, 20.0)])


def log_likelihood(x):
    return -np.sum(10.0 * (x[:, ::2] ** 2.0 - x[:, 1::2]) ** 2.0 + (x[:, ::2] - 1.0) ** 2.0, axis=1)


flow = sys.argv[1] if len(sys.argv) > 1 else "nsf6"
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
t0 = time.time()
sampler = pc.Sampler(prior=prior, likelihood=log_likelihood, vectorize=True, random_state=seed, flow=flow)
sampler.run(progress=False)
dt = time.time() - t0
samples, weights, logl, logp = sampler.posterior()
logz, logz_err = sampler.evidence()
print(f"flow={flow} wall={dt:.1f}s iterations={sampler.t} likelihood calls={sampler.calls} logZ={logz:.3f}+-{logz_err:.3f} "
      f"posterior mean x0,x1={np.average(samples[:, :2], weights=weights, axis=0)}")
