import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy.stats import uniform
import pocomc_amd as pc
D = 16
prior = pc.Prior(D * [uniform(-10.0, 20.0)])
def ll(x):
    return -np.sum(10.0 * (x[:, ::2] ** 2.0 - x[:, 1::2]) ** 2.0 + (x[:, ::2] - 1.0) ** 2.0, axis=1)
t0 = time.time()
s = pc.Sampler(prior=prior, likelihood=ll, vectorize=True, random_state=0, flow="maf6", n_active=4096, n_effective=8192,
               train_config=dict(epochs=100))
s.run(progress=False, n_total=8192, n_evidence=0)
print("wall", time.time() - t0, "iterations", s.t, "calls", s.calls)
