"""cProfile of the README example (BASELINE config 1): where a Sampler run spends its wall time.
    python scripts/profile_readme.py [flow]"""
import cProfile
import os
import pstats
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy.stats import uniform
import pocomc_amd as pc

n_dim = 10
prior = pc.Prior(n_dim * [uniform(-10.0, 20.0)])


def log_likelihood(x):
    return -np.sum(10.0 * (x[:, ::2] ** 2.0 - x[:, 1::2]) ** 2.0 + (x[:, ::2] - 1.0) ** 2.0, axis=1)


flow = sys.argv[1] if len(sys.argv) > 1 else "nsf6"
pc.Sampler(prior=prior, likelihood=log_likelihood, vectorize=True, random_state=1, flow=flow).run(progress=False)   # warm
s = pc.Sampler(prior=prior, likelihood=log_likelihood, vectorize=True, random_state=0, flow=flow)
pr = cProfile.Profile()
pr.enable()
s.run(progress=False)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(35)
