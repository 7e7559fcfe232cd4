"""Where the time of one mcmc kernel call goes at 1e4 x 32 (engine construction, state upload, steps, download)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scipy.stats import uniform
import pocomc_amd as pc
from pocomc_amd import mcmc as M
from pocomc_amd.geometry import Geometry
D, N = 32, 10000
prior = pc.Prior([uniform(-10, 20)] * D)
rng = np.random.default_rng(0)
scaler = pc.Reparameterize(D, bounds=prior.bounds); scaler.fit(rng.uniform(-10, 10, size=(4000, D)))
x = rng.uniform(-9, 9, size=(N, D)); u = scaler.forward(x)
like = lambda xx: (-0.5 * np.sum((xx / 3.0) ** 2, axis=1), None)
flow = pc.Flow(D, "maf3", seed=0)
geo = Geometry(); geo.fit(flow.forward(torch.from_numpy(u).float())[0].numpy().astype(np.float64))
state = dict(u=u, x=x, logdetj=scaler.inverse(u)[1], logl=like(x)[0], logp=prior.logpdf(x), beta=0.5, blobs=None)
funcs = dict(loglike=like, logprior=prior.logpdf, scaler=scaler, flow=flow, theta_geometry=geo)
for lanes in (1, 2):
    for n_max in (1, 21):
        opts = dict(n_max=n_max, n_steps=10 ** 6, progress_bar=None, proposal_scale=0.4, seed=1, x_order="F", lanes=lanes)
        M.preconditioned_pcn(state, funcs, opts)
        t0 = time.perf_counter()
        for _ in range(5):
            M.preconditioned_pcn(state, funcs, opts)
        print(f"lanes={lanes} n_max={n_max}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per kernel call")
