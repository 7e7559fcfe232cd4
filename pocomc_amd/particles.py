"""``Particles`` -- per-iteration history of the SMC run (``pocomc/particles.py``).

An append-only store keyed by quantity; ``compute_logw_and_logz`` (``particles.py:215-231``,
the O(T^2 N) persistent-sampling mixture weights evaluated ~10-20 times per beta bisection)
runs on the GPU through ``pocomc_amd.tools.compute_logw_and_logz``."""
from __future__ import annotations

import numpy as np

KEYS = ("u", "x", "logdetj", "logl", "logp", "logw", "blobs", "iter", "logz", "calls", "steps",
        "efficiency", "ess", "accept", "beta")


class Particles:
    def __init__(self, n_particles, n_dim):
        self.n_particles = n_particles
        self.n_dim = n_dim
        self.past = {k: [] for k in KEYS}
        self.results_dict = None

    def update(self, data):
        """Append every known key of ``data`` (``particles.py:93-148``)."""
        for key, value in data.items():
            if key in self.past:
                self.past[key].append(value)

    def pop(self, key):
        self.past[key].pop()

    def get(self, key, index=None, flat=False):
        """``particles.py:165-213``."""
        if index is not None:
            return self.past[key][index]
        return np.concatenate(self.past[key]) if flat else np.asarray(self.past[key])

    def compute_logw_and_logz(self, beta_final=1.0, normalize=True):
        from . import tools
        return tools.compute_logw_and_logz(self.get("logl"), self.get("beta"), self.get("logz"),
                                           beta_final, normalize)

    def pool_weights(self):
        """The history on the device for a series of trials at different ``beta_final`` (``tools.PoolWeights``)."""
        from . import tools
        return tools.PoolWeights(self.get("logl"), self.get("beta"), self.get("logz"))

    def compute_results(self):
        """``particles.py:233-302``."""
        if self.results_dict is None:
            self.results_dict = {k: self.get(k) for k in self.past}
            self.results_dict["logw"], _ = self.compute_logw_and_logz(1.0)
        return self.results_dict
