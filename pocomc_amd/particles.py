"""``Particles`` -- the persistent particle pool of the SMC run, RESIDENT IN HBM.

The reference keeps one numpy array per quantity and iteration in Python lists (``pocomc/particles.py:73-148``) and
concatenates them whenever the pool is needed (``:165-213``): every reweight, trim and resample of
``pocomc/sampler.py:717-805`` walks the whole history on the host.  Here the row quantities
(``u, x: (P, D)``; ``logdetj, logl, logp: (P,)``, float64) live in growing device buffers, one block of ``n_active``
rows per iteration; the kernels of ``csrc/mcmc_kernels.hip`` / ``csrc/pool.hip`` read them in place:

* ``compute_logw_and_logz`` (``particles.py:215-231``): ``pmc_logw`` + ``pmc_logw_stats`` on the resident ``logl``;
* importance weights, trimming, resampling indices and the row gather of the next walkers: ``select`` / ``take``.

Per-iteration scalars (``beta, logz, calls, ...``) stay host lists -- they are a handful of floats.  ``get`` keeps the
reference's accessor (numpy out) for results / posterior / tests.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib

ROW_KEYS = ("u", "x", "logdetj", "logl", "logp")
SCALAR_KEYS = ("iter", "logz", "calls", "steps", "efficiency", "ess", "accept", "beta")
KEYS = ROW_KEYS + ("logw", "blobs") + SCALAR_KEYS


class Particles:
    def __init__(self, n_particles, n_dim, device=None):
        self.n_particles = int(n_particles)               # rows appended per iteration (n_active)
        self.n_dim = int(n_dim)
        self.device = torch.device(device) if device is not None else _lib.require_gpu()
        self.lib = _lib.load()
        self.T = 0                                        # iterations held
        self._cap = 0
        self._rows = {}
        self._grow(32)
        self.scalars = {k: [] for k in SCALAR_KEYS}
        self.blobs = []                                   # host arrays (arbitrary dtype), one per iteration, or None
        self.results_dict = None
        self._stats = torch.zeros(4, dtype=torch.float64, device=self.device)
        self._h_stats = torch.zeros(4, dtype=torch.float64).pin_memory()
        self._ws = None
        self._lw = None

    # ------------------------------------------------------------------ storage
    def _grow(self, cap_iters):
        N, D, dev = self.n_particles, self.n_dim, self.device
        new = {k: torch.empty((cap_iters * N, D) if k in ("u", "x") else (cap_iters * N,), dtype=torch.float64, device=dev)
               for k in ROW_KEYS}
        for k, old in self._rows.items():
            new[k][:self.T * N].copy_(old[:self.T * N])
        self._rows, self._cap = new, cap_iters

    @property
    def P(self):
        return self.T * self.n_particles

    def rows(self, key):
        """Device view of a row quantity over the whole pool."""
        return self._rows[key][:self.P]

    def update(self, data):
        """Append one iteration (``particles.py:93-148``): row quantities as device tensors or numpy arrays of
        ``n_particles`` rows, scalars as numbers."""
        N = self.n_particles
        if self.T == self._cap:
            self._grow(2 * self._cap)
        lo = self.T * N
        for k in ROW_KEYS:
            v = data[k]
            t = v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v, dtype=np.float64))
            if t.shape[0] != N:
                raise ValueError(f"{k}: expected {N} rows, got {t.shape[0]}")
            self._rows[k][lo:lo + N].copy_(t.to(torch.float64), non_blocking=True)
        for k in SCALAR_KEYS:
            if k in data:
                self.scalars[k].append(data[k])
        self.blobs.append(data.get("blobs"))
        self.T += 1
        self.results_dict = None

    def pop(self, key):
        self.scalars[key].pop()

    def get(self, key, index=None, flat=False):
        """``particles.py:165-213`` (numpy out; row quantities are downloaded)."""
        N = self.n_particles
        if key in SCALAR_KEYS:
            return self.scalars[key][index] if index is not None else np.asarray(self.scalars[key])
        if key == "blobs":
            if index is not None:
                return self.blobs[index]
            return np.concatenate(self.blobs) if flat else np.asarray(self.blobs)
        if key == "logw":
            # (the reference keeps an empty list under this key and fills ``results["logw"]`` from
            #  compute_logw_and_logz(1.0), particles.py:297-299: the computed log-weights are what a caller can want)
            lw = self.compute_logw_and_logz(1.0)[0]
            if index is not None:
                return lw.reshape(self.T, N)[index]
            return lw if flat else lw.reshape(self.T, N)
        a = self.rows(key).cpu().numpy()
        if index is not None:
            return a.reshape((self.T, N) + a.shape[1:])[index]
        return a if flat else a.reshape((self.T, N) + a.shape[1:])

    # ------------------------------------------------------------ log-weights
    def _history(self):
        up = lambda v: torch.tensor(np.asarray(v, dtype=np.float64), dtype=torch.float64, device=self.device)
        return up(self.scalars["beta"]), up(self.scalars["logz"])

    def logw_stats(self, beta_final, k=0, history=None):
        """Mixture log-weights of the pool at ``beta_final`` (``particles.py:215-231``) into the resident buffer, and
        ``[max, sum exp(logw - max), sum exp(2 (logw - max)), sum 1 - (1 - w)^k]`` back (four doubles)."""
        P, lib = self.P, self.lib
        if self._lw is None or self._lw.numel() < P:
            self._lw = torch.empty(self._cap * self.n_particles, dtype=torch.float64, device=self.device)
            self._ws = torch.empty(int(lib.pmc_reduce_workspace_bytes(self._lw.numel())), dtype=torch.uint8, device=self.device)
        bd, zd = history if history is not None else self._history()
        with torch.cuda.device(self.device):
            st = _lib.stream_handle()
            _lib.check(lib.pmc_logw(_lib.ptr(self._rows["logl"]), _lib.ptr(bd), _lib.ptr(zd), float(beta_final),
                                    _lib.ptr(self._lw), self.T, self.n_particles, st), "pmc_logw")
            _lib.check(lib.pmc_logw_stats(_lib.ptr(self._lw), P, int(k), _lib.ptr(self._stats), _lib.ptr(self._ws), st),
                       "pmc_logw_stats")
            self._h_stats.copy_(self._stats, non_blocking=True)
            torch.cuda.current_stream().synchronize()
        return self._h_stats.numpy().copy()

    def compute_logw_and_logz(self, beta_final=1.0, normalize=True):
        """``particles.py:215-231``: host log-weights and logZ."""
        st = self.logw_stats(beta_final)
        lse = st[0] + np.log(st[1])
        logw = self._lw[:self.P].cpu().numpy()
        if normalize:
            logw -= lse
        return logw, lse - np.log(self.P)

    # ------------------------------------------------- weights, trimming, gather
    def select(self, ess=0.99, bins=1000):
        """After :meth:`logw_stats`: importance weights ``exp(logw - max) / sum`` (``sampler.py:779-781``) and
        ``trim_weights`` (``tools.py:10-53``) on the device.  Returns ``(weights, idx, w_trimmed)``: the normalised
        weights of the whole pool, the indices of the rows that survive the trimming (in pool order) and their
        renormalised weights -- device tensors; only the survivor count crosses to the host."""
        lib, P, dev = self.lib, self.P, self.device
        w = torch.empty(P, dtype=torch.float64, device=dev)
        idx = torch.empty(P, dtype=torch.int64, device=dev)
        wt = torch.empty(P, dtype=torch.float64, device=dev)
        cnt = torch.zeros(1, dtype=torch.int64, device=dev)
        res = torch.zeros(2, dtype=torch.float64, device=dev)
        nb1, nb2 = int(lib.pmc_trim_workspace_bytes(P)), int(lib.pmc_trim_select_workspace_bytes(P))
        ws = torch.empty(max(nb1, nb2), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            st = _lib.stream_handle()
            _lib.check(lib.pmc_weights_from_logw(_lib.ptr(self._lw), P, _lib.ptr(self._stats), _lib.ptr(w), st),
                       "pmc_weights_from_logw")
            _lib.check(lib.pmc_trim_threshold(_lib.ptr(w), P, float(ess), int(bins), _lib.ptr(res), _lib.ptr(ws), nb1, st),
                       "pmc_trim_threshold")
            _lib.check(lib.pmc_trim_select(_lib.ptr(w), P, _lib.ptr(res), _lib.ptr(idx), _lib.ptr(wt), _lib.ptr(cnt),
                                           _lib.ptr(ws), nb2, st), "pmc_trim_select")
        m = int(cnt.item())
        return w, idx[:m], wt[:m]

    def take(self, idx):
        """Rows ``idx`` (device int64) of the five row quantities (``sampler.py:707-713``, ``pmc_gather``): device tensors."""
        lib, dev, D = self.lib, self.device, self.n_dim
        n = int(idx.numel())
        out = {k: torch.empty((n, D) if k in ("u", "x") else (n,), dtype=torch.float64, device=dev) for k in ROW_KEYS}
        r = self._rows
        with torch.cuda.device(dev):
            _lib.check(lib.pmc_gather(_lib.ptr(idx), n, D, _lib.ptr(r["u"]), _lib.ptr(r["x"]), _lib.ptr(r["logdetj"]),
                                      _lib.ptr(r["logl"]), _lib.ptr(r["logp"]), _lib.ptr(out["u"]), _lib.ptr(out["x"]),
                                      _lib.ptr(out["logdetj"]), _lib.ptr(out["logl"]), _lib.ptr(out["logp"]),
                                      _lib.stream_handle()), "pmc_gather")
        return out

    # ---------------------------------------------------------------- results
    def compute_results(self):
        """``particles.py:233-302``."""
        if self.results_dict is None:
            self.results_dict = {k: self.get(k) for k in ROW_KEYS + SCALAR_KEYS}
            # every key of the reference's dict (particles.py:285-302): blobs too -- the stored blocks when a likelihood
            # returned any, None otherwise
            has_blobs = any(b is not None for b in self.blobs)
            self.results_dict["blobs"] = self.get("blobs") if has_blobs else None
            self.results_dict["logw"], _ = self.compute_logw_and_logz(1.0)
        return self.results_dict

    # ------------------------------------------------------------- checkpoints
    def __getstate__(self):
        return dict(n_particles=self.n_particles, n_dim=self.n_dim, T=self.T, scalars=self.scalars, blobs=self.blobs,
                    rows={k: self.rows(k).cpu().numpy() for k in ROW_KEYS})

    def __setstate__(self, st):
        self.__init__(st["n_particles"], st["n_dim"])
        T = st["T"]
        while self._cap < max(T, 1):
            self._grow(2 * self._cap)
        for k in ROW_KEYS:
            self._rows[k][:T * self.n_particles].copy_(torch.from_numpy(st["rows"][k]))
        self.T, self.scalars, self.blobs = T, st["scalars"], st["blobs"]
