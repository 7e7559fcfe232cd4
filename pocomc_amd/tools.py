"""Particle math around the MCMC step -- host-side mirror of ``pocomc/tools.py`` and
``pocomc/particles.py:215-231`` whose reductions run on the GPU
(``pmc_logw``, ``pmc_logw_stats``, ``pmc_resample_*``, ``pmc_gather``)."""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch

from . import _lib

SQRTEPS = math.sqrt(float(np.finfo(np.float64).eps))


def _dev():
    return _lib.require_gpu()


def _up(a, dtype=np.float64):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).to(_dev())


def logw_stats(logw_d, k=0):
    """``[max, sum exp(logw-max), sum exp(2(logw-max)), sum 1-(1-w)^k]`` of a device vector."""
    lib = _lib.load()
    P = logw_d.numel()
    stats = torch.zeros(4, dtype=torch.float64, device=logw_d.device)
    ws = torch.empty(int(lib.pmc_reduce_workspace_bytes(P)), dtype=torch.uint8, device=logw_d.device)
    with torch.cuda.device(logw_d.device):
        _lib.check(lib.pmc_logw_stats(_lib.ptr(logw_d), P, int(k), _lib.ptr(stats), _lib.ptr(ws),
                                      _lib.stream_handle()), "pmc_logw_stats")
    return stats.cpu().numpy()


def combine_logw_stats(per_shard):
    """Merge the ``[max, sum exp(logw-max), sum exp(2(logw-max))]`` triples of several shards into the
    triple of their union (rescaling every shard to the global maximum)."""
    per_shard = np.asarray(per_shard, dtype=np.float64).reshape(-1, per_shard[0].shape[-1] if hasattr(per_shard[0], "shape") else len(per_shard[0]))
    m = per_shard[:, 0]
    M = m.max()
    sc = np.exp(m - M)
    return np.array([M, np.sum(per_shard[:, 1] * sc), np.sum(per_shard[:, 2] * sc * sc)])


def allgather_logw_stats(local_stats, group=None):
    """Global ESS / logZ statistics of a walker-sharded pool (the reduction
    behind the temperature ladder, sampler.py:739-777): all-gather three doubles per rank, merge.
    Without a process group the local triple is returned."""
    import torch.distributed as dist
    local = np.asarray(local_stats, dtype=np.float64)[:3]
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local.copy()
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    mine = torch.tensor(local, dtype=torch.float64, device=dev)
    parts = [torch.empty_like(mine) for _ in range(dist.get_world_size(group))]
    dist.all_gather(parts, mine, group=group)
    return combine_logw_stats([p.cpu().numpy() for p in parts])


def ess_from_stats(st):
    """``effective_sample_size`` of normalised weights (tools.py:56-71): ``1 / sum w~^2``."""
    return float(st[1] * st[1] / st[2])


def compute_logw_and_logz(logl, beta, logz, beta_final=1.0, normalize=True):
    """``pocomc/particles.py:215-231`` on plain arrays (``logl`` is ``(T, N)``)."""
    lib = _lib.load()
    logl = np.asarray(logl, dtype=np.float64)
    T, N = logl.shape
    ld, bd, zd = _up(logl), _up(beta), _up(logz)
    lw = torch.empty(T * N, dtype=torch.float64, device=ld.device)
    with torch.cuda.device(ld.device):
        _lib.check(lib.pmc_logw(_lib.ptr(ld), _lib.ptr(bd), _lib.ptr(zd), float(beta_final), _lib.ptr(lw), T, N,
                                _lib.stream_handle()), "pmc_logw")
    st = logw_stats(lw)
    lse = st[0] + np.log(st[1])
    logz_new = lse - np.log(T * N)
    logw = lw.cpu().numpy()
    if normalize:
        logw -= lse
    return logw, logz_new


class PoolWeights:
    """The persistent pool's ``logl`` (T, N), ``beta`` (T,) and ``logz`` (T,) resident on the device for one
    ``Sampler._reweight``: the beta bisection (``sampler.py:739-777``) evaluates the mixture log-weights
    (``particles.py:215-231``) and their ESS a dozen times on the SAME history, so every trial is two launches
    (``pmc_logw``, ``pmc_logw_stats``) and four doubles back, instead of an upload of the history, a download of
    the log-weights and a second round trip for the ESS."""

    def __init__(self, logl, beta, logz):
        self.lib = _lib.load()
        logl = np.asarray(logl, dtype=np.float64)
        self.T, self.N = logl.shape
        self.P = self.T * self.N
        self.ld, self.bd, self.zd = _up(logl), _up(beta), _up(logz)
        dev = self.ld.device
        self.lw = torch.empty(self.P, dtype=torch.float64, device=dev)
        self.stats_d = torch.zeros(4, dtype=torch.float64, device=dev)
        self.ws = torch.empty(int(self.lib.pmc_reduce_workspace_bytes(self.P)), dtype=torch.uint8, device=dev)
        self.h_stats = torch.zeros(4, dtype=torch.float64).pin_memory()

    def stats(self, beta_final, k=0):
        """``[max, sum exp(logw-max), sum exp(2(logw-max)), sum 1-(1-w)^k]`` of the log-weights at ``beta_final``."""
        lib = self.lib
        with torch.cuda.device(self.lw.device):
            st = _lib.stream_handle()
            _lib.check(lib.pmc_logw(_lib.ptr(self.ld), _lib.ptr(self.bd), _lib.ptr(self.zd), float(beta_final),
                                    _lib.ptr(self.lw), self.T, self.N, st), "pmc_logw")
            _lib.check(lib.pmc_logw_stats(_lib.ptr(self.lw), self.P, int(k), _lib.ptr(self.stats_d), _lib.ptr(self.ws),
                                          st), "pmc_logw_stats")
            self.h_stats.copy_(self.stats_d, non_blocking=True)
            torch.cuda.current_stream().synchronize()
        return self.h_stats.numpy().copy()

    def ess(self, beta_final):
        """``effective_sample_size(exp(logw - max))`` (tools.py:56-71) at ``beta_final``."""
        st = self.stats(beta_final)
        return (st[1] * st[1]) / st[2]

    def uss(self, beta_final, k=None):
        """``unique_sample_size`` (tools.py:74-93) at ``beta_final``."""
        return self.stats(beta_final, k=self.P if k is None else int(k))[3]

    def logw_and_logz(self, beta_final, normalize=True):
        """``Particles.compute_logw_and_logz`` (particles.py:215-231): host log-weights and logZ."""
        st = self.stats(beta_final)
        lse = st[0] + np.log(st[1])
        logw = self.lw.cpu().numpy()
        if normalize:
            logw -= lse
        return logw, lse - np.log(self.P)


def trim_weights(samples, weights, ess=0.99, bins=1000):
    """``pocomc/tools.py:10-53`` (normalises ``weights`` in place like the reference).  The
    threshold search runs on the GPU (``pmc_trim_threshold``); the final mask / renormalisation
    are the reference's own two lines."""
    lib = _lib.load()
    weights /= np.sum(weights)
    wd = _up(weights)
    P = wd.numel()
    nbytes = int(lib.pmc_trim_workspace_bytes(P))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=wd.device)
    res = torch.zeros(2, dtype=torch.float64, device=wd.device)
    with torch.cuda.device(wd.device):
        _lib.check(lib.pmc_trim_threshold(_lib.ptr(wd), P, float(ess), int(bins), _lib.ptr(res), _lib.ptr(ws), nbytes,
                                          _lib.stream_handle()), "pmc_trim_threshold")
    threshold = float(res[0].item())
    mask = weights >= threshold
    weights_trimmed = weights[mask]
    weights_trimmed /= np.sum(weights_trimmed)
    return samples[mask], weights_trimmed


def effective_sample_size(weights):
    """``pocomc/tools.py:56-71`` (normalises ``weights`` in place like the reference)."""
    weights /= np.sum(weights)
    with np.errstate(divide="ignore"):
        st = logw_stats(_up(np.log(weights)))
    return (st[1] * st[1]) / st[2]


def unique_sample_size(weights, k=None):
    """``pocomc/tools.py:74-93``."""
    if k is None:
        k = len(weights)
    weights /= np.sum(weights)
    with np.errstate(divide="ignore"):
        st = logw_stats(_up(np.log(weights)), k=int(k))
    return st[3]


def compute_ess(logw):
    """``pocomc/tools.py:96-114``."""
    st = logw_stats(_up(logw))
    return (st[1] * st[1]) / st[2] / len(logw)


def increment_logz(logw):
    """``pocomc/tools.py:117-133``."""
    st = logw_stats(_up(logw))
    return st[0] + np.log(st[1])


def device_sum(a_d):
    """``np.sum`` of a float64 device vector, added in a fixed order on the device."""
    lib = _lib.load()
    out = torch.empty(1, dtype=torch.float64, device=a_d.device)
    with torch.cuda.device(a_d.device):
        _lib.check(lib.pmc_sum_f64(_lib.ptr(a_d), a_d.numel(), _lib.ptr(out), _lib.stream_handle()), "pmc_sum_f64")
    return float(out.item())


def systematic_resample(size, weights, random_state=None, offset=None, device_indices=False):
    """``pocomc/tools.py:136-186``; indices are bit-exact with the reference.  ``weights`` may be a float64 device
    tensor (the pool's weights); ``device_indices=True`` leaves the indices on the device."""
    lib = _lib.load()
    if random_state is not None:
        np.random.seed(random_state)
    if isinstance(weights, torch.Tensor):
        wd = weights.to(_dev(), torch.float64).contiguous()
        total = device_sum(wd)
        if abs(total - 1.) > SQRTEPS:
            wd = wd / total
    else:
        if abs(np.sum(weights) - 1.) > SQRTEPS:
            weights = np.array(weights) / np.sum(weights)
        wd = _up(weights)
    if offset is None:
        offset = np.random.random()
    cdf = torch.empty_like(wd)
    idx = torch.empty(int(size), dtype=torch.int64, device=wd.device)
    with torch.cuda.device(wd.device):
        _lib.check(lib.pmc_resample_systematic(_lib.ptr(wd), wd.numel(), float(offset), int(size), _lib.ptr(cdf),
                                               _lib.ptr(idx), _lib.stream_handle()), "pmc_resample_systematic")
    return idx if device_indices else idx.cpu().numpy()


def multinomial_resample(size, weights, uniforms=None, device_indices=False):
    """``np.random.choice(len(w), size, p=w)`` of ``pocomc/sampler.py:703`` (same uniforms from numpy's legacy stream,
    same ``cdf.searchsorted(u, 'right')``), with numpy's validation of ``p`` (``ValueError`` for NaN / negative /
    non-normalised probabilities).  ``weights`` may be a float64 device tensor."""
    lib = _lib.load()
    if isinstance(weights, torch.Tensor):
        wd = weights.to(_dev(), torch.float64).contiguous()
        total = device_sum(wd)                      # (NaN weights make the total NaN; negative ones: logw_stats)
    else:
        weights = np.asarray(weights, dtype=np.float64)
        if np.any(weights < 0):
            raise ValueError("probabilities are not non-negative")
        wd = _up(weights)
        total = float(np.sum(weights))
    if not np.isfinite(total):
        raise ValueError("probabilities contain NaN")
    if abs(total - 1.) > SQRTEPS:
        raise ValueError("probabilities do not sum to 1")
    if uniforms is None:
        uniforms = np.random.random_sample(int(size))
    ud = _up(uniforms)
    cdf = torch.empty_like(wd)
    idx = torch.empty(int(size), dtype=torch.int64, device=wd.device)
    with torch.cuda.device(wd.device):
        _lib.check(lib.pmc_resample_multinomial(_lib.ptr(wd), wd.numel(), _lib.ptr(ud), int(size), _lib.ptr(cdf),
                                                _lib.ptr(idx), _lib.stream_handle()), "pmc_resample_multinomial")
    return idx if device_indices else idx.cpu().numpy()


def gather(idx, u, x, logdetj, logl, logp):
    """``pocomc/sampler.py:707-713``: the five row gathers of ``_resample``."""
    lib = _lib.load()
    idx_d = _up(idx, np.int64)
    n_out = idx_d.numel()
    ud, xd, a, b, c = _up(u), _up(x), _up(logdetj), _up(logl), _up(logp)
    D = ud.shape[1]
    uo = torch.empty(n_out, D, dtype=torch.float64, device=ud.device)
    xo = torch.empty_like(uo)
    ao, bo, co = (torch.empty(n_out, dtype=torch.float64, device=ud.device) for _ in range(3))
    with torch.cuda.device(ud.device):
        _lib.check(lib.pmc_gather(_lib.ptr(idx_d), n_out, D, _lib.ptr(ud), _lib.ptr(xd), _lib.ptr(a), _lib.ptr(b),
                                  _lib.ptr(c), _lib.ptr(uo), _lib.ptr(xo), _lib.ptr(ao), _lib.ptr(bo), _lib.ptr(co),
                                  _lib.stream_handle()), "pmc_gather")
    g = lambda t: t.cpu().numpy()
    return g(uo), g(xo), g(ao), g(bo), g(co)
