"""The N>1 path on CPU: ``gloo`` processes, world size 2, 4 and 8 (BASELINE's metric is quoted at 1/2/4/8 GPUs, configs
4 and 5 are 8-rank configurations).  Each rank owns an N / world share of the walkers;
per step it reduces its own shard, all-reduces the D+4 sums (``pocomc_amd.mcmc.allreduce_sums``)
and runs the product's scalar logic (``pocomc_amd.mcmc.Adaptation``).  Both ranks must take
identical sigma / mu / stop decisions, equal to the unsharded oracle run."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cases
from oracle import mcmc as omcmc
from oracle.maf import OracleMAF, TorchFlowAdapter
from oracle.scaler import Reparameterize


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_trace(name):
    c = cases.MCMC_CASES[name]
    state, funcs, opts, aux = cases.build_case(name, Reparameterize)
    funcs["flow"] = TorchFlowAdapter(OracleMAF(aux["spec"], aux["flat"]))
    trace = []
    np.random.seed(c["seed"])
    res = getattr(omcmc, c["kind"])(state, funcs, opts, trace=trace)
    return c, state, funcs, opts, trace, res


def _worker(rank, world, port, name, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pocomc_amd.mcmc import Adaptation, allreduce_sums
    c, state, funcs, opts, trace, res = _oracle_trace(name)
    N, D = c["N"], c["D"]
    lo, hi = rank * N // world, (rank + 1) * N // world
    kind = c["kind"]
    pre = kind.startswith("preconditioned")
    tpcn = kind in ("preconditioned_pcn", "pcn")
    geo = funcs["theta_geometry"]
    init = torch.tensor([0.0, float(np.sum((state["logl"] + state["logp"])[lo:hi])),
                         float(np.sum((state["logl"] + state["logp"] + state["logdetj"])[lo:hi]))] + [0.0] * (D + 1),
                        dtype=torch.float64)
    allreduce_sums(init)
    ad = Adaptation(kind, D, N, opts["n_steps"], opts["n_max"], opts["proposal_scale"],
                    geo.t_mean if tpcn else None, float(init[1] if tpcn else init[2]) / N)
    sig, mus, stops = [], [], []
    for tr in trace:
        moved = tr["theta"] if pre else tr["u"]
        lp = tr["logl"] + tr["logp"]
        sums = torch.tensor(np.concatenate([[tr["alpha"][lo:hi].sum(), lp[lo:hi].sum(),
                                             (lp + tr["logdetj"])[lo:hi].sum(), tr["accept"][lo:hi].sum()],
                                            moved[lo:hi].astype(np.float64).sum(axis=0)]), dtype=torch.float64)
        allreduce_sums(sums)                       # the step's only exchange
        stops.append(bool(ad.update(sums.numpy())))
        sig.append(float(ad.sigma))
        mus.append(None if ad.mu is None else ad.mu.copy())
    out[rank] = dict(sigma=sig, mu=mus, stops=stops, steps=ad.i)
    dist.destroy_process_group()


@pytest.mark.parametrize("name,world", [("tpcn_n256_d10_normal", 2), ("tpcn_n256_d10_normal", 4),
                                        ("tpcn_n256_d10_normal", 8), ("prwm_n128_d8_uniform", 2),
                                        ("prwm_n128_d8_uniform", 8), ("pcn_n128_d8_uniform", 2),
                                        ("pcn_n128_d8_uniform", 4), ("rwm_n128_d8_normal", 2),
                                        ("rwm_n128_d8_normal", 8)])
def test_sharded_adaptation_equals_unsharded(name, world):
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, name, out), nprocs=world, join=True)
    a, b = out[0], out[world - 1]
    for r in range(1, world):                       # every rank takes the same decisions, bit for bit
        assert out[r]["sigma"] == a["sigma"] and out[r]["stops"] == a["stops"] and out[r]["steps"] == a["steps"]
        if a["mu"][0] is not None:
            assert all(np.array_equal(m0, m1) for m0, m1 in zip(a["mu"], out[r]["mu"]))
    c, state, funcs, opts, trace, res = _oracle_trace(name)
    assert a["steps"] == res["steps"] == len(trace)
    assert a["stops"][-1] and not any(a["stops"][:-1])
    np.testing.assert_allclose(a["sigma"], [t["sigma"] for t in trace], rtol=1e-12)
    if c["kind"] == "preconditioned_pcn":
        for m0, m1, t in zip(a["mu"], b["mu"], trace):
            np.testing.assert_array_equal(m0, m1)
            # np.mean of the float32 theta array (mcmc.py:156) vs float64 sums rounded to float32
            np.testing.assert_allclose(m0, t["mu"], rtol=1e-6, atol=1e-7)


def test_philox_offsets_make_shards_independent_of_world_size():
    """Host-side contract of the sharded RNG: the walker's global index keys the stream."""
    from pocomc_amd import _lib
    r = _lib.pmc_rng_t(gamma=None, normal=None, uniform=None, seed=7, step=3, offset=5000)
    assert r.offset == 5000 and r.seed == 7 and r.step == 3


# ---------------------------------------------------------------- ESS / logZ of a sharded pool
def _ess_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pocomc_amd.tools import allgather_logw_stats, ess_from_stats
    logw = _pool_logw()
    P = len(logw)
    mine = logw[rank * P // world:(rank + 1) * P // world]
    m = mine.max()                                   # what pmc_logw_stats returns for the shard
    local = np.array([m, np.exp(mine - m).sum(), np.exp(2 * (mine - m)).sum()])
    st = allgather_logw_stats(local)
    out[rank] = (ess_from_stats(st), st[0] + np.log(st[1]) - np.log(P))
    dist.destroy_process_group()


def _pool_logw():
    rng = np.random.default_rng(4)
    return np.concatenate([rng.normal(size=1500) * 3.0 - 40.0, rng.normal(size=500) * 0.5 + 25.0])


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_ess_and_logz_equal_the_oracle_on_the_whole_pool(world):
    """The temperature-ladder reduction (sampler.py:739-777) over a walker-sharded pool: three doubles
    per rank are all-gathered and merged; ESS and the logZ increment equal the oracle's on the
    concatenated weights, on every rank, although the shards' maxima differ by 60 nats."""
    from oracle import tools as otools
    logw = _pool_logw()
    w = np.exp(logw - logw.max())
    ess_ref = otools.effective_sample_size(w / w.sum())
    logz_ref = np.log(np.mean(np.exp(logw - logw.max()))) + logw.max()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_ess_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    for r in range(world):
        np.testing.assert_allclose(out[r][0], ess_ref, rtol=1e-12)
        np.testing.assert_allclose(out[r][1], logz_ref, rtol=1e-12)


@pytest.mark.parametrize("kind", ["preconditioned_pcn", "preconditioned_rwm", "pcn", "rwm"])
def test_adaptation_coefficients_reproduce_the_host_update(kind):
    """What the device-side adaptation is given (``Adaptation.coefficients``: mode, 1/(i+1)^0.75 or 1/(i+1), 1/(i+1),
    cap) and the expressions it evaluates (``adapt_apply`` in csrc/mcmc_kernels.hip, restated here in numpy float64
    in the same order) give bit for bit the sigma and mu of ``Adaptation.update`` (mcmc.py:152-156, :314, :476, :627)."""
    from pocomc_amd.mcmc import Adaptation, PMC_ADAPT_MU, PMC_ADAPT_PRWM, PMC_ADAPT_RWM, PMC_ADAPT_TPCN
    D, N = 7, 1000
    rng = np.random.default_rng(3)
    mu0 = rng.normal(size=D) if kind == "preconditioned_pcn" else None
    ad = Adaptation(kind, D, N, n_steps=10 ** 9, n_max=10 ** 9, sigma0=0.9, mu0=mu0, logp2_0=-np.inf)
    sigma = np.float64(ad.sigma)
    mu = None if mu0 is None else np.array(mu0)
    for step in range(40):
        sums = np.concatenate([[rng.uniform(0, N), rng.normal() * N, rng.normal() * N, rng.integers(0, N)],
                               rng.normal(size=D) * N])
        mode, c_sigma, c_mu, cap = ad.coefficients()
        # the device's arithmetic
        mean_alpha = sums[0] / np.float64(N)
        sn = sigma + np.float64(c_sigma) * (mean_alpha - 0.234)
        how = mode & 7
        if how == PMC_ADAPT_TPCN:
            sn = np.abs(np.minimum(sn, np.float64(cap)))
        elif how == PMC_ADAPT_RWM:
            sn = np.abs(sn)
        else:
            assert how == PMC_ADAPT_PRWM
        sigma = sn
        if mode & PMC_ADAPT_MU:
            mean_theta = (sums[4:4 + D] / np.float64(N)).astype(np.float32)
            mu = mu + np.float64(c_mu) * (mean_theta.astype(np.float64) - mu)
        ad.update(sums)
        assert np.float64(ad.sigma) == sigma, step
        if mu is not None:
            assert np.array_equal(ad.mu, mu), step
    assert (mu is not None) == (kind == "preconditioned_pcn")
