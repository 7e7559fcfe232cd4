"""The Sampler with one process per GPU: MCMC steps, likelihood calls and flow fits
sharded over the ranks, pool bookkeeping replicated.  The GPU box has one device, so the two ranks share
it and talk over ``gloo``; the collectives are the ``torch.distributed`` calls that run on RCCL with one
rank per GPU."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

D = 4
COV = np.array([[1.0, 0.6, 0.0, 0.0], [0.6, 1.5, 0.3, 0.0], [0.0, 0.3, 0.8, -0.2], [0.0, 0.0, -0.2, 0.5]])
ICOV = np.linalg.inv(COV)
NORM = -0.5 * (D * np.log(2 * np.pi) + np.linalg.slogdet(COV)[1])
TRUE_LOGZ = -0.5 * (D * np.log(2 * np.pi) + np.linalg.slogdet(COV + 25.0 * np.eye(D))[1])


def loglike(x):
    return NORM - 0.5 * np.einsum("ni,ij,nj->n", x, ICOV, x)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out, fit_parallel="auto"):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from scipy.stats import norm
    import pocomc_amd as pc
    calls = [0]

    def counted(x):
        calls[0] += len(x)
        return loglike(x)

    prior = pc.Prior([norm(0.0, 5.0)] * D)
    s = pc.Sampler(prior=prior, likelihood=counted, vectorize=True, flow="maf3", n_active=256, n_effective=512,
                   random_state=4, train_config=dict(fit_parallel=fit_parallel))
    assert s.world == world and s.rank == rank
    s.run(n_total=1024, n_evidence=1024, progress=False)
    logz, err = s.evidence()
    x, w, logl, logp = s.posterior()
    np.savez(out % rank, logz=logz, beta=np.asarray(s.particles.get("beta")), x=x, w=w, calls=s.calls,
             own_calls=calls[0], params=s.flow.params.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("fit_parallel", ["auto", "data"])
def test_two_rank_sampler(tmp_path, fit_parallel):
    """``fit_parallel="data"``: the flow fits are data parallel (gradients all-reduced before the clip); ``"auto"`` at this
    size (128 local rows per batch) lets every rank run the whole fit on the replicated pool -- either way both ranks hold
    the same flow bit for bit."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "rank%d.npz")
    mp.spawn(_worker, args=(2, _free_port(), out, fit_parallel), nprocs=2, join=True)
    r0, r1 = np.load(out % 0), np.load(out % 1)
    # replicated bookkeeping: identical temperature ladder, pool, evidence and flow on both ranks
    assert np.array_equal(r0["beta"], r1["beta"])
    assert np.array_equal(r0["x"], r1["x"]) and np.array_equal(r0["w"], r1["w"])
    assert float(r0["logz"]) == float(r1["logz"])
    assert np.array_equal(r0["params"], r1["params"])
    # the likelihood work is shared: each rank made about half of the calls
    assert int(r0["calls"]) == int(r1["calls"])
    assert abs(int(r0["own_calls"]) + int(r1["own_calls"]) - int(r0["calls"])) <= 0.02 * int(r0["calls"])
    assert 0.4 < int(r0["own_calls"]) / int(r0["calls"]) < 0.6
    # and the answer is right
    assert abs(float(r0["logz"]) - TRUE_LOGZ) < 0.35, (float(r0["logz"]), TRUE_LOGZ)
    m = np.average(r0["x"], weights=r0["w"], axis=0)
    c = np.cov(r0["x"].T, aweights=r0["w"])
    post = np.linalg.inv(ICOV + np.eye(D) / 25.0)
    assert np.abs(m).max() < 0.25
    assert np.abs(c - post).max() < 0.35


def _ckpt_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from scipy.stats import norm
    import pocomc_amd as pc
    prior = pc.Prior([norm(0.0, 5.0)] * D)
    mk = lambda: pc.Sampler(prior=prior, likelihood=loglike, vectorize=True, flow="maf3", n_active=128, n_effective=256,
                            random_state=9, train_config=dict(epochs=20), output_dir=out_dir, output_label="sh")
    s = mk()
    s.run(n_total=512, n_evidence=0, progress=False, save_every=2)
    dist.barrier()
    files = sorted(p_ for p_ in os.listdir(out_dir) if p_.endswith(".state"))
    assert "sh_final.state" in files and not any(p_.endswith(".temp") for p_ in os.listdir(out_dir))
    # every rank loads the file rank 0 wrote and keeps ITS OWN rank: a resumed run shards the walkers correctly
    first = sorted(f for f in files if "final" not in f)[0]
    s2 = mk()
    s2.run(n_total=512, n_evidence=0, progress=False, resume_state_path=os.path.join(out_dir, first))
    assert s2.rank == rank and s2.world == world
    x, w, _, _ = s2.posterior()
    np.savez(os.path.join(out_dir, f"resumed{rank}.npz"), x=x, w=w, logz=s2.evidence()[0], t=s2.t)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_checkpoint_and_resume(tmp_path):
    """save_state from one process per GPU (rank 0 writes, the others wait) and a resume in which every
    rank keeps its own rank / shard."""
    import torch.multiprocessing as mp
    mp.spawn(_ckpt_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "resumed0.npz"), np.load(tmp_path / "resumed1.npz")
    assert np.array_equal(r0["x"], r1["x"]) and np.array_equal(r0["w"], r1["w"]) and float(r0["logz"]) == float(r1["logz"])
    # the resumed walker sets were assembled from two DIFFERENT shards: no duplicated block of rows
    tail = r0["x"][-128:]
    assert len(np.unique(tail.round(12), axis=0)) > 64
    m = np.average(r0["x"], weights=r0["w"], axis=0)
    assert np.abs(m).max() < 0.4


def test_sharded_sampler_needs_a_seed():
    """the ranks replicate the pool bookkeeping from the same random streams -- random_state=None is refused."""
    import pocomc_amd as pc
    from scipy.stats import norm

    class TwoRanks(pc.sampler._Ranks):
        def __init__(self, group):
            self.group, self.world, self.rank = group, 2, 0
    orig = pc.sampler._Ranks
    pc.sampler._Ranks = TwoRanks
    try:
        with pytest.raises(ValueError, match="random_state"):
            pc.Sampler(prior=pc.Prior([norm(0, 1)] * 2), likelihood=lambda x: -0.5 * np.sum(x ** 2, axis=1), vectorize=True)
    finally:
        pc.sampler._Ranks = orig


def _kernel_case(Dk=6):
    from scipy.stats import uniform
    import torch
    import pocomc_amd as pc
    from pocomc_amd.geometry import Geometry
    N = 640
    prior = pc.Prior([uniform(-5, 10)] * Dk)
    rng = np.random.default_rng(12)
    scaler = pc.Reparameterize(Dk, bounds=prior.bounds)
    scaler.fit(rng.uniform(-5, 5, size=(2000, Dk)))
    x = rng.uniform(-4, 4, size=(N, Dk))
    u = scaler.forward(x)
    like = lambda xx: (-0.5 * np.sum(xx ** 2, axis=1), None)
    flow = pc.Flow(Dk, "maf3", seed=0)
    geo = Geometry()
    geo.fit(flow.forward(torch.from_numpy(u).float())[0].numpy().astype(np.float64))
    return prior, scaler, x, u, like, flow, geo


def _kernel_call(lo, hi, lanes, group_opts, Dk=6):
    from pocomc_amd import mcmc as pmcmc
    prior, scaler, x, u, like, flow, geo = _kernel_case(Dk)
    sl = slice(lo, hi)
    state = dict(u=u[sl].copy(), x=x[sl].copy(), logdetj=scaler.inverse(u[sl])[1], logl=like(x[sl])[0],
                 logp=prior.logpdf(x[sl]), beta=0.5, blobs=None)
    funcs = dict(loglike=like, logprior=prior.logpdf, scaler=scaler, flow=flow, theta_geometry=geo)
    opts = dict(n_max=6, n_steps=10 ** 6, progress_bar=None, proposal_scale=2.38 / Dk ** 0.5, seed=21, x_order="F",
                lanes=lanes, **group_opts)
    return pmcmc.preconditioned_pcn(state, funcs, opts)


def _kernel_worker(rank, world, port, out, lanes):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = rank * 640 // world, (rank + 1) * 640 // world
    r = _kernel_call(lo, hi, lanes, dict(group=None, shard_offset=lo))
    np.savez(out % rank, u=r["u"], logl=r["logl"], sigma=r["proposal_scale"], accept=r["accept"], steps=r["steps"])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,lanes", [(2, 1), (2, 2), (4, 2), (8, 2)])
def test_sharded_pipelined_kernel_call_equals_one_rank(tmp_path, world, lanes):
    """Walkers sharded over 2, 4 and 8 ranks (one GPU shared by the processes: the rank plumbing, not the links), adaptation
    on the device (lane sums -> exchange -> sigma / mu update, the next pre-step enqueued behind it): the same trajectory
    as the whole set on one rank, up to the order in which the sums are added."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "k%d.npz")
    mp.spawn(_kernel_worker, args=(world, _free_port(), out, lanes), nprocs=world, join=True)
    rs = [np.load(out % r) for r in range(world)]
    whole = _kernel_call(0, 640, 1, {})
    for r in rs:
        assert float(r["sigma"]) == float(rs[0]["sigma"]) and int(r["steps"]) == whole["steps"] == 6
    np.testing.assert_allclose(float(rs[0]["sigma"]), whole["proposal_scale"], rtol=1e-12)
    np.testing.assert_allclose(float(rs[0]["accept"]), whole["accept"], rtol=1e-12)
    u2 = np.concatenate([r["u"] for r in rs])
    same = np.isclose(u2, whole["u"], rtol=1e-9, atol=1e-12).all(axis=1)
    assert same.mean() >= 0.995, same.mean()
    np.testing.assert_allclose(np.concatenate([r["logl"] for r in rs])[same], whole["logl"][same], rtol=1e-8)


def _comm_worker(rank, world, port, out, c_allreduce, mailbox, Dk=6):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      PMC_C_ALLREDUCE=c_allreduce, PMC_COMM_MAILBOX=mailbox)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pocomc_amd import mcmc as pmcmc
    lo, hi = rank * 640 // world, (rank + 1) * 640 // world
    r = _kernel_call(lo, hi, 2, dict(group=None, shard_offset=lo), Dk)
    used = any(v[0] for v in pmcmc._COMMS.values())
    kinds = sorted({int(pmcmc._lib.load().pmc_comm_kind(v[0])) for v in pmcmc._COMMS.values() if v[0]})
    np.savez(out % rank, u=r["u"], x=r["x"], logl=r["logl"], sigma=r["proposal_scale"], accept=r["accept"], steps=r["steps"], used=used,
             kinds=np.array(kinds, dtype=np.int64))
    pmcmc.drop_comms()
    assert not pmcmc._COMMS
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,mailbox,Dk", [(2, "device", 6), (2, "host", 6), (4, "device", 6), (4, "host", 6),
                                              (8, "device", 6), (8, "host", 6), (8, "device", 32), (8, "host", 128)])
def test_the_sharded_step_behind_the_c_abi_equals_the_torch_distributed_path(tmp_path, world, mailbox, Dk):
    """2, 4 and 8 ranks, walkers sharded (D = 6, 32, 128: the two-wave sweep, the headline's, and the lane sweep of config 5's
    width): the step runs behind ``pmc_pipeline_*`` with the library's own exchange between the last accept and the
    adaptation (``pmc_comm_adapt_update``: mailboxes ``[2 parities][world][D + 5]``, the ranks' sums added in rank order)
    -- bit for bit what the Python pipeline produces with ``torch.distributed`` in that place (``PMC_C_ALLREDUCE=0``;
    ``allreduce_sums`` adds in rank order too).  ``mailbox="device"``: uncached HBM shared through hipIpc handles (the
    processes share one GPU here and map LOCAL memory; on a node every handle is a peer's).  ``mailbox="host"``: the
    mailboxes in pinned host memory (POSIX shared memory registered with every rank's runtime) -- every system-scope
    store, sequence word and acquire-poll of the protocol crosses PCIe to memory no device caches: the stand-in for a
    remote target on a one-GPU box.  No link between two GPUs is crossed by this test: what it rehearses is the 4- and
    8-way handle exchange, the mailbox indexing and the rank-ordered sums."""
    import torch.multiprocessing as mp
    res = {}
    for flag in ("1", "0"):
        out = str(tmp_path / f"c{flag}_%d.npz")
        mp.spawn(_comm_worker, args=(world, _free_port(), out, flag, mailbox, Dk), nprocs=world, join=True)
        res[flag] = [np.load(out % r) for r in range(world)]
    assert all(bool(r["used"]) for r in res["1"])                              # the communicator was created and connected
    assert all(r["kinds"].tolist() == [0 if mailbox == "device" else 1] for r in res["1"])
    assert not any(bool(r["used"]) for r in res["0"])
    for rk in range(world):
        a, b = res["1"][rk], res["0"][rk]
        assert int(a["steps"]) == int(b["steps"]) == 6
        assert float(a["sigma"]) == float(b["sigma"]) and float(a["accept"]) == float(b["accept"])
        for k in ("u", "x", "logl"):
            np.testing.assert_array_equal(a[k], b[k], err_msg=k)
        assert float(a["sigma"]) == float(res["1"][0]["sigma"])
