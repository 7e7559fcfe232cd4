"""Data-parallel flow training (gradient all-reduce before the global-norm
clip, loss all-reduce for the early-stop decision): two ranks, each with half of every global batch,
take the same optimizer steps as one process training on the whole batches.

The GPU box has one device, so both ranks share it and talk over ``gloo`` (the collectives are the same
``torch.distributed`` calls that run on RCCL with one rank per GPU)."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data(n=1024, D=6):
    rng = np.random.default_rng(11)
    x = (rng.normal(size=(n, D)) * np.linspace(0.5, 2.0, D) + 0.3).astype(np.float32)
    w = rng.uniform(0.2, 1.0, size=n).astype(np.float32)
    return x, w


def _worker(rank, world, port, flow_name, weighted, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pocomc_amd import Flow
    x, w = _data()
    B, lb = 256, 256 // world
    # this rank's rows: its slice of every global batch, batch after batch
    rows = np.concatenate([np.arange(b0 + rank * lb, b0 + (rank + 1) * lb) for b0 in range(0, len(x), B)])
    f = Flow(x.shape[1], flow_name, seed=5)
    hist = f.fit(torch.from_numpy(x[rows]), weights=torch.from_numpy(w[rows]) if weighted else None,
                 validation_split=0.75, epochs=4, batch_size=B, shuffle=False, annealing=False, patience=100)
    if rank == 0:
        np.savez(out, params=f.params.cpu().numpy(), loss=np.array(hist["loss"]), val=np.array(hist["val_loss"]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("flow_name,weighted", [("maf3", False), ("maf3", True), ("nsf3", True)])
def test_two_ranks_equal_one_process(tmp_path, flow_name, weighted):
    import torch.multiprocessing as mp
    from pocomc_amd import Flow
    x, w = _data()
    B = 256
    # one process: with shuffle=False and validation_split=0.75 the training rows are the first 768;
    # arrange the sharded run's rows so that both see the same global batches
    out = str(tmp_path / "r0.npz")
    mp.spawn(_worker, args=(2, _free_port(), flow_name, weighted, out), nprocs=2, join=True)
    got = np.load(out)
    # the sharded run trains on rows [0:3/4) of each rank's shard = the first 3 of its 4 batch slices,
    # i.e. global batches 0..2; the single process sees the same batches in rows [0:768)
    f = Flow(x.shape[1], flow_name, seed=5)
    hist = f.fit(torch.from_numpy(x), weights=torch.from_numpy(w) if weighted else None, validation_split=0.75,
                 epochs=4, batch_size=B, shuffle=False, annealing=False, patience=100, sharded=False)
    np.testing.assert_allclose(got["loss"], hist["loss"], rtol=2e-5)
    np.testing.assert_allclose(got["val"], hist["val_loss"], rtol=2e-5)
    np.testing.assert_allclose(got["params"], f.params.cpu().numpy(), rtol=2e-3, atol=2e-5)
