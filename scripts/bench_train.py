"""Flow-training throughput (BASELINE config sub-metric): epochs of Flow.fit on synthetic data.

    python scripts/bench_train.py [--dim 32] [--flow maf3|T,H] [--rows 10000] [--batch 512] [--epochs 20]

Prints one JSON line: ms/epoch, us/batch, samples/s."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pocomc_amd import Flow                     # noqa: E402
from pocomc_amd.maf_spec import MAFSpec         # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dim", type=int, default=32)
ap.add_argument("--flow", default="maf3")
ap.add_argument("--rows", type=int, default=10000)
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--epochs", type=int, default=20)
ap.add_argument("--warmup", type=int, default=3)
a = ap.parse_args()

flow = a.flow
if "," in flow:
    T, H = (int(v) for v in flow.split(","))
    flow = MAFSpec(a.dim, T, H)
f = Flow(a.dim, flow, seed=0)
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.normal(size=(a.rows, a.dim)).astype(np.float32)).cuda()
f.fit(x, epochs=a.warmup, batch_size=a.batch, validation_split=0.0)
torch.cuda.synchronize()
t0 = time.perf_counter()
h = f.fit(x, epochs=a.epochs, batch_size=a.batch, validation_split=0.0, patience=10 ** 6)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
nb = (a.rows + a.batch - 1) // a.batch
print(json.dumps({"flow": a.flow, "dim": a.dim, "rows": a.rows, "batch": a.batch, "epochs": a.epochs,
                  "ms_per_epoch": 1e3 * dt / a.epochs, "us_per_batch": 1e6 * dt / (a.epochs * nb),
                  "samples_per_s": a.rows * a.epochs / dt, "loss_first": h["loss"][0], "loss_last": h["loss"][-1]}))
