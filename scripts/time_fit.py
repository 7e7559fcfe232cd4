"""Time ``Flow.fit`` epochs on the GPU:  python scripts/time_fit.py D T [hidden] [rows] [batch] [precision]"""
import sys
import time

import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from pocomc_amd import Flow
from pocomc_amd.maf_spec import MAFSpec

D, T = int(sys.argv[1]), int(sys.argv[2])
H = int(sys.argv[3]) if len(sys.argv) > 3 and int(sys.argv[3]) > 0 else None
rows = int(sys.argv[4]) if len(sys.argv) > 4 else 5000
batch = int(sys.argv[5]) if len(sys.argv) > 5 else 512
prec = sys.argv[6] if len(sys.argv) > 6 else "f32"
spec = MAFSpec(D, T, hidden=H)
f = Flow(D, spec, precision=prec) if prec != "f32" else Flow(D, spec)
x = torch.from_numpy(np.random.default_rng(0).normal(size=(rows, D)))
for epochs in (3, 10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    h = f.fit(x, epochs=epochs, batch_size=batch, validation_split=0.0, patience=1000, verbose=0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = epochs * ((rows + batch - 1) // batch)
    # dense-equivalent flops: 3 x forward (forward, data gradient, weight gradient)
    fl = 3 * 2 * T * (D * spec.hidden + 2 * spec.hidden ** 2 + spec.hidden * 2 * D) * rows * epochs
    print(f"D={D} T={T} H={spec.hidden} rows={rows} batch={batch} {prec}: {epochs} epochs {dt * 1e3:.1f} ms, "
          f"{dt / steps * 1e6:.0f} us/step, {rows * epochs / dt:.3g} rows/s, {fl / dt / 1e12:.2f} TFLOP/s dense-equivalent, "
          f"loss {h['loss'][-1]:.3f}")
