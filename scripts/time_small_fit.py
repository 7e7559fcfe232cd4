"""The Sampler's usual fit (sampler.py:655-669 at the README's sizes): 512 rows, half of them validation, one batch per
epoch -- where an epoch's time goes.   python scripts/time_small_fit.py [flow] [D] [rows] [epochs]"""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pocomc_amd import Flow

flow = sys.argv[1] if len(sys.argv) > 1 else "nsf6"
D = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 512
epochs = int(sys.argv[4]) if len(sys.argv) > 4 else 300
f = Flow(D, flow, seed=0)
x = torch.from_numpy(np.random.default_rng(0).normal(size=(rows, D)).astype(np.float32)).cuda()
w = torch.full((rows,), 1.0 / rows).cuda()
f.fit(x, weights=w, epochs=5, batch_size=512, validation_split=0.5, patience=10 ** 6, annealing=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
h = f.fit(x, weights=w, epochs=epochs, batch_size=512, validation_split=0.5, patience=10 ** 6, annealing=False)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"flow": flow, "D": D, "rows": rows, "epochs": len(h["loss"]), "us_per_epoch": 1e6 * dt / len(h["loss"])}))
