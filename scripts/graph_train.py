"""Config-5 training step (128-D, 8 transforms, H = 512, bf16 matrix cores, batches of 512 rows): the 10 batches of an
epoch as the library enqueues them, and the same launches replayed from a captured hipGraph (torch.cuda.CUDAGraph).
Timing only: a replay repeats the captured optimizer step number, so the replayed training is not a valid one.
    python scripts/graph_train.py [rows] [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pocomc_amd import Flow
from pocomc_amd.maf_spec import MAFSpec
from pocomc_amd import train as T
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 5120
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 512
D = 128
f = Flow(D, MAFSpec(D, 8, 512), seed=0, precision="bf16")
x = torch.from_numpy(np.random.default_rng(0).normal(size=(rows, D)).astype(np.float32)).cuda()
f.fit(x, epochs=2, batch_size=batch, validation_split=0.0)          # (allocations, images)
opt = T.AdamW(f, 1e-3, 0.0)
perm = torch.randperm(rows, device="cuda")
acc = torch.zeros(1, dtype=torch.float32, device="cuda")
nb = (rows + batch - 1) // batch
def epoch():
    opt.epoch(x, None, perm, batch, 1.0, acc)
for _ in range(3): epoch()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): epoch()
torch.cuda.synchronize()
eager = (time.perf_counter() - t0) / 20
print(f"eager: {eager * 1e3:7.3f} ms per epoch of {nb} batches = {eager / nb * 1e6:6.1f} us per step")
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.stream(s):
        epoch()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            epoch()
    torch.cuda.synchronize()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    rep = (time.perf_counter() - t0) / 20
    print(f"graph: {rep * 1e3:7.3f} ms per epoch of {nb} batches = {rep / nb * 1e6:6.1f} us per step  ({(1 - rep / eager) * 100:.1f} % less)")
except Exception as e:
    print("capture failed:", repr(e)[:300])
