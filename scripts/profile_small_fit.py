"""cProfile of the Sampler-size fit (1024 rows, half validation, one batch per epoch): the host side of an epoch."""
import cProfile, pstats, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pocomc_amd import Flow
flow = sys.argv[1] if len(sys.argv) > 1 else "nsf6"
f = Flow(10, flow, seed=0)
x = torch.from_numpy(np.random.default_rng(0).normal(size=(1024, 10)).astype(np.float32)).cuda()
w = torch.full((1024,), 1.0 / 1024).cuda()
kw = dict(weights=w, batch_size=512, validation_split=0.5, patience=10 ** 6, annealing=False)
f.fit(x, epochs=5, **kw)
for ep in (1, 30, 300):
    torch.cuda.synchronize(); t0 = time.perf_counter(); f.fit(x, epochs=ep, **kw); torch.cuda.synchronize()
    print(f"epochs {ep}: {1e3 * (time.perf_counter() - t0):.2f} ms")
pr = cProfile.Profile(); pr.enable(); f.fit(x, epochs=300, **kw); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
