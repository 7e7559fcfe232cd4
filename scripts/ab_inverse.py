"""A/B of the inverse sweeps (AUTO / lone wave / two waves / lane-per-walker) on a few flow shapes; measurement only.
    [PMC_TRI6_SUBSETS=1|2|4] python scripts/ab_inverse.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pocomc_amd as pc
shapes = ((32, "maf3", 7008), (32, "maf3", 2992), (32, "maf3", 10000), (10, "maf6", 1000), (50, "maf6", 2992), (50, "maf6", 4096),
          (50, "maf6", 7008), (50, "maf6", 10000), (64, "maf3", 4096), (64, "maf3", 10000))
for (D, name, n) in shapes:
    f = pc.Flow(D, name, seed=0)
    z = torch.randn(n, D, device="cuda")
    res = {}
    for algo, label in ((0, "auto"), (6, "solo"), (7, "duo"), (8, "lane")):
        try:
            f.inverse_algo = algo
            for _ in range(3):
                f.inverse(z)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                f.inverse(z)
            e1.record(); torch.cuda.synchronize()
            res[label] = round(e0.elapsed_time(e1) / 20 * 1e3, 1)
        except Exception as e:
            res[label] = repr(e)[:40]
    print(D, name, n, res)
