"""Kernel-level timing of Flow.forward / Flow.inverse (HIP events around repeated launches).

usage: time_flow.py [n[,n...]] [D] [flow]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pocomc_amd import Flow
ns = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [10000]
D = int(sys.argv[2]) if len(sys.argv) > 2 else 32
name = sys.argv[3] if len(sys.argv) > 3 else "maf3"
f = Flow(D, name, seed=0)
for n in ns:
    x = torch.randn(n, D, device="cuda")
    for fn, label in ((f.forward, "forward"), (f.inverse, "inverse"), (f.log_prob, "log_prob")):
        for _ in range(5):
            fn(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn(x)
        e1.record()
        torch.cuda.synchronize()
        print(f"{label:9s} n={n} D={D} {name}: {e0.elapsed_time(e1) / 50 * 1e3:8.1f} us per call (incl. output allocation)")
