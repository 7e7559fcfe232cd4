"""Where does "every step 10 % slower in one process of ten" come from?  Times the benchmark's likelihood on a pinned,
device-written-like host buffer from several cores of the GPU's NUMA node, in a fresh process: run it a dozen times.
    python scripts/core_probe.py
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench

n, D = 6496, 32
x_h = torch.empty(D, n, dtype=torch.float64).pin_memory()          # column-major (n, D) like the engine's h_x
x = x_h.numpy().T
x[:] = np.random.default_rng(0).uniform(-10, 10, size=(n, D))
aff = sorted(os.sched_getaffinity(0))
try:
    pr = torch.cuda.get_device_properties(0)
    bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
    cl = open(f"/sys/bus/pci/devices/{bdf}/local_cpulist").read().strip()
    cpus = set()
    for part in cl.split(","):
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    cand = sorted(cpus & set(aff)) or aff
except Exception:
    cand = aff
np.setbufsize(1024)
out = []
for c in [cand[4 % len(cand)], cand[12 % len(cand)], cand[20 % len(cand)], cand[36 % len(cand)]]:
    os.sched_setaffinity(0, {c})
    for _ in range(20):
        bench.rosenbrock(x)
    ts = []
    for _ in range(200):
        t0 = time.perf_counter(); bench.rosenbrock(x); ts.append(time.perf_counter() - t0)
    out.append((c, float(np.median(ts)) * 1e6, float(np.min(ts)) * 1e6))
print(" ".join(f"core {c}: median {m:.1f} min {lo:.1f} us |" for c, m, lo in out), "addr x %x" % x_h.data_ptr())
