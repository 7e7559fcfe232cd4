"""Host-side cost of the bench's numpy likelihood: rows per call, numpy's iterator buffer size, memory order."""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from threadpoolctl import threadpool_limits
libc = ctypes.CDLL("libc.so.6"); libc.mallopt(-3, 1 << 30); libc.mallopt(-1, 1 << 30)
os.sched_setaffinity(0, {sorted(os.sched_getaffinity(0))[4]})
D = 32
t = bench.make_target("rosenbrock", D)
with threadpool_limits(limits=1):
    for bs in (8192, 1024):
        np.setbufsize(bs)
        for order in ("F", "C"):
            for n in (2500, 5008, 10000):
                if order == "F":
                    h = torch.empty(D, n, dtype=torch.float64).pin_memory()
                    h.copy_(torch.randn(D, n, dtype=torch.float64))
                    x = h.numpy().T
                else:
                    h = torch.empty(n, D, dtype=torch.float64).pin_memory()
                    h.copy_(torch.randn(n, D, dtype=torch.float64))
                    x = h.numpy()
                for _ in range(100): t(x)
                t0 = time.perf_counter()
                for _ in range(500): t(x)
                dt = (time.perf_counter() - t0) / 500 * 1e6
                print("bufsize", bs, "order", order, "rows", n, round(dt, 1), "us per call", round(dt / n * 1e3, 1), "ns/row")
