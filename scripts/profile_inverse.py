"""In-kernel phase profile of the triangular inverse (debug entry point, not part of the ABI)."""
import ctypes as C
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pocomc_amd import Flow, _lib

n, D = int(sys.argv[1]) if len(sys.argv) > 1 else 10000, 32
f = Flow(D, "maf3", seed=0)
lib = _lib.load()
which = sys.argv[2] if len(sys.argv) > 2 else "tri3"
fn = lib.pmc_debug_inverse3_profile if which == "tri3" else lib.pmc_debug_inverse_profile
fn.restype = C.c_int
fn.argtypes = [C.POINTER(_lib.pmc_maf_t)] + [C.c_void_p] * 3 + [C.c_int64, C.c_void_p, C.c_void_p]
z = torch.randn(n, D, device="cuda")
x = torch.empty_like(z)
l = torch.empty(n, device="cuda")
nb = (n + 15) // 16
prof = torch.zeros(nb, 8, dtype=torch.int64, device="cuda")
for _ in range(3):
    _lib.check(fn(C.byref(f._desc), _lib.ptr(z), _lib.ptr(x), _lib.ptr(l), n, _lib.ptr(prof), _lib.stream_handle()))
torch.cuda.synchronize()
p = prof.cpu().numpy().astype(np.float64)
names = (["total", "setup", "tile-top loads", "bursts", "stage+prefetch", "R-load", "chain", "tail(zero/rerank)"] if which == "tri3"
         else ["total", "zero+rank0", "burst", "prefetch_issue", "chain(h0..h2)", "out(burst/diag)", "x+rank1", "wall_clock64"])
print("waves", nb)
for i, nm in enumerate(names):
    print(f"{nm:18s} mean {p[:, i].mean():12.0f}  min {p[:, i].min():12.0f}  max {p[:, i].max():12.0f}")
if which != "tri3": print("cycles/wallclock-tick", (p[:, 0] / p[:, 7]).mean(), "(wall_clock64 is 100 MHz => shader MHz =", (p[:, 0] / p[:, 7]).mean() * 100, ")")
