"""Latency probe on the weight image (debug)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pocomc_amd import _lib
lib = _lib.load()
fn = lib.pmc_debug_latency_probe
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
nb = 32
w = torch.randn(nb * 64 * 64 * 4 + 4096, device="cuda")
out = torch.zeros(nb, 8, 8, dtype=torch.int64, device="cuda")
names = ["cold load", "same line", "4 new lines", "36 dep MFMA", "9x(ds_read+4MFMA)", "syncthreads", "lds barrier"]
for rep in range(3):
    if rep == 2:
        w.mul_(1.0)     # rewrite the image with another kernel
    _lib.check(fn(_lib.ptr(w), _lib.ptr(out), nb, _lib.stream_handle()))
    torch.cuda.synchronize()
    o = out.cpu().numpy().astype(float)
    print("launch", rep, "  ".join(f"{n}: {o[:, :, i].mean():.0f} (min {o[:, :, i].min():.0f})" for i, n in enumerate(names)))
