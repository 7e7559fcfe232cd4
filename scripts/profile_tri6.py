"""Cycle stamps of workgroup 0 of the lane-per-walker inverse sweep (pmc_debug_tri6_profile; measurement only)."""
import os as _os
# the in-kernel profile entry points exist only in the measurement build: make -C pocomc_amd/csrc DEBUG_HOOKS=1
_os.environ.setdefault("PMC_LIBRARY", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "pocomc_amd", "libpocomc_amd_debug.so"))
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pocomc_amd as pc
from pocomc_amd import _lib
D = int(sys.argv[1]) if len(sys.argv) > 1 else 32
name = sys.argv[2] if len(sys.argv) > 2 else "maf3"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 5008
f = pc.Flow(D, name, seed=0)
lib = _lib.load()
fn = lib.pmc_debug_tri6_profile
fn.restype = C.c_int
fn.argtypes = [C.POINTER(_lib.pmc_maf_t), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
z = torch.randn(n, D, device="cuda")
x = torch.empty_like(z); l = torch.empty(n, device="cuda")
nT, T = f.spec.nT, f.spec.n_transforms
prof = torch.zeros(T * nT, 4, 4, dtype=torch.int64, device="cuda")
for _ in range(3):
    _lib.check(fn(C.byref(f._desc), _lib.ptr(z), _lib.ptr(x), _lib.ptr(l), n, _lib.ptr(prof), _lib.stream_handle()))
torch.cuda.synchronize()
p = prof.cpu().numpy().astype(np.int64)
t0 = p[0, 0, 0]
print(f"D={D} {name} n={n}: nT={nT}; cycles relative to the chain's first tile")
print("tile | chain: start  P0-wait-done  end (len) | hA: start urgent-wait need-done end | hB ... | hC: start P3-done - P0next-done")
for i in range(T * nT):
    if p[i, 0, 0] == 0:
        continue
    c = p[i, 0] - t0
    row = f"{i:3d} | {c[0]:7d} {c[1] - c[0]:5d} {c[2] - c[0]:6d} |"
    for w in (1, 2, 3):
        h = p[i, w] - t0
        row += f" {h[0]:7d} {h[1] - h[0]:5d} {h[2] - h[1]:5d} {h[3] - h[0]:6d} |"
    print(row)
print("total cycles chain:", p[:, 0, 2].max() - t0)
