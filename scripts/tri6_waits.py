"""Measurement build only (scripts/abl_tri6.sh 32; PMC_LIBRARY=scripts/abl/lib6_32.so): how long the chain wavefront of
workgroup 0 waits for the helpers' staged partials per hidden tile of the lane sweep.
    python scripts/tri6_waits.py [n] [precision] [D] [T]
"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pocomc_amd as pc
from pocomc_amd import _lib
from pocomc_amd.maf_spec import MAFSpec
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
D = int(sys.argv[3]) if len(sys.argv) > 3 else 128
T = int(sys.argv[4]) if len(sys.argv) > 4 else 8
f = pc.Flow(D, MAFSpec(D, T), seed=0, inverse_precision=prec, inverse_guard=False)
lib = _lib.load()
fn = lib.pmc_debug_tri6_waits
fn.restype, fn.argtypes = C.c_int, [C.c_void_p]
z = torch.randn(n, D, device="cuda")
out = (C.c_uint64 * 16)()
for _ in range(3):
    f.inverse(z)
torch.cuda.synchronize(); fn(out)
reps = 5
for _ in range(reps):
    f.inverse(z)
torch.cuda.synchronize(); fn(out)
names = ["H0", "H1", "H2", "X", "P0", "P1", "P2", "P3"]
tiles = f.spec.nT * T * reps
print(f"n {n} {prec} D {D} T {T}: hidden tiles per sweep {f.spec.nT * T}")
for i, nm in enumerate(names):
    if out[8 + i]:
        print(f"  chain waited for {nm}: {out[i] / tiles:8.1f} cycles per tile, late in {out[8 + i] / tiles * 100:5.1f} % of the tiles")
