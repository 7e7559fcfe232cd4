"""Cycles of the chain wave of workgroup 0 of the two-wave spline sweep, by section of a degree group, summed over the
sweep (pmc_debug_nsf2_profile; measurement only).   python scripts/profile_nsf2.py [D] [flow] [n]"""
import os as _os
# the in-kernel profile entry points exist only in the measurement build: make -C pocomc_amd/csrc DEBUG_HOOKS=1
_os.environ.setdefault("PMC_LIBRARY", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "pocomc_amd", "libpocomc_amd_debug.so"))
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pocomc_amd as pc
from pocomc_amd import _lib
D = int(sys.argv[1]) if len(sys.argv) > 1 else 32
name = sys.argv[2] if len(sys.argv) > 2 else "nsf3"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 7008
f = pc.Flow(D, name, seed=0)
lib = _lib.load()
fn = lib.pmc_debug_nsf2_profile
fn.restype = C.c_int
fn.argtypes = [C.POINTER(_lib.pmc_maf_t), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
z = torch.randn(n, D, device="cuda")
x = torch.empty_like(z); l = torch.empty(n, device="cuda")
prof = torch.zeros(16, dtype=torch.int64, device="cuda")
for _ in range(3):
    _lib.check(fn(C.byref(f._desc), _lib.ptr(z), _lib.ptr(x), _lib.ptr(l), n, _lib.ptr(prof), _lib.stream_handle()))
torch.cuda.synchronize()
p = prof.cpu().numpy()
ranks = f.spec.n_transforms * (D - 1)
tiles = f.spec.n_transforms * int(f.spec.device_meta()[7])
names = ["-", "-", "hidden hops 1, 2 (+ the previous rank's side effects in their shadows)", "own-quad output MFMAs",
         "exchange + spline (+ the following rank's MFMAs in its slots)", "rank-1 update of the following quads", "(after the last group: its side effects)",
         "(group entry)", "h stores", "barrier E", "operand set copy"]
print(f"D={D} {name} n={n}: {ranks} ranks in {tiles} tiles; chain wave of workgroup 0, cycles")
for i, nm in enumerate(names):
    per = ranks if i < 6 or i == 7 else tiles
    print(f"  {nm:60s} {int(p[i]):9d}   {p[i] / per:8.1f} per {'rank' if per == ranks else 'tile'}")
print(f"  sum {int(p[:11].sum())}")
