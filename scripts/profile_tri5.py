"""Cycle stamps of the chain wave of workgroup 0 of the two-wave inverse sweep (pmc_debug_tri5_profile; measurement only)."""
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
fn = lib.pmc_debug_tri5_profile
fn.restype = C.c_int
fn.argtypes = [C.POINTER(_lib.pmc_maf_t), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
z = torch.randn(n, D, device="cuda")
x = torch.empty_like(z); l = torch.empty(n, device="cuda")
nT, T = f.spec.nT, f.spec.n_transforms
prof = torch.zeros(T * nT + 16, 8, dtype=torch.int64, device="cuda")
for _ in range(3):
    _lib.check(fn(C.byref(f._desc), _lib.ptr(z), _lib.ptr(x), _lib.ptr(l), n, _lib.ptr(prof), _lib.stream_handle()))
torch.cuda.synchronize()
p = prof.cpu().numpy().astype(np.int64)
t0 = p[0, 0]
print(f"D={D} {name} n={n}: chain wave, cycles per tile: tile start (staging reads, sums) | groups | wait E")
hw = p[T * nT:].reshape(-1)[:128].reshape(64, 2)
print('HW_ID (simd = bits 5:4, cu = bits 11:8, se = 15:13) of the chain / burst wave of workgroups 0..11:')
print([(f'cu{(int(a)>>8)&15}.se{(int(a)>>13)&7} simd {(int(a)>>4)&3}/{(int(b)>>4)&3}' + ('' if ((int(a)>>8)&0xff) == ((int(b)>>8)&0xff) else ' (other CU?)')) for a, b in hw[:12]])
p = p[:T * nT]
tot = np.zeros(3)
prev_end = None
for i in range(T * nT):
    if p[i, 0] == 0:
        continue
    d = np.diff(p[i, :4])
    tot += d
    gap = "" if prev_end is None else f"  (since the previous tile's end: {p[i,0]-prev_end})"
    prev_end = p[i, 3]
    print(f"{i:3d} start {p[i,0]-t0:7d} | " + " ".join(f"{v:6d}" for v in d) + gap)
print("sum:", " ".join(f"{int(v):7d}" for v in tot), " total", int(p[:, 3].max() - t0))
