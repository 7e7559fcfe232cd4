"""When the two wavefronts of workgroup 0 of the two-wave spline sweep reach and leave every barrier (a library built with
-DNSF2_TILE_STAMPS: scripts/abl_nsf.sh stamps; measurement only).
    PMC_LIBRARY=scripts/abl/lib_nsf_stamps.so python scripts/profile_nsf2_tiles.py [D] [flow] [n]"""
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
T = f.spec.n_transforms
nTl = int(f.spec.device_meta()[7])
nb = T * (nTl + 1)
prof = torch.zeros(16 + 8 * nb + 64, dtype=torch.int64, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    _lib.check(fn(C.byref(f._desc), _lib.ptr(z), _lib.ptr(x), _lib.ptr(l), n, _lib.ptr(prof), _lib.stream_handle()))
torch.cuda.synchronize()
e0.record()
for _ in range(20):
    _lib.check(fn(C.byref(f._desc), _lib.ptr(z), _lib.ptr(x), _lib.ptr(l), n, _lib.ptr(prof), _lib.stream_handle()))
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
pall = prof.cpu().numpy()
p = pall[16:16 + 4 * nb].reshape(nb, 4)
sec = pall[16 + 4 * nb:16 + 8 * nb].reshape(nb, 4)          # burst wave, per barrier: start of outputs | hidden | layer 0 | eager
t0 = p[0].min()
span = p[-1].max() - t0
print(f"D={D} {name} n={n}: {T} transforms x {nTl} live tiles; launch {us:.1f} us; first to last barrier of workgroup 0: {span} cycles")
print(" barrier   chain: work   wait |  burst: work   wait   (cycles; work = previous leave -> arrive, wait = arrive -> leave)")
tot = np.zeros(4)
for b in range(nb):
    ca, cl, ba, bl = p[b]
    pc_, pb_ = (p[b - 1][1], p[b - 1][3]) if b else (t0, t0)
    row = (ca - pc_, cl - ca, ba - pb_, bl - ba)
    tot += row
    tt, e = divmod(b, nTl + 1)
    print(f"  t{T - 1 - tt} E({e - 1:2d})   {row[0]:7d} {row[1]:6d} | {row[2]:7d} {row[3]:6d}")
print(f"  sum       {int(tot[0]):7d} {int(tot[1]):6d} | {int(tot[2]):7d} {int(tot[3]):6d}")
print(" burst wave by section (cycles): before the tile body | output partials | hidden layers | layer 0 + staging | eager jobs + to the barrier")
for b in range(1, nb):
    if sec[b][0] == 0:
        continue
    tt, e = divmod(b, nTl + 1)
    s0, s1, s2, s3 = sec[b]
    print(f"  t{T - 1 - tt} E({e - 1:2d})   {s0 - p[b - 1][3]:6d} {s1 - s0:7d} {s2 - s1:7d} {s3 - s2:7d} {p[b][2] - s3:7d}")
