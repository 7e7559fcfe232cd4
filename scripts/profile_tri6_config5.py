import os as _os
# the in-kernel profile entry points exist only in the measurement build: make -C pocomc_amd/csrc DEBUG_HOOKS=1
_os.environ.setdefault("PMC_LIBRARY", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "pocomc_amd", "libpocomc_amd_debug.so"))
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pocomc_amd as pc
from pocomc_amd import _lib
from pocomc_amd.maf_spec import MAFSpec
# python scripts/profile_tri6_config5.py [n] [inverse_precision] [D] [T]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
prec = sys.argv[2] if len(sys.argv) > 2 else "f32"
D = int(sys.argv[3]) if len(sys.argv) > 3 else 128
T = int(sys.argv[4]) if len(sys.argv) > 4 else 8
f = pc.Flow(D, MAFSpec(D, T), seed=0, inverse_precision=prec)
print("n", n, "helpers", prec)
lib = _lib.load()
fn = lib.pmc_debug_tri6_profile
fn.restype = C.c_int
fn.argtypes = [C.POINTER(_lib.pmc_maf_t), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
z = torch.randn(n, D, device="cuda")
x = torch.empty_like(z); l = torch.empty(n, device="cuda")
nT = f.spec.nT
prof = torch.zeros(T * nT * 16 + T * nT * 8, dtype=torch.int64, device="cuda")
for _ in range(3):
    _lib.check(fn(C.byref(f._desc), _lib.ptr(z), _lib.ptr(x), _lib.ptr(l), n, _lib.ptr(prof), _lib.stream_handle()))
torch.cuda.synchronize()
pall = prof.cpu().numpy().astype(np.int64)
p = pall[:T * nT * 16].reshape(T * nT, 4, 4)
pe = pall[T * nT * 16:].reshape(T * nT, 8)
t0 = p[0, 0, 0]
print("nT", nT, "Hp", f.spec.Hp)
hw = [int(p[0, 0, 3])] + [int(p[0, w, 2]) for w in (1, 2, 3)]
print("HW_ID per wave:", [hex(h) for h in hw], " simd:", [(h >> 4) & 3 for h in hw], " cu:", [(h >> 8) & 15 for h in hw])
for i in range(0, nT + 2):
    c = p[i, 0] - t0
    row = f"{i:3d} | chain {c[0]:7d} P0wait {c[1] - c[0]:5d} len {c[2] - c[0]:6d} |"
    for w in (1, 2, 3):
        h = p[i, w] - t0
        if w < 3:
            row += f" h{w} {h[0]:7d} w {max(min(h[1] - h[0], 99999), -1):5d} need {max(min(h[2] - h[1], 99999), -1):5d} len {max(min(h[3] - h[0], 99999), -1):6d} |"
        else:       # wave 3: [2] = prefixes done, [1] = output partials published, [3] = next tile's layer-0 partial published
            row += f" h3 {h[0]:7d} pre {max(min(h[2] - h[0], 99999), -1):5d} out {max(min(h[1] - h[0], 99999), -1):5d} len {max(min(h[3] - h[0], 99999), -1):6d} |"
    e = pe[i] - t0
    h0 = p[i, 3, 0] - t0
    row += f" w3: need@{e[0]-h0:6d} l0 issue@{e[4]-h0:6d} x@{e[5]-h0:6d} mma@{e[1]-h0:6d} ahead_end@{e[3]-h0:6d}"
    print(row)
print("total cycles chain:", p[:, 0, 2].max() - t0, " per transform", (p[:, 0, 2].max() - t0) // T)
