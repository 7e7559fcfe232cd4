"""Timing-only ablations of the inverse sweep (debug entry point, results are wrong by design)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pocomc_amd import Flow, _lib
n, D = 10000, 32
NSF = len(sys.argv) > 1 and sys.argv[1] == "nsf"
f = Flow(D, "nsf3" if NSF else "maf3", seed=0)
lib = _lib.load()
fn = lib.pmc_debug_inverse_nsf_ablate if NSF else lib.pmc_debug_inverse4_ablate
fn.restype = C.c_int
fn.argtypes = [C.POINTER(_lib.pmc_maf_t)] + [C.c_void_p] * 3 + [C.c_int64, C.c_int, C.c_void_p]
z = torch.randn(n, D, device="cuda"); x = torch.empty_like(z); l = torch.empty(n, device="cuda")
names = {0: "full", 1: "no right-looking out updates", 2: "no non-critical hidden updates", 3: "1+2", 4: "no bursts",
         8: "no chain", 16: "no next-tile prefetch", 32: "no tile-top fragment loads", 64: "no exp in x update",
         12: "no bursts, no chain", 21: "no bursts / prefetch / right-looking out (a chain wave's share)", 20: "no bursts / prefetch", 5: "no bursts / right-looking out", 60: "no bursts/chain/prefetch/frag loads", 127: "everything off"}
if NSF:
    names = {0: "full", 1: "no spline solve", 2: "no output product", 4: "no hidden chain", 8: "no output fragment loads",
             3: "no spline, no output product", 7: "no spline/output/chain", 15: "everything off"}
for abl, nm in names.items():
    for _ in range(3):
        _lib.check(fn(C.byref(f._desc), _lib.ptr(z), _lib.ptr(x), _lib.ptr(l), n, abl, _lib.stream_handle()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn(C.byref(f._desc), _lib.ptr(z), _lib.ptr(x), _lib.ptr(l), n, abl, _lib.stream_handle())
    e1.record(); torch.cuda.synchronize()
    print(f"ABL {abl:3d}  {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us   {nm}")
