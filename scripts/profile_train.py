"""In-kernel phase profile of the loss/gradient kernel (debug entry point, not part of the ABI)."""
import os as _os
# the in-kernel profile entry points exist only in the measurement build: make -C pocomc_amd/csrc DEBUG_HOOKS=1
_os.environ.setdefault("PMC_LIBRARY", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "pocomc_amd", "libpocomc_amd_debug.so"))
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pocomc_amd import Flow, _lib
from pocomc_amd.train import _train_state

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
D = int(sys.argv[2]) if len(sys.argv) > 2 else 32
f = Flow(D, sys.argv[3] if len(sys.argv) > 3 else "maf3", seed=0)
ts = _train_state(f)
ts.repack(f)
ts.ensure_sets(n)
lib = _lib.load()
fn = lib.pmc_debug_lossgrad_profile
fn.restype = C.c_int
fn.argtypes = [C.POINTER(_lib.pmc_maf_t), C.POINTER(_lib.pmc_maf_train_t)] + [C.c_void_p] * 3 + [C.c_int64, C.c_void_p, C.c_void_p]
x = torch.randn(n, D, device="cuda")
nb = min(ts.n_sets, (n + 15) // 16)
NW = lib.pmc_maf_train_waves(C.byref(f._desc))
prof = torch.zeros(nb, NW, 16, dtype=torch.int64, device="cuda")
for _ in range(3):
    _lib.check(fn(C.byref(f._desc), C.byref(ts.desc), _lib.ptr(x), _lib.ptr(ts.grad), _lib.ptr(ts.scal), n, _lib.ptr(prof),
                  _lib.stream_handle()))
torch.cuda.synchronize()
p = prof.cpu().numpy().astype(np.float64)
names = ["total", "load/loss", "fwd hidden", "fwd out", "bwd recompute", "elementwise", "L3 (da2,dW3)", "L2 (da1,dW2)",
         "L1 (da0,dW1)", "L0 (dx,dW0,rerank)", "barrier wait", "fwd spline panel product", "fwd spline evaluation", "hid L1 + L2",
         "hid barriers", "hid L0"]
print("workgroups", nb, "cycles; per wave mean over workgroups")
for i, nm in enumerate(names[:16]):
    print(f"{nm:20s} " + "  ".join(f"w{w}: {p[:, w, i].mean():9.0f}" for w in range(NW)))
