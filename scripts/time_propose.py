"""Time of pmc_propose (one-wavefront kernel / four wavefronts per 16 walkers for D > 64: PMC_PROPOSE_SOLO=1 / 0).

    python scripts/time_propose.py [D] [N]
"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pocomc_amd import _lib
lib = _lib.load()
D = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
rng = np.random.default_rng(0)
A = rng.normal(size=(D, D)); cov = A @ A.T / D + np.eye(D)
up = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dt).cuda()
mu, icov, chol = up(rng.normal(size=D)), up(np.linalg.inv(cov)), up(np.linalg.cholesky(cov))
cur32 = up(rng.normal(size=(N, D)), torch.float32)
g, z = up(rng.gamma(60.0, size=N)), up(rng.normal(size=(N, D)))
mk = lambda *s, dt=torch.float64: torch.zeros(*s, dtype=dt, device="cuda")
t64, t32, qa, qb = mk(N, D), mk(N, D, dt=torch.float32), mk(N), mk(N)
for label, replay in (("Philox in the kernel", False), ("variates given", True)):
    r = _lib.pmc_rng_t(gamma=g.data_ptr() if replay else None, normal=z.data_ptr() if replay else None, uniform=None, seed=99, step=3, offset=11)
    call = lambda: _lib.check(lib.pmc_propose(0, _lib.ptr(cur32), None, _lib.ptr(mu), _lib.ptr(icov), _lib.ptr(chol), 5.0, 0.3,
                                              float((1 - 0.3 ** 2) ** 0.5), C.byref(r), _lib.ptr(t64), _lib.ptr(t32), _lib.ptr(qa), _lib.ptr(qb),
                                              N, D, _lib.stream_handle()))
    for _ in range(5): call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): call()
    e1.record(); torch.cuda.synchronize()
    print(f"D={D} N={N} solo={os.environ.get('PMC_PROPOSE_SOLO', '0')} {label}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us")
