"""What the proposal prologue of the fused proposal + inverse launch costs (pmc_propose_inverse against pmc_maf_inverse), with
Philox variates and with replayed variates (no random-number work in the kernel); measurement only."""
import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pocomc_amd as pc
from pocomc_amd import _lib
lib = _lib.load()
D = 32
flow = pc.Flow(D, sys.argv[1] if len(sys.argv) > 1 else "maf3", seed=1)
rng = np.random.default_rng(D)
A = rng.normal(size=(D, D)); cov = A @ A.T / D + np.eye(D)
up = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dt).cuda()
mu, icov, chol = up(rng.normal(size=D)), up(np.linalg.inv(cov)), up(np.linalg.cholesky(cov))
st = _lib.stream_handle()
def timed(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for N in (16, 1000, 4096, 6496):
    cur32 = up(rng.normal(size=(N, D)), torch.float32)
    mk = lambda *s, dt=torch.float64: torch.zeros(*s, dtype=dt, device="cuda")
    t64, qa, qb = mk(N, D), mk(N), mk(N)
    u, l = mk(N, D, dt=torch.float32), mk(N, dt=torch.float32)
    gam, nor = up(rng.gamma(18.5, size=N)), up(rng.normal(size=(N, D)))
    r_ph = _lib.pmc_rng_t(gamma=None, normal=None, uniform=None, seed=1234, step=7, offset=5)
    r_rp = _lib.pmc_rng_t(gamma=_lib.ptr(gam), normal=_lib.ptr(nor), uniform=None, seed=1234, step=7, offset=5)
    def fused(r):
        return lambda: _lib.check(lib.pmc_propose_inverse(0, _lib.ptr(cur32), _lib.ptr(mu), _lib.ptr(icov), _lib.ptr(chol), 5.0, 0.4, float((1 - 0.4 ** 2) ** 0.5), C.byref(r), _lib.ptr(t64), _lib.ptr(qa), _lib.ptr(qb), C.byref(flow._desc), _lib.ptr(u), _lib.ptr(l), N, st))
    plain = lambda: _lib.check(lib.pmc_maf_inverse(C.byref(flow._desc), _lib.ptr(cur32), _lib.ptr(u), _lib.ptr(l), N, 0, st))
    a, b, c = timed(plain), timed(fused(r_ph)), timed(fused(r_rp))
    print(f"n={N:5d}: plain sweep {a:6.1f} us | proposal + sweep {b:6.1f} (+{b - a:4.1f}) | with replayed variates {c:6.1f} (+{c - a:4.1f})")
