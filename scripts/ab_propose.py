"""A/B of the fused proposal + inverse launch between two builds (PMC_LIBRARY): timing and an .npz of theta', the quadratic
form and u' for a bitwise comparison; measurement only.   python scripts/ab_propose.py out.npz"""
import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, "/root/repo")
import pocomc_amd as pc
from pocomc_amd import _lib
lib = _lib.load()
out = {}
for D, N in ((32, 7008), (10, 1000), (50, 3000), (7, 333), (31, 100)):
    rng = np.random.default_rng(D)
    flow = pc.Flow(D, "maf3", seed=1)
    A = rng.normal(size=(D, D)); cov = A @ A.T / D + np.eye(D)
    up = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dt).cuda()
    mu, icov, chol = up(rng.normal(size=D)), up(np.linalg.inv(cov)), up(np.linalg.cholesky(cov))
    cur32 = up(rng.normal(size=(N, D)), torch.float32)
    r = _lib.pmc_rng_t(gamma=None, normal=None, uniform=None, seed=1234, step=7, offset=5)
    mk = lambda *s, dt=torch.float64: torch.zeros(*s, dtype=dt, device="cuda")
    st = _lib.stream_handle()
    t64, qa, qb = mk(N, D), mk(N), mk(N)
    u, l = mk(N, D, dt=torch.float32), mk(N, dt=torch.float32)
    for _ in range(3):
        _lib.check(lib.pmc_propose_inverse(0, _lib.ptr(cur32), _lib.ptr(mu), _lib.ptr(icov), _lib.ptr(chol), 5.0, 0.4, float((1 - 0.4 ** 2) ** 0.5), C.byref(r), _lib.ptr(t64), _lib.ptr(qa), _lib.ptr(qb), C.byref(flow._desc), _lib.ptr(u), _lib.ptr(l), N, st))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        lib.pmc_propose_inverse(0, _lib.ptr(cur32), _lib.ptr(mu), _lib.ptr(icov), _lib.ptr(chol), 5.0, 0.4, float((1 - 0.4 ** 2) ** 0.5), C.byref(r), _lib.ptr(t64), _lib.ptr(qa), _lib.ptr(qb), C.byref(flow._desc), _lib.ptr(u), _lib.ptr(l), N, st)
    e1.record(); torch.cuda.synchronize()
    print(D, N, "fused propose+inverse us:", round(e0.elapsed_time(e1) / 20 * 1e3, 1))
    out[f"t{D}"] = t64.cpu().numpy(); out[f"q{D}"] = qb.cpu().numpy(); out[f"u{D}"] = u.cpu().numpy()
np.savez(sys.argv[1], **out)
