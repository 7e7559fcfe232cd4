"""Timing of the pool-level calls at BASELINE config 4's size: 8 iterations x 1e4 particles."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pocomc_amd import tools
T, N = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 10000
rng = np.random.default_rng(0)
logl = rng.normal(size=(T, N)) * 3 - 20
beta = np.sort(rng.uniform(0, 1, size=T)); logz = np.cumsum(rng.normal(size=T))


def tm(f, reps=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): r = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, r


ms, (logw, lz) = tm(lambda: tools.compute_logw_and_logz(logl, beta, logz, 0.7))
print(f"compute_logw_and_logz  P={T * N}: {ms:8.3f} ms per call (host arrays in, host arrays out)")
w = np.exp(logw - logw.max()); w /= w.sum()
samples = rng.normal(size=(T * N, 4))
ms, _ = tm(lambda: tools.trim_weights(np.arange(T * N), w, ess=0.99, bins=1000))
print(f"trim_weights                     : {ms:8.3f} ms")
ms, _ = tm(lambda: tools.compute_ess(logw))
print(f"compute_ess                      : {ms:8.3f} ms")
ms, _ = tm(lambda: tools.systematic_resample(N, w, offset=0.3))
print(f"systematic_resample -> {N}     : {ms:8.3f} ms")
ms, _ = tm(lambda: tools.multinomial_resample(N, w, uniforms=rng.uniform(size=N)))
print(f"multinomial_resample -> {N}    : {ms:8.3f} ms")
