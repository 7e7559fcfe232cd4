"""Host-side timing of Rosenbrock formulations on the GPU box's CPU (the benchmark's black box is numpy on one core)."""
import os, time
import numpy as np
os.sched_setaffinity(0, {sorted(os.sched_getaffinity(0))[4]})
np.setbufsize(1024)
def v_old(x):
    a, b = x[:, ::2], x[:, 1::2]
    t = a * a; t -= b; t *= t; t *= 10.0
    u = a - 1.0; u *= u; t += u
    return -t.sum(axis=1)
def v_einsum(x):
    xT = x.T; a, b = xT[::2], xT[1::2]
    t = np.multiply(a, a); t -= b
    s = np.einsum("ji,ji->i", t, t)
    np.subtract(a, 1.0, out=t); s *= 10.0; s += np.einsum("ji,ji->i", t, t)
    return np.negative(s, out=s)
def v_sq_sum(x):
    xT = x.T; a, b = xT[::2], xT[1::2]
    t = np.multiply(a, a); t -= b; t *= t
    s = t.sum(axis=0)
    np.subtract(a, 1.0, out=t); t *= t; s *= 10.0; s += t.sum(axis=0)
    return np.negative(s, out=s)
_ones = {}
def v_dot(x):
    xT = x.T; a, b = xT[::2], xT[1::2]
    k = a.shape[0]
    o = _ones.setdefault(k, np.ones(k))
    t = np.multiply(a, a); t -= b; t *= t
    s = o @ t
    np.subtract(a, 1.0, out=t); t *= t; s *= 10.0; s += o @ t
    return np.negative(s, out=s)
def v_expand(x):       # sum (a-1)^2 = sum a^2 - 2 sum a + k
    xT = x.T; a, b = xT[::2], xT[1::2]
    t = np.multiply(a, a)
    s2 = t.sum(axis=0)
    t -= b
    s = np.einsum("ji,ji->i", t, t); s *= 10.0
    s += s2; s -= 2.0 * a.sum(axis=0); s += a.shape[0]
    return np.negative(s, out=s)
for n in (5008, 10000):
    x = np.asfortranarray(np.random.default_rng(0).uniform(-10, 10, size=(n, 32)))
    ref = v_old(x)
    for f in (v_old, v_einsum, v_sq_sum, v_dot, v_expand):
        r = f(x)
        ts = []
        for _ in range(400):
            t0 = time.perf_counter(); f(x); ts.append(time.perf_counter() - t0)
        print(n, f.__name__, round(np.median(ts) * 1e6, 1), "us", float(np.abs(r - ref).max() / np.abs(ref).max()))
