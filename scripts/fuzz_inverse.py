"""Random flow shapes: the two-wave / lone-wave sweeps against the D-pass inverse on the device and against each other.
    python scripts/fuzz_inverse.py [n_cases] [seed]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pocomc_amd import Flow
from pocomc_amd.maf_spec import MAFSpec

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n_cases):
    D = int(rng.integers(2, 65))
    T = int(rng.integers(1, 8))
    H = int(rng.choice([max(D - 1, 4), D + 3, 2 * D, 3 * D + 1, 128, 4 * (D - 1) + 5]))
    H = max(H, D - 1)
    n = int(rng.choice([1, 15, 16, 17, 100, 1000, 5000]))
    try:
        spec = MAFSpec(D, T, hidden=H)
    except Exception as e:                       # (shapes the spec refuses)
        continue
    f = Flow(D, spec, seed=case)
    f.set_params((spec.init_params(case) * np.float32(1.2)).astype(np.float32))
    z = torch.randn(n, D, generator=torch.Generator().manual_seed(case)) * 1.1
    out = {}
    for algo in (2, 6, 7):
        f.inverse_algo = algo
        try:
            out[algo] = [t.numpy() for t in f.inverse(z)]
        except Exception as e:
            out[algo] = None
    if out[6] is None or out[7] is None:
        print(f"case {case}: D={D} T={T} H={H} n={n}: sweeps not available ({'tri' if spec.tri_ok else 'not tri'})")
        continue
    ok_bits = np.array_equal(out[6][0], out[7][0]) and np.array_equal(out[6][1], out[7][1])
    fin = np.isfinite(out[2][0]).all(axis=1) & np.isfinite(out[7][0]).all(axis=1)
    sc = np.maximum(1.0, np.abs(out[2][0][fin]).max(axis=1, keepdims=True)) if fin.any() else 1.0
    err = float((np.abs(out[7][0][fin] - out[2][0][fin]) / sc).max()) if fin.any() else 0.0
    flag = "" if (ok_bits and err < 2e-4) else "   <-- CHECK"
    bad += bool(flag)
    print(f"case {case}: D={D} T={T} H={H} Hp={spec.Hp} n={n}: solo == duo bitwise {ok_bits}, duo vs D-pass {err:.1e} ({int(fin.sum())} finite rows){flag}")
print("suspicious cases:", bad)
