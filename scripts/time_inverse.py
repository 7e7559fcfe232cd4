"""A/B of the flow-inverse sweeps on the device: agreement with zuko's D-pass algorithm (on the device) and time per launch.

    python scripts/time_inverse.py [D] [flow] [n ...]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import pocomc_amd as pc

D = int(sys.argv[1]) if len(sys.argv) > 1 else 32
name = sys.argv[2] if len(sys.argv) > 2 else "maf3"
ns = [int(v) for v in sys.argv[3:]] or [16, 1000, 5008, 10000]
ALGOS = {"lane(8)": 8, "solo(6)": 6, "duo(7)": 7, "auto(0)": 0}
if name.startswith("custom"):
    from pocomc_amd.maf_spec import MAFSpec
    f = pc.Flow(D, MAFSpec(D, int(name[6:])), seed=0)
    ALGOS = {"lane(8)": 8, "auto(0)": 0, "v2(4)": 4}
else:
    f = pc.Flow(D, name, seed=0)
f.set_params(f.params.cpu().numpy() * np.float32(1.15))
for n in ns:
    z = torch.randn(n, D, generator=torch.Generator().manual_seed(n)) * 1.2
    f.inverse_algo = 2
    xr, lr = f.inverse(z)
    line = f"D={D} {name} n={n}:"
    for label, a in ALGOS.items():
        f.inverse_algo = a
        try:
            x, l = f.inverse(z)
        except Exception as e:
            line += f"  {label}: {str(e)[:40]}"
            continue
        ex = ((x - xr).abs().max(dim=1).values / xr.abs().max(dim=1).values.clamp_min(1e-30)).max().item()
        el = (l - lr).abs().max().item()
        zc = z.cuda()
        for _ in range(3):
            f.inverse(zc)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            f.inverse(zc)
        e1.record()
        torch.cuda.synchronize()
        line += f"  {label}: {e0.elapsed_time(e1) / reps * 1e3:7.1f} us (x {ex:.1e}, ladj {el:.1e})"
    print(line, flush=True)
