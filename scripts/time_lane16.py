"""The lane-per-walker inverse sweep with 16-bit helper operands (Flow(inverse_precision="bf16" | "f16")) against the
float32 sweep: time per launch and distance to the float32 result / to zuko's D-pass algorithm on the device.

    python scripts/time_lane16.py [D] [T] [n ...]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import pocomc_amd as pc
from pocomc_amd.maf_spec import MAFSpec

D = int(sys.argv[1]) if len(sys.argv) > 1 else 128
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ns = [int(v) for v in sys.argv[3:]] or [512, 4096, 5000]
spec = MAFSpec(D, T)
flows = {p: pc.Flow(D, spec, seed=0, inverse_precision=p, inverse_guard=False) for p in ("f32", "bf16", "f16")}
par = flows["f32"].params.cpu().numpy() * np.float32(1.15)
for f in flows.values():
    f.set_params(par)
for n in ns:
    z = torch.randn(n, D, generator=torch.Generator().manual_seed(n)) * 1.2
    ref = None
    if n <= 1024:
        flows["f32"].inverse_algo = 2
        ref = flows["f32"].inverse(z)
        flows["f32"].inverse_algo = 0
    x32, l32 = flows["f32"].inverse(z)
    line = f"D={D} T={T} n={n}:"
    for p, f in flows.items():
        x, l = f.inverse(z)
        ex = ((x - x32).abs().max(dim=1).values / x32.abs().max(dim=1).values.clamp_min(1e-30))
        el = (l - l32).abs()
        zc = z.cuda()
        for _ in range(3):
            f.inverse(zc)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            f.inverse(zc)
        e1.record()
        torch.cuda.synchronize()
        line += f"\n   {p}: {e0.elapsed_time(e1) / reps * 1e3:7.1f} us; vs f32 sweep: x max {ex.max().item():.1e} median {ex.median().item():.1e}, ladj max {el.max().item():.1e} median {el.median().item():.1e}"
        if ref is not None:
            er = ((x - ref[0]).abs().max(dim=1).values / ref[0].abs().max(dim=1).values.clamp_min(1e-30)).max().item()
            line += f"; vs D-pass: x {er:.1e}, ladj {(l - ref[1]).abs().max().item():.1e}"
    print(line, flush=True)
