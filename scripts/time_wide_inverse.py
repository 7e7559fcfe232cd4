import sys, time
sys.path.insert(0, "/root/repo")
import torch
import pocomc_amd as pc
from pocomc_amd.maf_spec import MAFSpec
for (D, T, n) in ((128, 1, 4096), (128, 8, 4096), (128, 8, 5000)):
    f = pc.Flow(D, MAFSpec(D, T), seed=0)
    z = torch.randn(n, D, device="cuda")
    f.inverse_algo = 0
    x, l = f.inverse(z)
    f.inverse_algo = 2
    x2, l2 = f.inverse(z[:16])
    err = (x[:16] - x2).abs().max().item() / max(1.0, x2.abs().max().item())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    f.inverse_algo = 0
    e0.record()
    for _ in range(5):
        f.inverse(z)
    e1.record(); torch.cuda.synchronize()
    print(f"D={D} T={T} n={n}: {e0.elapsed_time(e1) / 5 * 1e3:8.1f} us  (vs D-pass on 16 rows: {err:.1e})")
