import os, sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
import pocomc_amd as pc
from pocomc_amd.maf_spec import MAFSpec
sp = MAFSpec(128, 8)
f = pc.Flow(128, sp, seed=0, precision="bf16")
x = torch.randn(5000, 128, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
for n in (5000, 512):
    xs = x[:n]
    for _ in range(3): f.log_prob(xs)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f.log_prob(xs)
    e1.record(); torch.cuda.synchronize()
    print(os.environ.get("PMC_FWD_BF16_RS", "auto"), n, round(e0.elapsed_time(e1) / 20 * 1e3, 1), "us")
