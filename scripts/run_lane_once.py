import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import pocomc_amd as pc
from pocomc_amd.maf_spec import MAFSpec
prec = sys.argv[1]; n = int(sys.argv[2])
f = pc.Flow(128, MAFSpec(128, 8), seed=0, inverse_precision=prec)
z = torch.randn(n, 128).cuda()
for _ in range(5): f.inverse(z)
torch.cuda.synchronize()
