import sys; sys.path.insert(0,'/root/repo')
import torch, numpy as np
import pocomc_amd as pc
for name in ("maf3","nsf3"):
    f = pc.Flow(5, name, seed=0)
    x0 = torch.zeros(0,5)
    z,l = f.forward(x0); print(name,'forward empty', z.shape, l.shape)
    x,l = f.inverse(x0); print(name,'inverse empty', x.shape, l.shape)
    print(name,'log_prob empty', f.log_prob(x0).shape)
    xs,lq = f.sample(0); print(name,'sample 0', xs.shape, lq.shape)
    xs,lq = f.sample(1); print(name,'sample 1', xs.shape, bool(torch.isfinite(xs).all()))
    h = f.fit(torch.randn(3,5), epochs=2); print(name,'fit 3 rows', h['loss'])
    h = f.fit(torch.randn(2,5), epochs=2, validation_split=0.5); print(name,'fit 2 rows split', h['loss'], h['val_loss'])
    try:
        f.forward(torch.zeros(4,6))
    except ValueError as e: print('shape error ok')
    try:
        f.forward(torch.zeros(4,5, dtype=torch.int32))
    except Exception as e: print('dtype error:', type(e).__name__)
