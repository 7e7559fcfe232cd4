import sys, torch
sys.path.insert(0, '/root/repo')
import pocomc_amd as pc
for D, name in ((32,'nsf3'),(10,'nsf6'),(50,'nsf3'),(5,'nsf3'),(64,'nsf3'),(17,'nsf12')):
    f = pc.Flow(D, name, seed=1)
    # random weights so that the output layer is not near zero
    with torch.no_grad():
        f.params.add_(0.05 * torch.randn_like(f.params)) if hasattr(f, 'params') else None
    z = torch.randn(1000, D, device='cuda')
    outs = {}
    for algo in (7, 6, 2):
        f.inverse_algo = algo
        try:
            x, l = f.inverse(z)
            outs[algo] = (x.clone(), l.clone())
        except Exception as e:
            print(D, name, algo, 'ERR', str(e)[:100])
    def rel(a, b): return float(((a - b).abs() / (b.abs() + 1e-3)).max())
    for a in (7, 6):
        if a in outs and 2 in outs:
            print(D, name, 'algo', a, 'vs naive: x', rel(outs[a][0], outs[2][0]), 'ladj', rel(outs[a][1], outs[2][1]))
    if 7 in outs and 6 in outs:
        print(D, name, 'duo vs solo identical?', bool((outs[7][0] == outs[6][0]).all()), rel(outs[7][0], outs[6][0]))
