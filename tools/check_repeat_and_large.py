"""Run-to-run determinism at the bench shape (300 repeats, bit-identical) and large clouds (up to 30000 points: implicit vs
materialised path, transported mass, nearest-neighbour spot check).  One-off checks, run through gpurun."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import pair
from pointcloudcounterfactual_amd import backend
from pointcloudcounterfactual_amd import neighbour_ops as ops
dev = torch.device('cuda:0')
# 1. run-to-run determinism at the bench shape (fixed summation orders everywhere except the Chamfer backward's LDS atomics)
a, c = pair(5, 32, 2048, 2048)
t1, t2 = torch.from_numpy(a).to(dev), torch.from_numpy(c).to(dev)
ref = [x.clone() for x in backend.ChamferEMD(t1, t2, True, True)]
bad = 0
for it in range(300):
    out = backend.ChamferEMD(t1, t2, True, True)
    if not all(torch.equal(x, y) for x, y in zip(out, ref)): bad += 1
print('bench shape, 300 repeats: differing runs', bad, flush=True)
x = torch.randn(32, 64, 2048, device=dev); p = t1.transpose(1, 2).contiguous()
r1, r2 = ops.hip_knn(x, 25), ops.hip_knn(p, 25)
bad = sum(int(not torch.equal(ops.hip_knn(x, 25), r1)) + int(not torch.equal(ops.hip_knn(p, 25), r2)) for _ in range(50))
print('knn 50 repeats: differing runs', bad, flush=True)
# 2. large clouds: implicit vs materialised, mass conservation
for (b, n, m) in ((2, 16384, 16384), (1, 30000, 9000), (3, 9000, 16384)):
    a, c = pair(9, b, n, m, 'uniform')
    t1, t2 = torch.from_numpy(a).to(dev), torch.from_numpy(c).to(dev)
    t0 = time.time()
    cost, g1, g2 = backend.MatchCostImplicit(t1, t2, True)
    torch.cuda.synchronize(); ti = time.time() - t0
    match, _t, cm = backend.ApproxMatchCost(t1, t2)
    h1, h2 = backend.MatchCostGrad(t1, t2, match)
    rel = float(((cost - cm).abs() / cm.abs()).max())
    gr = float((g1 - h1).abs().max() / h1.abs().max()), float((g2 - h2).abs().max() / h2.abs().max())
    mass = match.sum(dim=(1, 2)).cpu().numpy() / min(n, m) / max(1, max(n, m) // min(n, m))
    d1, i1, d2, i2 = backend.NNDistance(t1, t2)
    # spot check of the nearest neighbour of 64 random queries
    q = torch.randint(0, n, (64,), device=dev)
    dd = ((t1[0, q][:, None, :] - t2[0][None, :, :]) ** 2).sum(-1)
    ok = bool((dd.argmin(1).int() == i1[0, q]).all())
    print(f'b={b} n={n} m={m}: implicit {ti*1e3:.1f} ms; cost rel diff {rel:.2e}; grad rel diff {gr[0]:.2e} {gr[1]:.2e}; mass/expected {mass}; nn spot check {ok}; finite {bool(torch.isfinite(cost).all())}', flush=True)
    del match
