import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import oracle
from tests.util import pair
from pointcloudcounterfactual_amd import backend
dev = torch.device('cuda:0')
oracle.set_threads(16)
def ev(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us
a, c = pair(1236, 32, 2048, 2048)
t1, t2 = torch.from_numpy(a).to(dev), torch.from_numpy(c).to(dev)
import os
print('nn fwd us', ev(lambda: backend.NNDistance(t1, t2)))
d1, i1, d2, i2 = backend.NNDistance(t1, t2)
g1 = torch.full_like(d1, 1/2048); g2 = torch.full_like(d2, 1/2048)
print('nn bwd us', ev(lambda: backend.NNDistanceGrad(t1, t2, i1, i2, g1, g2)))
print('approxmatch us', ev(lambda: backend.ApproxMatch(t1, t2), iters=5, warm=1))
match, temp = backend.ApproxMatch(t1, t2)
print('approxmatch+cost us', ev(lambda: backend.ApproxMatchCost(t1, t2), iters=5, warm=1))
print('matchcost us', ev(lambda: backend.MatchCost(t1, t2, match), iters=5, warm=1))
print('matchcostgrad us', ev(lambda: backend.MatchCostGrad(t1, t2, match), iters=5, warm=1))
# accuracy vs oracle at B=2 of full size
om, _ = oracle.approxmatch(a[:2], c[:2]); om64, _ = oracle.approxmatch_f64(a[:2], c[:2])
got = match[:2].cpu().numpy()
print('match err ours-vs-f64', np.abs(got-om64).max(), 'oracle-vs-f64', np.abs(om-om64).max())
cost = backend.MatchCost(t1, t2, match)[:2].cpu().numpy()
print('cost ours', cost, 'oracle32', oracle.matchcost(a[:2], c[:2], om), 'f64', oracle.matchcost_f64(a[:2], c[:2], om64))
