import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import pair
from pointcloudcounterfactual_amd import _lib
L = _lib.lib
dev = torch.device('cuda:0')
B, N = 32, 2048
a, c = pair(1236, B, N, N)
t1, t2 = torch.from_numpy(a).to(dev), torch.from_numpy(c).to(dev)
d1 = torch.empty(B, N, device=dev); d2 = torch.empty(B, N, device=dev)
i1 = torch.empty(B, N, device=dev, dtype=torch.int32); i2 = torch.empty(B, N, device=dev, dtype=torch.int32)
g1 = torch.full((B, N), 1 / N, device=dev); g2 = torch.full((B, N), 1 / N, device=dev)
o1 = torch.empty(B, N, 3, device=dev); o2 = torch.empty(B, N, 3, device=dev)
match = torch.empty(B, N, N, device=dev); temp = torch.empty(B, 4 * N, device=dev); cost = torch.empty(B, device=dev)
st = torch.cuda.current_stream().cuda_stream
def ev(fn, iters=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
P = lambda t: t.data_ptr()
which = sys.argv[1] if len(sys.argv) > 1 else 'all'
if which in ('all', 'nn'):
    print('nn fwd us', ev(lambda: L.pcc_nndistance(B, N, P(t1), N, P(t2), P(d1), P(i1), P(d2), P(i2), st)))
    print('nn bwd us', ev(lambda: L.pcc_nndistancegrad(B, N, P(t1), N, P(t2), P(g1), P(i1), P(g2), P(i2), P(o1), P(o2), st)))
if which in ('all', 'am'):
    print('approxmatch us', ev(lambda: L.pcc_approxmatch(B, N, N, P(t1), P(t2), P(match), P(temp), st), iters=10, warm=2))
    print('approxmatch+cost us', ev(lambda: L.pcc_approxmatch_cost(B, N, N, P(t1), P(t2), P(match), P(temp), P(cost), st), iters=10, warm=2))
    print('matchcost us', ev(lambda: L.pcc_matchcost(B, N, N, P(t1), P(t2), P(match), P(cost), st), iters=10, warm=2))
    print('matchcostgrad us', ev(lambda: L.pcc_matchcostgrad(B, N, N, P(t1), P(t2), P(match), P(o1), P(o2), st), iters=10, warm=2))
