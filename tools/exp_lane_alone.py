"""How long does ONE half-batch lane take alone on the chip, against the two lanes of the bench batch running together?"""
import os, sys, torch
os.environ['PCC_TEST_HOOKS'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import pair
from pointcloudcounterfactual_amd import _lib, backend
dev = torch.device('cuda:0')
a, c = pair(1236, 32, 2048, 2048)
t1, t2 = torch.from_numpy(a).to(dev), torch.from_numpy(c).to(dev)
def ev(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
h1, h2 = t1[:16].contiguous(), t2[:16].contiguous()
q1, q2 = t1[:8].contiguous(), t2[:8].contiguous()
for rep in range(2):
    _lib.set_tuning('am_nosplit', 0)
    both = ev(lambda: backend.MatchCostImplicit(t1, t2, True))
    _lib.set_tuning('am_nosplit', 1)
    full1 = ev(lambda: backend.MatchCostImplicit(t1, t2, True))
    half = ev(lambda: backend.MatchCostImplicit(h1, h2, True))
    quarter = ev(lambda: backend.MatchCostImplicit(q1, q2, True))
    _lib.set_tuning('am_nosplit', 0)
    print(f'B=32 two lanes {both:.1f} us | B=32 one stream {full1:.1f} us | B=16 one stream (a lane alone) {half:.1f} us | B=8 one stream {quarter:.1f} us')
