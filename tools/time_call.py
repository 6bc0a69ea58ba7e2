import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import pair
from pointcloudcounterfactual_amd import backend, chamfer_emd
what = sys.argv[1] if len(sys.argv) > 1 else 'match_cost'; tag = sys.argv[2] if len(sys.argv) > 2 else ''
dev = torch.device('cuda:0')
a, c = pair(1236, 32, 2048, 2048)
t1, t2 = torch.from_numpy(a).to(dev), torch.from_numpy(c).to(dev)
t1g = t1.clone().requires_grad_(True)
def step():
    t1g.grad = None
    lc, le = chamfer_emd(t1g, t2)
    (lc + le).sum().backward()
fn = {'match_cost': lambda: backend.MatchCostImplicit(t1, t2, True), 'chamfer_emd': lambda: backend.ChamferEMD(t1, t2, True, True), 'step': step,
      'match_cost8': lambda: backend.MatchCostImplicit(t1[:8], t2[:8], True)}[what]
def ev(iters=40, warm=8):
    for _ in range(warm): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
r = sorted(ev() for _ in range(5))
print(f'{tag:4s} {what}: median {r[2]:.1f} us  min {r[0]:.1f}  max {r[-1]:.1f}')
