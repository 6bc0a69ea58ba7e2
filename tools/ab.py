"""A/B of one tuning switch inside ONE process, interleaved (pcc_test_set_tuning; needs PCC_TEST_HOOKS=1):
   PCC_TEST_HOOKS=1 python tools/ab.py <key> [what]      what = match_cost (default) | chamfer_emd | step"""
import os, sys, torch
os.environ.setdefault('PCC_TEST_HOOKS', '1')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import pair
from pointcloudcounterfactual_amd import _lib, backend, chamfer_emd
L = _lib.lib
key = int(sys.argv[1]); what = sys.argv[2] if len(sys.argv) > 2 else 'match_cost'
dev = torch.device('cuda:0')
a, c = pair(1236, 32, 2048, 2048)
t1, t2 = torch.from_numpy(a).to(dev), torch.from_numpy(c).to(dev)
t1g = t1.clone().requires_grad_(True)
def step():
    t1g.grad = None
    lc, le = chamfer_emd(t1g, t2)
    (lc + le).sum().backward()
fn = {'match_cost': lambda: backend.MatchCostImplicit(t1, t2, True), 'chamfer_emd': lambda: backend.ChamferEMD(t1, t2, True, True), 'step': step}[what]
def ev(iters=40, warm=5):
    for _ in range(warm): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
VALUES = tuple(int(x) for x in os.environ.get('AB_VALUES', '0,1').split(','))
res = {v: [] for v in VALUES}
for rep in range(6):
    for v in VALUES:
        assert L.pcc_test_set_tuning(key, v) == 1
        res[v].append(ev())
L.pcc_test_set_tuning(key, 0)
for v in VALUES:
    r = sorted(res[v]); print(f'switch {key} = {v}: median {r[len(r)//2]:.1f} us  min {r[0]:.1f}  max {r[-1]:.1f}   ({what})')
