"""A/B inside one process: pcc_chamfer_emd with the odd lane's nearest-neighbour search behind its passes (0) or every
lane's search right behind its sort (1)."""
import os, sys
os.environ.setdefault('PCC_TEST_HOOKS', '1')
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import pair
from pointcloudcounterfactual_amd import backend, chamfer_emd, _lib
dev = torch.device('cuda:0')
a, c = pair(1236, 32, 2048, 2048)
t1, t2 = torch.from_numpy(a).to(dev), torch.from_numpy(c).to(dev)
t1g = t1.clone().requires_grad_(True)
def step():
    t1g.grad = None
    lc, le = chamfer_emd(t1g, t2)
    (lc + le).sum().backward()
def ev(fn, iters=40, warm=8):
    for _ in range(warm): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for what, fn in (('chamfer_emd', lambda: backend.ChamferEMD(t1, t2, True, True)), ('step', step)):
    for rep in range(3):
        for v in (0, 1):
            _lib.set_tuning('nn_head', v)
            r = sorted(ev(fn) for _ in range(5))
            print(f'nn_head={v} {what}: median {r[2]:.1f} us  min {r[0]:.1f}  max {r[-1]:.1f}')
_lib.set_tuning('nn_head', 0)
