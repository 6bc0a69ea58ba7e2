"""Host time of one call (returns when everything is enqueued) against the GPU time per call: is the loop host-bound?"""
import os, sys, time
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
cost = torch.empty(32, device=dev); g1 = torch.empty_like(t1); g2 = torch.empty_like(t2)
st = torch.cuda.current_stream().cuda_stream
def raw():
    _lib.lib.pcc_match_cost(32, 2048, 2048, t1.data_ptr(), t2.data_ptr(), None, cost.data_ptr(), g1.data_ptr(), g2.data_ptr(), st)
for name, fn in (('pcc_match_cost (C call alone)', raw), ('match_cost', lambda: backend.MatchCostImplicit(t1, t2, True)),
                 ('chamfer_emd', lambda: backend.ChamferEMD(t1, t2, True, True)), ('step', step)):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    single = []
    for _ in range(10):  # one call on an idle GPU: no back-pressure from a full queue
        t0 = time.perf_counter(); fn(); single.append((time.perf_counter() - t0) * 1e6); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): fn()
    t_host = (time.perf_counter() - t0) / 10 * 1e6
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(40): fn()
    e.record(); torch.cuda.synchronize()
    print(f'{name}: host {sorted(single)[5]:.0f} us for one call on an idle GPU, {t_host:.0f} us per call for 10 back to back; GPU-timed {s.elapsed_time(e) / 40 * 1e3:.0f} us per call')
