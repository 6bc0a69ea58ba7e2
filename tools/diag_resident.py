import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import pair
from pointcloudcounterfactual_amd import backend
dev = torch.device('cuda:0')
a, c = pair(1236, 32, 2048, 2048)
t1, t2 = torch.from_numpy(a).to(dev), torch.from_numpy(c).to(dev)
def run(name, fn, k):
    bad = 0
    t0 = time.perf_counter()
    for i in range(k):
        try:
            out = fn()
            torch.cuda.synchronize()
            if not torch.isfinite(out).all(): bad += 1; print(name, i, 'non-finite', flush=True)
        except RuntimeError as e:
            bad += 1; print(name, i, 'ERR', str(e)[-160:], flush=True)
    print(f'{name}: {k} calls, {bad} bad, {(time.perf_counter() - t0) / k * 1e6:.0f} us/call (synchronised)', flush=True)
run('match_cost', lambda: backend.MatchCostImplicit(t1, t2, True)[0], 50)
run('chamfer_emd', lambda: backend.ChamferEMD(t1, t2, True, True)[3], 50)
run('match_cost b=5', lambda: backend.MatchCostImplicit(t1[:5].contiguous(), t2[:5].contiguous(), True)[0], 20)
