"""Lane count of the approximate-EMD schedule (switch am_lanes of pcc_test_hooks.h), interleaved in one process.
(Round 3 also measured a staggered start -- lane l starting when lane l-1 had finished s passes: 2 lanes 451 -> 498 / 548 /
597 us at s = 3 / 7 / 11, 3 lanes 442 -> 617 / 678 / 744 us at s = 5 / 7 / 9; that code is gone.)"""
import os, sys, torch
os.environ['PCC_TEST_HOOKS'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import pair
from pointcloudcounterfactual_amd import _lib, backend
L = _lib.lib
dev = torch.device('cuda:0')
a, c = pair(1236, 32, 2048, 2048)
t1, t2 = torch.from_numpy(a).to(dev), torch.from_numpy(c).to(dev)
what = sys.argv[1] if len(sys.argv) > 1 else 'match_cost'
fn = (lambda: backend.MatchCostImplicit(t1, t2, True)) if what == 'match_cost' else (lambda: backend.ChamferEMD(t1, t2, True, True))
def ev(iters=30, warm=5):
    for _ in range(warm): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
ref = fn()
configs = [(1, 0), (2, 0), (3, 0), (4, 0)]
res = {cfg: [] for cfg in configs}
for rep in range(3):
    for cfg in configs:
        L.pcc_test_set_tuning(9, cfg[0])
        out = fn(); torch.cuda.synchronize()
        # (lane sizes change the workgroup shape of some passes -- one or two accumulators per owner -- hence the summation
        # order: equal up to rounding, not bitwise)
        assert torch.allclose(ref[0], out[0], rtol=1e-5), cfg
        res[cfg].append(ev())
L.pcc_test_set_tuning(9, 0)
for cfg in configs:
    r = sorted(res[cfg]); print(f'lanes {cfg[0]}: median {r[1]:.1f} us  (min {r[0]:.1f} max {r[2]:.1f})   [{what}]')
