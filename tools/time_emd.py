"""match_cost forward+backward (pcc_match_cost) and the ChamferEMD node at the bench size: wall per call (hipEvents over
20 calls) and the library's per-kernel averages (pcc_profile_enable(1): one event pair per launch, which serialises the
lanes' kernels a little -- use for the kernels' own durations, not for the call)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import pair
from pointcloudcounterfactual_amd import _lib, backend
L = _lib.lib
dev = torch.device('cuda:0')
B, N = 32, 2048
kind = sys.argv[1] if len(sys.argv) > 1 else 'recon'
a, c = pair(1236, B, N, N, kind)
t1, t2 = torch.from_numpy(a).to(dev), torch.from_numpy(c).to(dev)

def ev(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

for rep in range(3):
    print(f'[{kind}] match_cost fwd+bwd {ev(lambda: backend.MatchCostImplicit(t1, t2, True)):.1f} us | fwd only '
          f'{ev(lambda: backend.MatchCostImplicit(t1, t2, False)):.1f} us | chamfer_emd fwd {ev(lambda: backend.ChamferEMD(t1, t2, True, True)):.1f} us')
L.pcc_profile_enable(1)
for _ in range(5): backend.MatchCostImplicit(t1, t2, True)
torch.cuda.synchronize()
for name in [b'am_sort_kernel', b'am_fine_persist_kernel', b'am_phase_kernel<A> L0'] + [b'am_phase_kernel<B> L%d' % i for i in range(9)] + \
        [b'am_phase_kernel<CA> L%d' % i for i in range(8)] + [b'am_phase_kernel<C> L8', b'am_pair_kernel<grad>', b'pair_finish_kernel']:
    us = ctypes.c_double(); n = ctypes.c_int()
    L.pcc_profile_read(name, ctypes.byref(us), ctypes.byref(n))
    if n.value: print(f'  {name.decode():28s} {us.value:7.1f} us x{n.value}')
L.pcc_profile_enable(0)
cost = backend.MatchCostImplicit(t1, t2, True)[0]
print('cost sum', float(cost.double().sum()))
