import sys, os, ctypes, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import pair
from pointcloudcounterfactual_amd import _lib
L = _lib.lib
dev = torch.device('cuda:0')
B, N = 32, 2048
a, c = pair(1236, B, N, N)
t1, t2 = torch.from_numpy(a).to(dev), torch.from_numpy(c).to(dev)
match = torch.empty(B, N, N, device=dev); temp = torch.empty(B, 4 * N, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(2): L.pcc_approxmatch(B, N, N, t1.data_ptr(), t2.data_ptr(), match.data_ptr(), temp.data_ptr(), st)
torch.cuda.synchronize()
L.pcc_profile_enable(1)
for _ in range(5): L.pcc_approxmatch(B, N, N, t1.data_ptr(), t2.data_ptr(), match.data_ptr(), temp.data_ptr(), st)
torch.cuda.synchronize()
names = [b'am_phase_kernel<A>'] + [b'am_phase_kernel<B> L%d' % i for i in range(9)] + [b'am_phase_kernel<CA> L%d' % i for i in range(8)] + [b'am_phase_kernel<C>', b'am_phase_kernel', b'am_sort', b'am_materialise']
for name in names:
    us = ctypes.c_double(); n = ctypes.c_int()
    L.pcc_profile_read(name, ctypes.byref(us), ctypes.byref(n))
    print(name.decode(), f'{us.value:.1f} us x{n.value}')
print('cost check', float(match.sum()))
import time
cost = torch.empty(B, device=dev); g1 = torch.empty(B, N, 3, device=dev); g2 = torch.empty(B, N, 3, device=dev)
for grad in (False, True):
    L.pcc_profile_enable(0)
    args = (B, N, N, t1.data_ptr(), t2.data_ptr(), None, cost.data_ptr(), g1.data_ptr() if grad else None, g2.data_ptr() if grad else None, st)
    for _ in range(3): L.pcc_match_cost(*args)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): L.pcc_match_cost(*args)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    L.pcc_profile_enable(1)
    for _ in range(5): L.pcc_match_cost(*args)
    torch.cuda.synchronize()
    us = ctypes.c_double(); n = ctypes.c_int()
    L.pcc_profile_read(b'am_pair_kernel', ctypes.byref(us), ctypes.byref(n))
    print(f'pcc_match_cost grad={grad}: {dt*1e6:.1f} us per call; am_pair_kernel {us.value:.1f} us x{n.value}; cost sum {float(cost.sum()):.6f}')

