"""Fine-level passes (levels 0-2) at a batch whose resident launch fits the chip: per-kernel hipEvent averages."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import pair
from pointcloudcounterfactual_amd import _lib, backend
L = _lib.lib
if os.environ.get('PCC_AM_NORESIDENT') == '1':  # (tool-side switch; the library itself reads no behaviour variables)
    os.environ['PCC_TEST_HOOKS'] = '1'; _lib.set_tuning('am_noresident', 1)
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
a, c = pair(1236, B, 2048, 2048)
t1, t2 = torch.from_numpy(a).to(dev), torch.from_numpy(c).to(dev)
for _ in range(5): backend.MatchCostImplicit(t1, t2, True)
torch.cuda.synchronize()
L.pcc_profile_enable(1)
for _ in range(10): backend.MatchCostImplicit(t1, t2, True)
torch.cuda.synchronize()
tot = 0.0
for name in [b'am_fine_persist_kernel', b'am_phase_kernel<A> L0'] + [b'am_phase_kernel<B> L%d' % i for i in range(3)] + [b'am_phase_kernel<CA> L%d' % i for i in range(3)]:
    us = ctypes.c_double(); n = ctypes.c_int()
    L.pcc_profile_read(name, ctypes.byref(us), ctypes.byref(n))
    if n.value:
        print(f'  {name.decode():28s} {us.value:7.1f} us x{n.value}'); tot += us.value
print(f'B={B} resident={os.environ.get("PCC_AM_NORESIDENT") != "1"}: fine levels total {tot:.1f} us')
