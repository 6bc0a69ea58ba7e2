"""k-NN graph timings (c <= 3 sorted search and the MFMA kernel) on Gaussian and on surface-like clouds."""
import sys, os
os.environ.setdefault('PCC_TEST_HOOKS', '1')
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudcounterfactual_amd import neighbour_ops as ops
dev = torch.device('cuda:0')
def ev(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
B, N = 32, 2048
torch.manual_seed(0)
g = torch.randn(B, 3, N, device=dev)
sph = torch.nn.functional.normalize(torch.randn(B, 3, N, device=dev), dim=1)
for name, x in (('gauss', g), ('sphere', sph)):
    for k in (25, 20, 4):
        print(f'knn c=3 k={k} {name}: {ev(lambda: ops.hip_knn(x, k)):.1f} us')
if len(sys.argv) > 1:
    os.environ.setdefault('PCC_TEST_HOOKS', '1')
    from pointcloudcounterfactual_amd import _lib
    for bb in (32, 16, 8):
        for c in (64, 128, 16):
            x = torch.randn(bb, c, N, device=dev)
            t = []
            for nosplit in (0, 1):  # 0: the product's choice; 1: the 128-query kernel everywhere
                _lib.set_tuning('knn_nosplit', nosplit)
                t.append(ev(lambda: ops.hip_knn(x, 25)))
            _lib.set_tuning('knn_nosplit', 0)
            print(f'knn B={bb} c={c} k=25: {t[0]:.1f} us   (128-query kernel: {t[1]:.1f} us)')
    # the reference's CPU formula (torch_knn, neighbour_ops.py:53-74) run by stock PyTorch on the GPU: expanded-form
    # distances through bmm, then topk -- what a user gets without this library
    def torch_knn(x, k):
        inner = -2 * torch.bmm(x.transpose(2, 1), x)
        xx = (x ** 2).sum(dim=1, keepdim=True)
        return (inner + xx + xx.transpose(2, 1)).topk(k, dim=-1, largest=False)[1]
    for c in (3, 64, 128):
        x = torch.randn(B, c, N, device=dev)
        print(f'stock torch bmm+topk c={c} k=25: {ev(lambda: torch_knn(x, 25), iters=5, warm=2):.1f} us')
