import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudcounterfactual_amd import neighbour_ops as ops
from pointcloudcounterfactual_amd.edgeconv import neighbour_sum
dev = torch.device('cuda:0')
def ev(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
B, N, K = 32, 2048, 25
for c in (64, 128, 256):
    x = torch.randn(B, c, N, device=dev)
    idx = ops.hip_knn(x[:, :64].contiguous(), K)
    y = torch.randn(B, c, N, device=dev, requires_grad=True)
    out = neighbour_sum(y, idx)
    g = torch.randn_like(out)
    def bwd():
        y.grad = None
        out.backward(g, retain_graph=True)
    print(f'neighbour_sum c={c}: fwd {ev(lambda: neighbour_sum(y, idx)):.1f} us, bwd {ev(bwd):.1f} us')
