import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudcounterfactual_amd import neighbour_ops as ops, _lib
from emd import emdModule
dev = torch.device('cuda:0')
def ev(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
B, N = 32, 2048
for c, k in ((3, 25), (3, 4), (64, 25), (128, 25), (64, 20)):
    x = torch.randn(B, c, N, device=dev)
    print(f'knn c={c} k={k}: {ev(lambda: ops.hip_knn(x, k)):.1f} us')
x = torch.randn(B, 64, N, device=dev); idx = ops.hip_knn(x, 25)
print('graph_features c=64 k=25 fwd us', ev(lambda: ops.get_graph_features(x, idx, 25)))
xr = x.clone().requires_grad_(True)
def fb():
    xr.grad = None
    ops.get_graph_features(xr, idx, 25)[1].sum().backward()
print('graph_features c=64 k=25 fwd+bwd us', ev(fb))
print('max_pool c=64 us', ev(lambda: ops.graph_max_pooling(x, idx, 25)))
x2 = torch.randn(B, 1024, N, device=dev)
print('global max pool [32,1024,2048] us', ev(lambda: ops.global_max_pool(x2)), ' torch:', ev(lambda: x2.max(dim=2)))
a = torch.rand(B, N, 3, device=dev); b = torch.rand(B, N, 3, device=dev)
print('auction eps=0.005 iters=50 us', ev(lambda: emdModule()(a, b, 0.005, 50), iters=3, warm=1))
