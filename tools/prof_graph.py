import sys, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
from pointcloudcounterfactual_amd import neighbour_ops as ops
dev = torch.device('cuda:0')
B, N = 32, 2048
x = torch.randn(B, 64, N, device=dev); idx = ops.hip_knn(x, 25)
xr = x.clone().requires_grad_(True)
for _ in range(3):
    xr.grad = None
    ops.get_graph_features(xr, idx, 25)[1].sum().backward()
    ops.graph_max_pooling(xr, idx, 25).sum().backward()
x3 = torch.randn(B, 3, N, device=dev)
for _ in range(3):
    ops.hip_knn(x3, 25); ops.hip_knn(x, 25)
torch.cuda.synchronize()
