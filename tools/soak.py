"""Soak run: repeated calls with changing shapes (allocator pools, side stream, cluster barriers) -- prints memory high-water marks."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudcounterfactual_amd import backend
from pointcloudcounterfactual_amd.losses import chamfer, chamfer_emd, match_cost
from pointcloudcounterfactual_amd import neighbour_ops as ops
from emd import emdModule
dev = torch.device('cuda:0')
rng = np.random.default_rng(0)
t0 = time.time()
free0, total = torch.cuda.mem_get_info()
for it in range(300):
    b = int(rng.choice([1, 4, 9, 32])); n = int(rng.choice([64, 500, 1024, 2048, 3000])); m = int(rng.choice([64, 700, 2048]))
    x = torch.rand(b, n, 3, device=dev, requires_grad=True); y = torch.rand(b, m, 3, device=dev)
    (chamfer(x, y) + match_cost(x, y)).sum().backward()
    assert torch.isfinite(x.grad).all()
    # encoder primitives on the same changing shapes: sorted k-NN search (c <= 3), staged index rows in the graph kernels
    k = int(rng.integers(1, min(n, 32) + 1))
    pts = x.detach().transpose(1, 2).contiguous()
    idx = ops.hip_knn(pts, k)
    assert int(idx.min()) >= 0 and int(idx.max()) < n and bool((idx[:, :, 0] == torch.arange(n, device=dev)).all())
    f = torch.randn(b, 16, n, device=dev)
    if it % 10 == 0:  # feature-space k-NN: both kernels (the role-split one takes the large launches), against each other
        fk = torch.randn(32 if it % 20 == 0 else b, 24, n, device=dev)
        i1 = ops.hip_knn(fk, k)
        assert int(i1.min()) >= 0 and int(i1.max()) < n
    assert torch.isfinite(ops.graph_max_pooling(f, idx, k)).all()
    lc, le = chamfer_emd(x, y)
    assert torch.isfinite(lc + le).all()
    if it % 25 == 0:
        a = torch.rand(8, 1024, 3, device=dev); c = torch.rand(8, 1024, 3, device=dev)
        d, _ = emdModule()(a, c, 0.005, 30)
        assert torch.isfinite(d).all()
        torch.cuda.synchronize()
        free, _ = torch.cuda.mem_get_info()
        print(f'it {it}: device memory in use {(total - free) / 2**20:.0f} MiB, torch peak {torch.cuda.max_memory_allocated() / 2**20:.0f} MiB', flush=True)
x = torch.rand(32, 2048, 3, device=dev, requires_grad=True); y = torch.rand(32, 2048, 3, device=dev)
for it in range(3000):
    x.grad = None
    (chamfer(x, y) + match_cost(x, y)).sum().backward()
torch.cuda.synchronize()
free, _ = torch.cuda.mem_get_info()
print(f'after 3000 bench-shaped steps: device memory in use {(total - free) / 2**20:.0f} MiB; elapsed {time.time() - t0:.1f} s')
