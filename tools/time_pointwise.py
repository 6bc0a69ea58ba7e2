"""Micro-benchmark: ways to run a 1x1 Conv1d over [B,C,N] (forward + backward) on PyTorch-ROCm."""
import torch, torch.nn.functional as F
dev = torch.device('cuda:0')
def ev(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
B, N = 32, 2048
for cin, cout in ((1024, 1024), (1024, 256), (256, 16), (512, 1024), (64, 1024)):
    x = torch.randn(B, cin, N, device=dev, requires_grad=True)
    w = torch.randn(cout, cin, 1, device=dev, requires_grad=True)
    g = torch.randn(B, cout, N, device=dev)
    def run(f):
        def step():
            x.grad = None; w.grad = None
            f().backward(g)
        return step
    conv = lambda: F.conv1d(x, w)
    mm = lambda: torch.matmul(w.squeeze(-1), x)
    def folded():
        x2 = x.permute(1, 0, 2).reshape(cin, B * N)
        return torch.mm(w.squeeze(-1), x2).view(cout, B, N).permute(1, 0, 2)
    def lin():  # channels-last linear
        return F.linear(x.transpose(1, 2), w.squeeze(-1)).transpose(1, 2)
    fl = 2 * B * N * cin * cout * 3 / 1e9
    res = {name: ev(run(f)) for name, f in (('conv1d', conv), ('matmul', mm), ('folded_mm', folded), ('linear_t', lin))}
    print(f'{cin}->{cout}: ' + ', '.join(f'{k} {v:.2f} ms ({fl / v:.0f} TF/s)' for k, v in res.items()))
