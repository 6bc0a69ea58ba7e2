"""RCCL initialised FIRST (as bench.py does under the launcher), then the autograd step."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29535')
os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1'); os.environ.setdefault('LOCAL_RANK', '0')
import torch.distributed as dist
dev = torch.device('cuda:0'); torch.cuda.set_device(0)
mode = sys.argv[1] if len(sys.argv) > 1 else 'first'
if mode == 'first':
    dist.init_process_group('nccl', device_id=dev)
from tests.util import pair
from pointcloudcounterfactual_amd import chamfer_emd
B, N = 32, 2048
a, c = pair(1236, B, N, N, 'recon')
t1, t2 = torch.from_numpy(a).to(dev), torch.from_numpy(c).to(dev)
t1g = t1.clone().requires_grad_(True)
def step():
    t1g.grad = None
    lc, le = chamfer_emd(t1g, t2)
    (lc + le).sum().backward()
def wall(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    h = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6, h / iters * 1e6
tag = f"{mode}, torchrun={'TORCHELASTIC_RUN_ID' in os.environ}"
print(f'[{tag}] autograd step: wall %.0f us (host enqueue %.0f)' % wall(step), flush=True)
if mode == 'first':
    dist.barrier(); torch.cuda.synchronize()
    print(f'[{tag}] after barrier: wall %.0f us (host enqueue %.0f)' % wall(step), flush=True)
    t = torch.tensor([1.0], device=dev, dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX); float(t.item())
    print(f'[{tag}] after all_reduce: wall %.0f us (host enqueue %.0f)' % wall(step), flush=True)
    dist.destroy_process_group()
print({k: v for k, v in os.environ.items() if k.startswith(('TORCH', 'NCCL', 'OMP', 'HSA', 'HIP'))}, flush=True)
