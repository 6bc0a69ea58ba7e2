"""Where does an initialised RCCL process group slow the bench step?  Wall time per iteration (30 iterations, one sync at
the end) of: the C call, the Python backend call, the autograd step -- before and after init_process_group."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29534')
os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1'); os.environ.setdefault('LOCAL_RANK', '0')
from tests.util import pair
from pointcloudcounterfactual_amd import _lib, backend, chamfer_emd
import torch.distributed as dist
L = _lib.lib
dev = torch.device('cuda:0'); torch.cuda.set_device(0)
B, N = 32, 2048
a, c = pair(1236, B, N, N, 'recon')
t1, t2 = torch.from_numpy(a).to(dev), torch.from_numpy(c).to(dev)
t1g = t1.clone().requires_grad_(True)
cost = torch.empty(B, device=dev); g1 = torch.empty(B, N, 3, device=dev); g2 = torch.empty(B, N, 3, device=dev)
st = torch.cuda.current_stream().cuda_stream
def c_call():
    L.pcc_match_cost(B, N, N, t1.data_ptr(), t2.data_ptr(), None, cost.data_ptr(), g1.data_ptr(), g2.data_ptr(), st)
def py_call():
    backend.ChamferEMD(t1, t2, True, True)
def fwd_only():
    with torch.no_grad():
        lc, le = chamfer_emd(t1, t2)
        (lc + le).sum()
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
def report(tag):
    print(f'[{tag}] ' + '; '.join(f'{n}: wall {w:.0f} us (host enqueue {h:.0f})' for n, (w, h) in
          (('C pcc_match_cost', wall(c_call)), ('backend.ChamferEMD', wall(py_call)), ('no_grad fwd', wall(fwd_only)), ('autograd step', wall(step)))), flush=True)
report('before init_process_group')
dist.init_process_group('nccl', device_id=dev)
report('after init_process_group')
dist.barrier(); torch.cuda.synchronize()
report('after barrier')
import threading
print('threads:', threading.active_count(), 'cpu affinity:', len(os.sched_getaffinity(0)), flush=True)
dist.destroy_process_group()
report('after destroy')
