"""Does an initialised RCCL process group change the cost of this library's launches?  Host enqueue time and GPU time of
pcc_match_cost before init_process_group('nccl'), after it, and after the first collective."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1'); os.environ.setdefault('LOCAL_RANK', '0')
from tests.util import pair
from pointcloudcounterfactual_amd import _lib
import torch.distributed as dist
L = _lib.lib
dev = torch.device('cuda:0'); torch.cuda.set_device(0)
B, N = 32, 2048
a, c = pair(1236, B, N, N, 'recon')
t1, t2 = torch.from_numpy(a).to(dev), torch.from_numpy(c).to(dev)
cost = torch.empty(B, device=dev); g1 = torch.empty(B, N, 3, device=dev); g2 = torch.empty(B, N, 3, device=dev)
st = torch.cuda.current_stream().cuda_stream
def call():
    L.pcc_match_cost(B, N, N, t1.data_ptr(), t2.data_ptr(), None, cost.data_ptr(), g1.data_ptr(), g2.data_ptr(), st)
def report(tag):
    for _ in range(5): call()
    torch.cuda.synchronize()
    hs, gs = [], []
    for _ in range(10):
        torch.cuda.synchronize()
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); s.record(); call(); e.record(); t1_ = time.perf_counter()
        torch.cuda.synchronize()
        hs.append((t1_ - t0) * 1e6); gs.append(s.elapsed_time(e) * 1e3)
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); s.record()
    for _ in range(40): call()
    e.record(); t1_ = time.perf_counter(); torch.cuda.synchronize()
    print(f'[{tag}] single call: host {sorted(hs)[5]:.1f} us, gpu {sorted(gs)[5]:.1f} us; 40 calls: host {(t1_-t0)/40*1e6:.1f} us/call, gpu {s.elapsed_time(e)/40*1e3:.1f} us/call', flush=True)
report('before init_process_group')
mode = sys.argv[1] if len(sys.argv) > 1 else 'device_id'
if mode == 'device_id':
    dist.init_process_group('nccl', device_id=dev)
else:
    dist.init_process_group('nccl')
report(f'after init_process_group ({mode})')
dist.barrier(); torch.cuda.synchronize()
report('after barrier')
t = torch.ones(1, device=dev); dist.all_reduce(t); torch.cuda.synchronize()
report('after all_reduce')
dist.destroy_process_group()
report('after destroy_process_group')
