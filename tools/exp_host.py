"""Is pcc_match_cost host-bound?  Host enqueue time of ONE call on an idle queue (no back-pressure possible), GPU time
of the same call, and the steady-state numbers."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import pair
from pointcloudcounterfactual_amd import _lib
L = _lib.lib
dev = torch.device('cuda:0')
B, N = 32, 2048
a, c = pair(1236, B, N, N, 'recon')
t1, t2 = torch.from_numpy(a).to(dev), torch.from_numpy(c).to(dev)
cost = torch.empty(B, device=dev); g1 = torch.empty(B, N, 3, device=dev); g2 = torch.empty(B, N, 3, device=dev)
st = torch.cuda.current_stream().cuda_stream
if os.environ.get('PCC_AM_NOSPLIT') == '1':  # tool-side switch -> the library's measurement hook (it reads no behaviour variables)
    os.environ['PCC_TEST_HOOKS'] = '1'
    from pointcloudcounterfactual_amd import _lib as _l
    _l.set_tuning('am_nosplit', 1)
tag = ' '.join(f'{k}={v}' for k, v in sorted(os.environ.items()) if k.startswith('PCC_')) or 'default'
def call():
    L.pcc_match_cost(B, N, N, t1.data_ptr(), t2.data_ptr(), None, cost.data_ptr(), g1.data_ptr(), g2.data_ptr(), st)
for _ in range(5): call()
torch.cuda.synchronize()
hs, gs = [], []
for _ in range(10):
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); s.record(); call(); e.record(); t1_ = time.perf_counter()
    torch.cuda.synchronize()
    hs.append((t1_ - t0) * 1e6); gs.append(s.elapsed_time(e) * 1e3)
print(f'[{tag}] single call on an idle queue: host {sorted(hs)[len(hs)//2]:.1f} us, gpu {sorted(gs)[len(gs)//2]:.1f} us', flush=True)
for iters in (3, 10, 40):
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); s.record()
    for _ in range(iters): call()
    e.record(); t1_ = time.perf_counter()
    torch.cuda.synchronize()
    print(f'[{tag}] {iters} calls back to back: host {(t1_ - t0) / iters * 1e6:.1f} us/call, gpu {s.elapsed_time(e) / iters * 1e3:.1f} us/call', flush=True)
# a trivial kernel launched the same number of times: the per-launch floor of this runtime
x = torch.zeros(64, device=dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): L.pcc_chamfer_loss(0, 0, None, 0, None, 0, None, None, None, None, None, st)  # returns before launching
t1_ = time.perf_counter()
print(f'ctypes call floor: {(t1_ - t0) / 200 * 1e6:.2f} us', flush=True)
