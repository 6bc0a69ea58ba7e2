"""A/B: pcc_match_cost (forward + gradients) as direct launches and as a replayed hipGraph, for the lane count /
culling switches given in the environment (they are read once per process).  usage: exp_lanes.py [recon|uniform]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import pair
from pointcloudcounterfactual_amd import _lib
L = _lib.lib
kind = sys.argv[1] if len(sys.argv) > 1 else 'recon'
dev = torch.device('cuda:0')
B, N = 32, 2048
a, c = pair(1236, B, N, N, kind)
t1, t2 = torch.from_numpy(a).to(dev), torch.from_numpy(c).to(dev)
cost = torch.empty(B, device=dev); g1 = torch.empty(B, N, 3, device=dev); g2 = torch.empty(B, N, 3, device=dev)
tag = ' '.join(f'{k}={v}' for k, v in sorted(os.environ.items()) if k.startswith('PCC_')) or 'default'

def call(st):
    rc = L.pcc_match_cost(B, N, N, t1.data_ptr(), t2.data_ptr(), None, cost.data_ptr(), g1.data_ptr(), g2.data_ptr(), st)
    assert rc == 0, rc

def ev(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

st = torch.cuda.current_stream().cuda_stream
d = ev(lambda: call(st))
ref = (cost.clone(), g1.clone(), g2.clone())
print(f'[{kind}] {tag}: direct {d:.1f} us', flush=True)
if os.environ.get('PCC_EXP_GRAPH', '1') == '1':
    try:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            call(s.cuda_stream)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s, capture_error_mode='relaxed'):
            call(s.cuda_stream)
        cost.zero_(); g1.zero_(); g2.zero_()
        r = ev(g.replay)
        same = all(torch.equal(x, y) for x, y in zip(ref, (cost, g1, g2)))
        print(f'[{kind}] {tag}: graph replay {r:.1f} us, outputs identical to direct: {same}', flush=True)
    except Exception as e:
        print(f'[{kind}] {tag}: graph failed: {e!r}', flush=True)
