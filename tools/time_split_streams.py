"""Experiment: one pcc_match_cost call over B=32 vs the batch split over K HIP streams (bubbles of the 19 dependent
phase launches of one stream filled by the other streams' kernels)."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import pair
from pointcloudcounterfactual_amd import _lib
L = _lib.lib
dev = torch.device('cuda:0')
B, N = 32, 2048
a, c = pair(1236, B, N, N)
t1, t2 = torch.from_numpy(a).to(dev), torch.from_numpy(c).to(dev)
cost = torch.empty(B, device=dev); g1 = torch.empty(B, N, 3, device=dev); g2 = torch.empty(B, N, 3, device=dev)

def run(K):
    streams = [torch.cuda.Stream() for _ in range(K)]
    per = B // K
    def once():
        main = torch.cuda.current_stream()
        ev = torch.cuda.Event(); ev.record(main)
        for k, s in enumerate(streams):
            s.wait_event(ev)
            o = k * per
            L.pcc_match_cost(per, N, N, t1[o:].data_ptr(), t2[o:].data_ptr(), None, cost[o:].data_ptr(), g1[o:].data_ptr(), g2[o:].data_ptr(), s.cuda_stream)
            e2 = torch.cuda.Event(); e2.record(s); main.wait_event(e2)
    for _ in range(3): once()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): once()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 20 * 1e6, float(cost.sum())

for K in (1, 2, 4, 1, 2, 4):
    us, cs = run(K)
    print(f'K={K} streams: {us:.1f} us per B=32 batch, cost sum {cs:.4f}')
