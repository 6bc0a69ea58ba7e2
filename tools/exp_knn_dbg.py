"""Timing of the role-split k-NN kernel with parts switched off (results are wrong then): where a stage's time goes."""
import os, sys
os.environ.setdefault('PCC_TEST_HOOKS', '1')
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudcounterfactual_amd import neighbour_ops as ops, _lib
def ev(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
_lib.set_tuning('knn_nosplit', 2)
names = {0: 'all', 1: 'no rounds', 3: 'no rounds, no screen', 4: 'no MFMA', 7: 'no rounds/screen/MFMA', 23: '.. and no dist store', 8: 'no final phase',
         31: 'staging and barriers only', 5: 'no rounds, no MFMA', 9: 'no rounds, no final'}
for c in (16, 64, 128):
    x = torch.randn(32, c, 2048, device='cuda:0')
    for d, name in names.items():
        _lib.set_tuning('knn_dbg', d)
        print(f'c={c:3d} dbg={d:2d} {name:28s} {ev(lambda: ops.hip_knn(x, 25)):7.1f} us')
_lib.set_tuning('knn_dbg', 0)
