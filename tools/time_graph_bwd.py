"""get_graph_features / get_neighbours backward at the encoder's shapes: the C call alone on a resident gradient tensor
(autograd's `.sum().backward()` would first materialise its expanded scalar: a 838 MB copy that is not this kernel's)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudcounterfactual_amd import neighbour_ops as ops, _lib
L = _lib.lib
if os.environ.get('PCC_EDGE_SCATTER') == '1':  # tool-side switch -> the library's measurement hook
    os.environ['PCC_TEST_HOOKS'] = '1'; _lib.set_tuning('edge_scatter', 1)
dev = torch.device('cuda:0')
def ev(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
B, N = 32, 2048
st = torch.cuda.current_stream().cuda_stream
import ctypes
for c, k in ((64, 25), (3, 25), (128, 25), (64, 20), (3, 4)):
    x = torch.randn(B, c, N, device=dev)
    idx = ops.hip_knn(x, k)
    g2 = torch.randn(B, 2 * c, N, k, device=dev)
    g1 = torch.randn(B, c, N, k, device=dev)
    gx = torch.empty(B, c, N, device=dev)
    t_f = ev(lambda: L.pcc_graph_features_bwd(B, c, N, k, idx.data_ptr(), g2.data_ptr(), gx.data_ptr(), st))
    t_g = ev(lambda: L.pcc_gather_neighbours_bwd(B, c, N, k, idx.data_ptr(), g1.data_ptr(), gx.data_ptr(), st))
    gb = g2.numel() * 4 / 1e9
    print(f'c={c:3d} k={k:2d}: graph_features_bwd {t_f:7.1f} us ({gb / (t_f * 1e-6) / 1e3:.2f} TB/s of gradient read)  '
          f'gather_bwd {t_g:7.1f} us ({gb / 2 / (t_g * 1e-6) / 1e3:.2f} TB/s)')
    L.pcc_profile_enable(1)
    for _ in range(5):
        L.pcc_graph_features_bwd(B, c, N, k, idx.data_ptr(), g2.data_ptr(), gx.data_ptr(), st)
        L.pcc_gather_neighbours_bwd(B, c, N, k, idx.data_ptr(), g1.data_ptr(), gx.data_ptr(), st)
    torch.cuda.synchronize()
    for name in (b'edge_chunk_sort_kernel', b'edge_self_sum_kernel', b'edge_stream_bwd_kernel<features>', b'edge_stream_bwd_kernel<gather>'):
        us = ctypes.c_double(); cnt = ctypes.c_int()
        L.pcc_profile_read(name, ctypes.byref(us), ctypes.byref(cnt))
        if cnt.value: print(f'      {name.decode():36s} {us.value:7.1f} us x{cnt.value}')
    L.pcc_profile_enable(0); L.pcc_profile_reset()
