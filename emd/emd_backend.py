"""The two functions of the reference's ``emd_backend`` pybind module (``external/emd/src/emd.cpp:14-30``) with their
argument lists, served by the persistent HIP auction kernel (``include/pcc_emd.h``).  The ten work tensors the
reference's seven kernels communicate through are accepted and left untouched: the auction state lives in LDS."""

from __future__ import annotations

import torch

from pointcloudcounterfactual_amd import _lib

_L = _lib.lib


def _chk(t: torch.Tensor, name: str, dtype: torch.dtype) -> None:
    if t.device.type != 'cuda':
        raise RuntimeError(f'{name} must be a CUDA tensor')
    if not t.is_contiguous():
        raise RuntimeError(f'{name} must be contiguous')
    if t.dtype != dtype:
        raise RuntimeError(f'{name} must be {dtype}')


def forward(xyz1, xyz2, dist, assignment, price=None, assignment_inv=None, bid=None, bid_increments=None,
            max_increments=None, unass_idx=None, unass_cnt=None, unass_cnt_sum=None, cnt_tmp=None, max_idx=None,
            eps: float = 0.005, iters: int = 50) -> int:
    """emd_cuda_forward (emd_cuda.cu:227-281): fills ``dist[B,n]`` and ``assignment[B,n]``; returns 1, or -1 with the
    reference's input errors (:235-248)."""
    _chk(xyz1, 'xyz1', torch.float32)
    _chk(xyz2, 'xyz2', torch.float32)
    _chk(dist, 'dist', torch.float32)
    _chk(assignment, 'assignment', torch.int32)
    b, n, _ = xyz1.shape
    if xyz2.shape[1] != n:
        print('Input Error! The two point clouds should have the same size.')
        return -1
    if b > 512:
        print('Input Error! The batch size should be less than 512.')
        return -1
    if n % 1024 != 0:
        print('Input Error! The size of the point clouds should be a multiple of 1024.')
        return -1
    with torch.cuda.device(xyz1.device):
        _lib.check(_L.pcc_auction_forward(b, n, xyz1.data_ptr(), xyz2.data_ptr(), float(eps), int(iters),
                                          dist.data_ptr(), assignment.data_ptr(),
                                          torch.cuda.current_stream(xyz1.device).cuda_stream), 'emd forward')
    return 1


def backward(xyz1, xyz2, gradxyz, graddist, idx) -> int:
    """emd_cuda_backward (emd_cuda.cu:301-315): ``gradxyz = 2 graddist (xyz1 - xyz2[idx])``."""
    _chk(xyz1, 'xyz1', torch.float32)
    _chk(xyz2, 'xyz2', torch.float32)
    _chk(gradxyz, 'gradxyz', torch.float32)
    _chk(graddist, 'graddist', torch.float32)
    _chk(idx, 'idx', torch.int32)
    b, n, _ = xyz1.shape
    with torch.cuda.device(xyz1.device):
        _lib.check(_L.pcc_auction_backward(b, n, xyz1.data_ptr(), xyz2.data_ptr(), graddist.data_ptr(), idx.data_ptr(),
                                           gradxyz.data_ptr(), torch.cuda.current_stream(xyz1.device).cuda_stream),
                   'emd backward')
    return 1
