"""kNN-graph operations with the reference's Python API (``src/utils/neighbour_ops.py:63-133``).

On the accelerator every function runs hand-written HIP kernels through the C ABI
(``include/pcc_neighbour.h``); the reference uses PyKeOps there, which has no ROCm backend.  For CPU tensors the
functions follow the reference's own device dispatch (``neighbour_ops.py:29,65``) and evaluate its torch
formulas -- that is the reference's CPU path, not a fallback of the GPU path: an accelerator tensor never
leaves the device, and a missing HIP library is an ``ImportError``.

Layouts are the reference's: ``x[B,C,N]`` float32, ``indices[B,N,k]`` int64 (an empty ``indices`` tensor means
"compute the kNN now", ``:88-91``).
"""

from __future__ import annotations

from typing import Any

import torch
from torch.autograd import Function

from pointcloudcounterfactual_amd import _lib

_L = _lib.lib


def _stream(x: torch.Tensor) -> int:
    return torch.cuda.current_stream(x.device).cuda_stream


def _prep(x: torch.Tensor, indices: torch.Tensor | None = None) -> None:
    if x.dtype != torch.float32 or not x.is_contiguous():
        raise RuntimeError('x must be a contiguous float32 tensor [B,C,N]')
    if indices is not None and (indices.dtype != torch.int64 or not indices.is_contiguous()
                                or indices.device != x.device):
        raise RuntimeError('indices must be a contiguous int64 tensor [B,N,k] on the device of x')


# ---- squared distances (reference neighbour_ops.py:16-50) -----------------------------------------------------------------


def index_k_neighbours(pcs: list[Any], k: int) -> Any:
    """Host-side kNN precompute of the dataset readers (reference ``neighbour_ops.py:16-24``; ``modelnet.py:18``):
    a KD-tree per cloud, ``[len(pcs), N, k]``.  Dataset preparation, not on the device path."""
    import numpy as np
    from sklearn.neighbors import KDTree

    indices_list = []
    for pc in pcs:
        indices = KDTree(pc).query(pc, k, return_distance=False)
        indices_list.append(indices.reshape(-1, k))
    return np.stack(indices_list)


def pykeops_square_distance(t1: torch.Tensor, t2: torch.Tensor) -> Any:
    """Lazy ``D[b,i,j] = |t1[b,i] - t2[b,j]|^2`` (reference ``neighbour_ops.py:35-40``) over this package's
    ``LazyTensor`` (``keops_shim``): its reductions -- ``argmin`` / ``min`` over either axis, ``argKmin``, ``sum`` --
    run the HIP kernels; the matrix never exists.  Imported by ``metrics_and_losses.py:18`` and ``quantize.py:6``."""
    from pointcloudcounterfactual_amd.keops_shim import LazyTensor

    t1_lazy = LazyTensor(t1[:, :, None, :])
    t2_lazy = LazyTensor(t2[:, None, :, :])
    return ((t1_lazy - t2_lazy) ** 2).sum(-1)


def torch_square_distance(t1: torch.Tensor, t2: torch.Tensor) -> torch.Tensor:
    """Dense expanded-form ``[B,N,M]`` squared distances (reference ``neighbour_ops.py:43-50``): the reference's CPU
    path (``torch_chamfer``, ``metrics_and_losses.py:44-47``)."""
    from pointcloudcounterfactual_amd.losses import torch_square_distance as impl

    return impl(t1, t2)


def square_distance(t1: torch.Tensor, t2: torch.Tensor) -> Any:
    """Device dispatch of the reference (``neighbour_ops.py:27-32``): lazy on the accelerator, dense on the CPU."""
    if t1.device.type == 'cuda':
        return pykeops_square_distance(t1, t2)
    return torch_square_distance(t1, t2)


# ---- kNN ------------------------------------------------------------------------------------------------------


def self_square_distance(t1: torch.Tensor) -> torch.Tensor:
    """Expanded-form self distances (reference ``neighbour_ops.py:53-60``); CPU path of ``knn``."""
    t2 = t1.transpose(-1, -2)
    square_component = torch.sum(t1**2, -2, keepdim=True)
    dist = torch.tensor(-2) * torch.matmul(t2, t1)
    dist += square_component
    dist += square_component.transpose(-1, -2)
    return dist


def torch_knn(x: torch.Tensor, k: int) -> torch.Tensor:
    """Reference CPU kNN (``neighbour_ops.py:71-74``)."""
    return self_square_distance(x).topk(k=k, largest=False)[1]


def hip_knn(x: torch.Tensor, k: int) -> torch.Tensor:
    """``x[B,C,N] -> indices[B,N,k]`` int64, ascending distance, ties by ascending index (replaces ``pykeops_knn``)."""
    x = x.contiguous()
    _prep(x)
    b, c, n = x.shape
    out = torch.empty((b, n, k), dtype=torch.int64, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_L.pcc_knn(b, c, n, k, x.data_ptr(), out.data_ptr(), _stream(x)), 'knn')
    return out


def pykeops_knn(x: torch.Tensor, k: int) -> torch.Tensor:
    """The reference's accelerator kNN (``neighbour_ops.py:77-82``: ``argKmin`` of the lazy self distance), here the
    HIP search."""
    return hip_knn(x.detach(), k)


def knn(x: torch.Tensor, k: int) -> torch.Tensor:
    """Device dispatch of the reference (``neighbour_ops.py:63-68``)."""
    if x.device.type == 'cuda':
        return hip_knn(x.detach(), k)
    return torch_knn(x, k)


# ---- gather / edge features / max over k -------------------------------------------------------------------------


class _Gather(Function):
    @staticmethod
    def forward(ctx: Any, x: torch.Tensor, indices: torch.Tensor) -> torch.Tensor:
        x = x.contiguous()
        indices = indices.contiguous()
        _prep(x, indices)
        b, c, n = x.shape
        k = indices.shape[2]
        out = torch.empty((b, c, n, k), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_L.pcc_gather_neighbours(b, c, n, k, x.data_ptr(), indices.data_ptr(), out.data_ptr(),
                                                _stream(x)), 'gather_neighbours')
        ctx.save_for_backward(indices)
        ctx.shape = (b, c, n, k)
        return out

    @staticmethod
    def backward(ctx: Any, grad: torch.Tensor) -> tuple[torch.Tensor, None]:
        (indices,) = ctx.saved_tensors
        b, c, n, k = ctx.shape
        grad = grad.contiguous()
        gx = torch.empty((b, c, n), dtype=torch.float32, device=grad.device)
        with torch.cuda.device(grad.device):
            _lib.check(_L.pcc_gather_neighbours_bwd(b, c, n, k, indices.data_ptr(), grad.data_ptr(), gx.data_ptr(),
                                                    _stream(grad)), 'gather_neighbours_bwd')
        return gx, None


class _GraphFeatures(Function):
    @staticmethod
    def forward(ctx: Any, x: torch.Tensor, indices: torch.Tensor) -> torch.Tensor:
        x = x.contiguous()
        indices = indices.contiguous()
        _prep(x, indices)
        b, c, n = x.shape
        k = indices.shape[2]
        out = torch.empty((b, 2 * c, n, k), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_L.pcc_graph_features(b, c, n, k, x.data_ptr(), indices.data_ptr(), out.data_ptr(),
                                             _stream(x)), 'graph_features')
        ctx.save_for_backward(indices)
        ctx.shape = (b, c, n, k)
        return out

    @staticmethod
    def backward(ctx: Any, grad: torch.Tensor) -> tuple[torch.Tensor, None]:
        (indices,) = ctx.saved_tensors
        b, c, n, k = ctx.shape
        grad = grad.contiguous()
        gx = torch.empty((b, c, n), dtype=torch.float32, device=grad.device)
        with torch.cuda.device(grad.device):
            _lib.check(_L.pcc_graph_features_bwd(b, c, n, k, indices.data_ptr(), grad.data_ptr(), gx.data_ptr(),
                                                 _stream(grad)), 'graph_features_bwd')
        return gx, None


class _GraphMaxPool(Function):
    @staticmethod
    def forward(ctx: Any, x: torch.Tensor, indices: torch.Tensor) -> torch.Tensor:
        x = x.contiguous()
        indices = indices.contiguous()
        _prep(x, indices)
        b, c, n = x.shape
        k = indices.shape[2]
        out = torch.empty((b, c, n), dtype=torch.float32, device=x.device)
        arg = torch.empty((b, c, n), dtype=torch.int32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_L.pcc_graph_max_pool(b, c, n, k, x.data_ptr(), indices.data_ptr(), out.data_ptr(),
                                             arg.data_ptr(), _stream(x)), 'graph_max_pool')
        ctx.save_for_backward(indices, arg)
        ctx.shape = (b, c, n, k)
        return out

    @staticmethod
    def backward(ctx: Any, grad: torch.Tensor) -> tuple[torch.Tensor, None]:
        indices, arg = ctx.saved_tensors
        b, c, n, k = ctx.shape
        grad = grad.contiguous()
        gx = torch.empty((b, c, n), dtype=torch.float32, device=grad.device)
        with torch.cuda.device(grad.device):
            _lib.check(_L.pcc_graph_max_pool_bwd(b, c, n, k, indices.data_ptr(), arg.data_ptr(), grad.data_ptr(),
                                                 gx.data_ptr(), _stream(grad)), 'graph_max_pool_bwd')
        return gx, None


class _GlobalMaxPool(Function):
    @staticmethod
    def forward(ctx: Any, x: torch.Tensor) -> torch.Tensor:
        x = x.contiguous()
        _prep(x)
        b, c, n = x.shape
        out = torch.empty((b, c), dtype=torch.float32, device=x.device)
        arg = torch.empty((b, c), dtype=torch.int32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_L.pcc_global_pool(b, c, n, x.data_ptr(), out.data_ptr(), arg.data_ptr(), None, _stream(x)),
                       'global_pool')
        ctx.save_for_backward(arg)
        ctx.n = n
        return out

    @staticmethod
    def backward(ctx: Any, grad: torch.Tensor) -> torch.Tensor:
        (arg,) = ctx.saved_tensors
        gx = torch.zeros(grad.shape + (ctx.n,), dtype=grad.dtype, device=grad.device)
        gx.scatter_(2, arg.long().unsqueeze(2), grad.unsqueeze(2))
        return gx


def _on_gpu(x: torch.Tensor) -> bool:
    return x.device.type == 'cuda'


def get_neighbours(x: torch.Tensor, indices: torch.Tensor, k: int) -> tuple[torch.Tensor, torch.Tensor]:
    """``(indices, neighbours[B,C,N,k])`` (reference ``neighbour_ops.py:85-94``)."""
    batch, n_feat, n_points = x.size()
    if not indices.numel():
        indices = knn(x, k)
    if _on_gpu(x):
        return indices, _Gather.apply(x, indices.to(x.device))
    indices_expanded = indices.contiguous().view(batch, 1, k * n_points).expand(-1, n_feat, -1)
    neighbours = torch.gather(x, 2, indices_expanded).view(batch, n_feat, n_points, k)
    return indices, neighbours


def get_local_covariance(x: torch.Tensor, indices: torch.Tensor, k: int = 16) -> torch.Tensor:
    """Reference ``neighbour_ops.py:97-103`` (dense part stays in PyTorch-ROCm)."""
    neighbours = get_neighbours(x, indices, k)[1]
    neighbours = neighbours - neighbours.mean(3, keepdim=True)
    covariances = torch.matmul(neighbours.transpose(1, 2), neighbours.permute(0, 2, 3, 1))
    return torch.cat([x, covariances.flatten(start_dim=2).transpose(1, 2)], dim=1).contiguous()


def graph_max_pooling(x: torch.Tensor, indices: torch.Tensor, k: int = 16) -> torch.Tensor:
    """``max_j x[:, :, indices[:, n, j]]`` (reference ``neighbour_ops.py:106-110``)."""
    if _on_gpu(x):
        if not indices.numel():
            indices = knn(x, k)
        return _GraphMaxPool.apply(x, indices.to(x.device))
    neighbours = get_neighbours(x, indices, k)[1]
    return torch.max(neighbours, dim=-1)[0]


def get_graph_features(x: torch.Tensor, indices: torch.Tensor, k: int = 20) -> tuple[torch.Tensor, torch.Tensor]:
    """``(indices, cat([neighbours - x, x])[B,2C,N,k])`` (reference ``neighbour_ops.py:113-119``)."""
    if _on_gpu(x):
        if not indices.numel():
            indices = knn(x, k)
        return indices, _GraphFeatures.apply(x, indices.to(x.device))
    indices_out, neighbours = get_neighbours(x, indices, k)
    xe = x.unsqueeze(3).expand(-1, -1, -1, k)
    return indices_out, torch.cat([neighbours - xe, xe], dim=1).contiguous()


def graph_filtering(x: torch.Tensor, k: int = 4) -> torch.Tensor:
    """Laplacian-like smoothing of the decoder output (reference ``neighbour_ops.py:122-133``)."""
    neighbours = get_neighbours(x, k=k, indices=torch.empty(0))[1]
    neighbours = neighbours[..., 1:]  # the closest neighbour is the point itself
    diff = x.unsqueeze(-1).expand(-1, -1, -1, k - 1) - neighbours
    dist = torch.sqrt(abs((diff**2).sum(1)))
    sigma = torch.clamp(dist[..., 0:1].mean(1, keepdim=True), min=0.005)
    weights = torch.exp(-(dist / sigma))
    x_weight = weights.sum(2).unsqueeze(1).expand(-1, 3, -1)
    weighted_neighbours = weights.unsqueeze(1).expand(-1, 3, -1, -1) * neighbours
    return (1 + x_weight) * x - weighted_neighbours.sum(-1)


def global_max_pool(x: torch.Tensor) -> torch.Tensor:
    """``x[B,C,N].max(dim=2)[0]`` of the encoders / classifier (``encoders.py:58,90``; ``classifier.py:63``)."""
    if _on_gpu(x):
        return _GlobalMaxPool.apply(x)
    return x.max(dim=2, keepdim=False)[0]


def global_max_mean_pool(x: torch.Tensor) -> torch.Tensor:
    """``cat(max_n, mean_n)`` of the classifier head (``classifier.py:63-65``) in one read (inference only)."""
    if _on_gpu(x) and not x.requires_grad:
        x = x.contiguous()
        _prep(x)
        b, c, n = x.shape
        out = torch.empty((b, 2 * c), dtype=torch.float32, device=x.device)
        mx = out[:, :c]
        mean = torch.empty((b, c), dtype=torch.float32, device=x.device)
        mxc = torch.empty((b, c), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_L.pcc_global_pool(b, c, n, x.data_ptr(), mxc.data_ptr(), None, mean.data_ptr(), _stream(x)),
                       'global_pool')
        mx.copy_(mxc)
        out[:, c:] = mean
        return out
    return torch.cat((global_max_pool(x), x.mean(dim=2)), 1)
