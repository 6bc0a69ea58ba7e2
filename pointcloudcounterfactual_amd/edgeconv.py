"""Fused EdgeConv front-end.

The reference's EdgeConv block (``src/module/encoders.py:50-53`` with ``EdgeConvLayer``, ``src/module/layers.py:159-203``)

    feat = cat([x[idx] - x, x])            # [B, 2C, N, k]   up to 1.68 GB per layer at B=32, N=2048, k=25, C=128
    y = max_j act(BatchNorm2d(Conv2d_1x1(feat)))                        # act = LeakyReLU(0.2) or identity

never needs the ``[B,2C,N,k]`` tensor.  The 1x1 convolution is linear, so with ``W = [Wa | Wb]``

    z[b,:,n,j] = Wa (x_j - x_n) + Wb x_n = y1[b,:,idx[n,j]] + y2[b,:,n],     y1 = Wa x,  y2 = (Wb - Wa) x      ([B,C',N])

and, because BatchNorm is a per-channel affine map and the activation is increasing,

    max_j act(bn(z_j)) = act(bn(max_j z_j))  if the BatchNorm scale of the channel is >= 0, act(bn(min_j z_j)) otherwise,
    max_j z_j = max_j y1[idx[n,j]] + y2[n]   (y2 does not depend on j).

Training-mode BatchNorm statistics over (b,n,j) are exact sums of ``[B,C',N]`` quantities:

    sum z   = sum_t deg[t] y1[t] + k sum_n y2[n]
    sum z^2 = sum_t deg[t] y1[t]^2 + k sum_n y2[n]^2 + 2 sum_n y2[n] S1[n],      S1[n] = sum_j y1[idx[n,j]]

(``deg[t]`` = in-degree of t in the kNN graph).  Everything is differentiable PyTorch on ``[B,C',N]`` tensors plus three
small HIP kernels (neighbour sum and its scatter backward, min/max neighbour selection), so autograd gives the exact
gradients of the unfused block, including the statistics' dependence on every edge.  The parameters are the
reference's (``Conv2d(2C, C', 1, bias=False)`` + ``BatchNorm2d(C')``), so state dicts are interchangeable.
"""

from __future__ import annotations

from typing import Any

import torch
from torch import nn
from torch.autograd import Function

from pointcloudcounterfactual_amd import _lib
from pointcloudcounterfactual_amd import neighbour_ops as ops

_L = _lib.lib


class _NeighbourSum(Function):
    """``S[b,c,n] = sum_j y[b,c,idx[b,n,j]]``; backward scatters ``g[b,c,n]`` along every edge."""

    @staticmethod
    def forward(ctx: Any, y: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        y = y.contiguous()
        b, c, n = y.shape
        k = idx.shape[2]
        out = torch.empty_like(y)
        with torch.cuda.device(y.device):
            _lib.check(_L.pcc_neighbour_sum(b, c, n, k, y.data_ptr(), idx.data_ptr(), out.data_ptr(),
                                            torch.cuda.current_stream(y.device).cuda_stream), 'neighbour_sum')
        ctx.save_for_backward(idx)
        return out

    @staticmethod
    def backward(ctx: Any, grad: torch.Tensor) -> tuple[torch.Tensor, None]:
        (idx,) = ctx.saved_tensors
        grad = grad.contiguous()
        b, c, n = grad.shape
        gy = torch.empty_like(grad)
        with torch.cuda.device(grad.device):
            _lib.check(_L.pcc_neighbour_sum_bwd(b, c, n, idx.shape[2], idx.data_ptr(), grad.data_ptr(), gy.data_ptr(),
                                                torch.cuda.current_stream(grad.device).cuda_stream), 'neighbour_sum_bwd')
        return gy, None


def neighbour_sum(y: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    if y.device.type == 'cuda':
        return _NeighbourSum.apply(y, idx.contiguous())
    return ops.get_neighbours(y, idx, idx.shape[2])[1].sum(-1)


def neighbour_minmax_target(y: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """``tsel[B,2,C,N]`` int64: per (channel, point) the neighbour with the largest / smallest ``y`` (no gradient)."""
    y = y.detach().contiguous()
    b, c, n = y.shape
    if y.device.type == 'cuda':
        tsel = torch.empty((b, 2, c, n), dtype=torch.int64, device=y.device)
        with torch.cuda.device(y.device):
            _lib.check(_L.pcc_neighbour_minmax_target(b, c, n, idx.shape[2], y.data_ptr(), idx.contiguous().data_ptr(),
                                                      tsel.data_ptr(), torch.cuda.current_stream(y.device).cuda_stream),
                       'neighbour_minmax_target')
        return tsel
    nb = ops.get_neighbours(y, idx, idx.shape[2])[1]                      # [B,C,N,k]
    ie = idx[:, None, :, :].expand(-1, c, -1, -1)
    return torch.stack([torch.gather(ie, 3, nb.argmax(3, keepdim=True))[..., 0],
                        torch.gather(ie, 3, nb.argmin(3, keepdim=True))[..., 0]], dim=1)


class FusedEdgeConv(nn.Module):
    """Drop-in for ``get_graph_features -> EdgeConvLayer -> max(dim=3)`` of the reference's DGCNN blocks."""

    def __init__(self, in_dim: int, out_dim: int, act: bool = True, momentum: float = 0.1, eps: float = 1e-5) -> None:
        super().__init__()
        self.conv = nn.Conv2d(2 * in_dim, out_dim, 1, bias=False)         # same parameters as the unfused block
        self.bn = nn.BatchNorm2d(out_dim, momentum=momentum, eps=eps)
        self.act = nn.LeakyReLU(0.2) if act else nn.Identity()
        self.in_dim = in_dim

    def forward(self, x: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        """``x[B,C,N]``, ``idx[B,N,k]`` -> ``[B,C',N]``."""
        b, c, n = x.shape
        k = idx.shape[2]
        w = self.conv.weight[:, :, 0, 0]
        wa, wb = w[:, :c], w[:, c:]
        y1 = torch.matmul(wa, x)                                          # [B,C',N]
        y2 = torch.matmul(wb - wa, x)
        if self.training or not self.bn.track_running_stats:
            m = float(b * n * k)
            deg = torch.zeros((b, n), dtype=torch.float32, device=x.device)
            deg.scatter_add_(1, idx.reshape(b, -1), torch.ones((b, n * k), dtype=torch.float32, device=x.device))
            s1 = neighbour_sum(y1, idx)
            y1d, y2d, s1d, degd = y1.double(), y2.double(), s1.double(), deg.double()[:, None, :]
            sum_z = (degd * y1d).sum((0, 2)) + k * y2d.sum((0, 2))
            sum_z2 = (degd * y1d * y1d).sum((0, 2)) + k * (y2d * y2d).sum((0, 2)) + 2.0 * (y2d * s1d).sum((0, 2))
            mean = sum_z / m
            var = (sum_z2 / m - mean * mean).clamp_min(0.0)                # biased, as BatchNorm normalises with
            if self.training and self.bn.track_running_stats:
                with torch.no_grad():
                    mom = self.bn.momentum
                    self.bn.running_mean.mul_(1 - mom).add_(mom * mean.float())
                    self.bn.running_var.mul_(1 - mom).add_(mom * (var * m / max(m - 1.0, 1.0)).float())
                    self.bn.num_batches_tracked += 1
            mean, var = mean.float(), var.float()
        else:
            mean, var = self.bn.running_mean, self.bn.running_var
        scale = self.bn.weight * torch.rsqrt(var + self.bn.eps)
        shift = self.bn.bias - mean * scale
        tsel = neighbour_minmax_target(y1, idx)                           # [B,2,C',N]
        pick = torch.where((scale >= 0)[None, :, None], tsel[:, 0], tsel[:, 1])
        z = torch.gather(y1, 2, pick) + y2                                # the edge that survives max over k
        return self.act(z * scale[None, :, None] + shift[None, :, None])


def reference_edgeconv(x: torch.Tensor, idx: torch.Tensor, conv: nn.Conv2d, bn: nn.BatchNorm2d, act: nn.Module) -> torch.Tensor:
    """The unfused block exactly as the reference composes it (used by the parity tests)."""
    _i, feat = ops.get_graph_features(x, idx, idx.shape[2])
    return act(bn(conv(feat))).max(dim=3)[0]
