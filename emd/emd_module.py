"""``emdModule`` / ``emdFunction`` with the reference's interface (``external/emd/emd/emd_module.py:16-100``)."""

from __future__ import annotations

from typing import Any

import torch
from torch import nn
from torch.autograd import Function

from emd import emd_backend


class emdFunction(Function):  # noqa: N801  (reference spelling)
    @staticmethod
    def forward(ctx: Any, *args: Any, **kwargs: Any):
        xyz1, xyz2, eps, iters, *_ = args
        batch_size1, n, _ = xyz1.size()
        batch_size2, m, _ = xyz2.size()
        # the reference's checks, emd_module.py:23-30
        if n != m:
            raise ValueError('Input point clouds should have the same number of points')
        if batch_size1 != batch_size2:
            raise ValueError('Batch size must be the same')
        if n % 1024:
            raise ValueError('Only valid for clouds of a size multiple of 1024')
        if batch_size1 > 512:
            raise ValueError('Batch size should not exceed 512')
        xyz1 = xyz1.contiguous().float().cuda()
        xyz2 = xyz2.contiguous().float().cuda()
        dist = torch.zeros(batch_size1, n, device=xyz1.device)
        assignment = torch.full((batch_size1, n), -1, device=xyz1.device, dtype=torch.int32)
        emd_backend.forward(xyz1, xyz2, dist, assignment, eps=eps, iters=iters)
        ctx.save_for_backward(xyz1, xyz2, assignment)
        return dist, assignment

    @staticmethod
    def backward(ctx: Any, *grad_outputs: Any) -> Any:
        grad_dist, *_ = grad_outputs
        xyz1, xyz2, assignment = ctx.saved_tensors
        grad_dist = grad_dist.contiguous()
        grad_xyz1 = torch.zeros(xyz1.size(), device=xyz1.device)
        grad_xyz2 = torch.zeros(xyz2.size(), device=xyz2.device)  # only xyz1 receives a gradient (:76-79)
        emd_backend.backward(xyz1, xyz2, grad_xyz1, grad_dist, assignment)
        return grad_xyz1, grad_xyz2, None, None


class emdModule(nn.Module):  # noqa: N801
    def forward(self, input1: torch.Tensor, input2: torch.Tensor, eps: float, iters: int):
        """-> (dist[B,n] squared distances to the assigned points, assignment[B,n] int32)."""
        return emdFunction.apply(input1, input2, eps, iters)
