"""Autograd surface of the structural losses.

``nn_distance`` / ``match_cost`` keep the reference's signatures and autograd behaviour
(``external/pytorch_structural_losses/structural_losses/nn_distance.py:9-43``,
``.../match_cost.py:11-50``); ``chamfer`` is the Chamfer loss the reference's GPU training path
computes with PyKeOps (``src/train/metrics_and_losses.py:21-41``), expressed through ``nn_distance``
(SURVEY.md section 8 row A7); ``torch_chamfer`` is the reference's CPU Chamfer (``:44-47``).
"""

from __future__ import annotations

from typing import Any

import torch
from torch.autograd import Function

from pointcloudcounterfactual_amd import backend


class NNDistanceFunction(Function):
    """``(set1[B,N,3], set2[B,M,3]) -> (dist1[B,N], dist2[B,M])`` squared nearest-neighbour distances."""

    @staticmethod
    def forward(ctx: Any, *args: Any, **kwargs: Any) -> Any:
        set1, set2, *_ = args
        ctx.save_for_backward(set1, set2)
        dist1, idx1, dist2, idx2 = backend.NNDistance(set1, set2)
        ctx.idx1 = idx1  # indices are constants of the backward pass (nn_distance.py:22-24)
        ctx.idx2 = idx2
        return dist1, dist2

    @staticmethod
    def backward(ctx: Any, *grad_outputs: Any) -> Any:
        set1, set2 = ctx.saved_tensors
        grad1, grad2 = backend.NNDistanceGrad(
            set1, set2, ctx.idx1, ctx.idx2, grad_outputs[0].contiguous(), grad_outputs[1].contiguous()
        )
        return grad1, grad2


class MatchCostFunction(Function):
    """``(set1[B,N,3], set2[B,M,3]) -> cost[B]`` approximate earth mover's distance.

    ``mode`` selects how the reference's ApproxMatch -> MatchCost / MatchCostGrad sequence (match_cost.py:25-27,
    39-42) is carried out; the three give the same cost and gradients up to float summation order:

    * ``'implicit'`` (default): ``match`` never exists.  One pass evaluates every match element in registers and
      accumulates the cost and -- when an input requires grad -- both gradients (they depend on the inputs only:
      the reference treats ``match`` as a constant); backward multiplies by ``grad_output``.  Saves the 4*B*M*N-byte
      tensor the reference keeps alive on ``ctx`` (512 MiB at B=32, N=2048) and two full passes over it.
    * ``'fused'``: ``match`` is materialised once (cost accumulated by the same pass) and read once in backward.
    * ``'reference'``: the reference's three backend calls, one after the other.
    """

    mode = 'implicit'

    @staticmethod
    def forward(ctx: Any, *args: torch.Tensor, **kwargs: Any) -> torch.Tensor:
        set1, set2, *_ = args
        mode = MatchCostFunction.mode
        ctx.mode = mode
        if mode == 'implicit':
            with_grad = bool(ctx.needs_input_grad[0] or ctx.needs_input_grad[1])
            out = backend.MatchCostImplicit(set1, set2, with_grad)
            if with_grad:
                ctx.save_for_backward(out[1], out[2])
            return out[0]
        ctx.save_for_backward(set1, set2)
        if mode == 'fused':
            match, _temp, cost = backend.ApproxMatchCost(set1, set2)
        elif mode == 'reference':
            match, _temp = backend.ApproxMatch(set1, set2)
            cost = backend.MatchCost(set1, set2, match)
        else:
            raise ValueError(f'unknown MatchCostFunction.mode {mode!r}')
        ctx.match = match  # kept alive until backward, as the reference does (match_cost.py:26)
        return cost

    @staticmethod
    def backward(ctx: Any, *grad_outputs: Any) -> tuple[torch.Tensor, torch.Tensor]:
        grad_output = grad_outputs[0]
        if ctx.mode == 'implicit':
            grad1, grad2 = ctx.saved_tensors
            scale = grad_output.unsqueeze(1).unsqueeze(2)
            return (grad1 * scale if ctx.needs_input_grad[0] else None,
                    grad2 * scale if ctx.needs_input_grad[1] else None)
        set1, set2 = ctx.saved_tensors
        if ctx.mode == 'fused':  # upstream gradient folded into the reduction of the gradient kernel
            grad1, grad2 = backend.MatchCostGradScaled(set1, set2, ctx.match, grad_output.contiguous().float())
            return grad1, grad2
        grad1, grad2 = backend.MatchCostGrad(set1, set2, ctx.match)
        scale = grad_output.unsqueeze(1).unsqueeze(2)
        return grad1 * scale, grad2 * scale

    @classmethod
    def apply(cls, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:  # type: ignore[override]
        return super().apply(x, y)  # type: ignore[return-value]


nn_distance = NNDistanceFunction.apply
match_cost = MatchCostFunction.apply


class ChamferFunction(Function):
    """``(t1[B,N,3], t2[B,M,3], mean) -> loss[B]``: nearest-neighbour search, loss reduction and -- in backward --
    the spreading of ``grad_loss[b]`` over the points, all inside the library (two launches forward, one backward)
    instead of six small elementwise / reduction launches around ``nn_distance``."""

    @staticmethod
    def forward(ctx: Any, *args: Any, **kwargs: Any) -> torch.Tensor:
        t1, t2, mean = args
        loss, _d1, idx1, _d2, idx2 = backend.ChamferLoss(t1, t2, bool(mean))
        ctx.save_for_backward(t1, t2, idx1, idx2)  # indices are constants of the backward pass
        ctx.mean = bool(mean)
        return loss

    @staticmethod
    def backward(ctx: Any, *grad_outputs: Any) -> Any:
        t1, t2, idx1, idx2 = ctx.saved_tensors
        g = grad_outputs[0]
        grad1, grad2 = backend.ChamferLossGrad(t1, t2, idx1, idx2, g if g.dtype == torch.float32 else g.float(), ctx.mean)
        return grad1, grad2, None


def chamfer(t1: torch.Tensor, t2: torch.Tensor, reduction: str = 'mean') -> torch.Tensor:
    """Chamfer loss ``[B]`` on the accelerator.

    ``reduction='mean'`` reproduces ``pykeops_chamfer`` (metrics_and_losses.py:21-41):
    ``dist2.mean(1) + dist1.mean(1)``; ``'sum'`` reproduces the scale of ``torch_chamfer`` (:44-47).
    Gradients flow through the gathered nearest neighbours only (indices are constants), as in both.
    Equal to the same expression written with ``nn_distance`` (tests/test_gpu_structural.py) up to the order of
    the float sums.
    """
    if reduction not in ('mean', 'sum'):
        raise ValueError(f"reduction must be 'mean' or 'sum', got {reduction!r}")
    return ChamferFunction.apply(t1, t2, reduction == 'mean')


class ChamferEMDFunction(Function):
    """``(t1[B,N,3], t2[B,M,3], mean) -> (chamfer[B], emd[B])``: the two terms of the reference's ``ChamferEMD``
    reconstruction loss (``src/train/metrics_and_losses.py:70-79``: Chamfer and ``match_cost`` on the SAME pair of
    clouds) as one autograd node.  Same kernels and the same bits as ``chamfer(t1, t2)`` and ``match_cost(t1, t2)``;
    what the fusion buys is scheduling:

    * forward: the two losses do not depend on each other, and the approximate EMD is a chain of 19 dependent
      launches whose late passes are latency-bound and leave most of the chip idle.  One library call
      (``pcc_chamfer_emd``) enqueues the nearest-neighbour search (VALU-bound, one big launch) on an internal stream
      that starts when the chain reaches those passes, and joins it before returning.
    * backward: one launch (``pcc_chamfer_emd_grad``) writes the total gradient -- Chamfer's scatter term plus the
      saved ``match_cost`` gradients times their upstream scalar -- instead of three launches and two accumulations.
    """

    @staticmethod
    def forward(ctx: Any, *args: Any, **kwargs: Any) -> Any:
        t1, t2, mean = args
        with_grad = bool(ctx.needs_input_grad[0] or ctx.needs_input_grad[1])
        out = backend.ChamferEMD(t1, t2, bool(mean), with_grad)
        if with_grad:
            ctx.save_for_backward(t1, t2, out[1], out[2], out[4], out[5])
        ctx.mean = bool(mean)
        return out[0], out[3]

    @staticmethod
    def backward(ctx: Any, *grad_outputs: Any) -> Any:
        t1, t2, idx1, idx2, e1, e2 = ctx.saved_tensors
        gc, ge = grad_outputs
        gc = gc if gc.dtype == torch.float32 else gc.float()
        ge = ge if ge.dtype == torch.float32 else ge.float()
        grad1, grad2 = backend.ChamferEMDGrad(t1, t2, idx1, idx2, gc, ctx.mean, e1, e2, ge)
        return (grad1 if ctx.needs_input_grad[0] else None, grad2 if ctx.needs_input_grad[1] else None, None)


def chamfer_emd(t1: torch.Tensor, t2: torch.Tensor, reduction: str = 'mean') -> tuple[torch.Tensor, torch.Tensor]:
    """``(chamfer(t1, t2, reduction), match_cost(t1, t2))`` as one autograd node (see ``ChamferEMDFunction``): the
    reference's ``ChamferEMD`` reconstruction loss is their sum (``metrics_and_losses.py:70-79``)."""
    if reduction not in ('mean', 'sum'):
        raise ValueError(f"reduction must be 'mean' or 'sum', got {reduction!r}")
    return ChamferEMDFunction.apply(t1, t2, reduction == 'mean')


def torch_square_distance(t1: torch.Tensor, t2: torch.Tensor) -> torch.Tensor:
    """Expanded-form squared distances ``[B,N,M]`` (reference ``src/utils/neighbour_ops.py:43-50``)."""
    t2 = t2.transpose(-1, -2)
    dist = -2 * torch.matmul(t1, t2)
    dist += torch.sum(t1**2, -1, keepdim=True)
    dist += torch.sum(t2**2, -2, keepdim=True)
    return dist


def torch_chamfer(t1: torch.Tensor, t2: torch.Tensor) -> torch.Tensor:
    """The reference's CPU Chamfer (sum over points; ``metrics_and_losses.py:44-47``).  This is the
    reference's own host path for ``user.cpu`` runs (BASELINE config 1), not a fallback of ``chamfer``."""
    dist = torch_square_distance(t1, t2)
    return torch.min(dist, dim=-1)[0].sum(1) + torch.min(dist, dim=-2)[0].sum(1)
