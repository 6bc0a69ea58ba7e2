"""Host-side mirror of the reference's ``structural_losses.structural_losses_backend`` pybind module
(``external/pytorch_structural_losses/src/structural_loss.cpp:24-135``): the same five functions, the
same argument meaning, output shapes/dtypes, ownership (fresh ``torch.empty`` outputs) and error
behaviour (``RuntimeError`` when an input is not a contiguous accelerator tensor), enqueuing on the
current stream without synchronising.  The arithmetic is the C-ABI library's HIP kernels; PyTorch is
only used for device memory and the stream handle.
"""

from __future__ import annotations

import torch

from pointcloudcounterfactual_amd import _lib

_L = _lib.lib


def _check_input(x: torch.Tensor, name: str) -> None:
    # CHECK_INPUT, structural_loss.cpp:6-8
    if not x.device.type == 'cuda':
        raise RuntimeError(f'{name} must be a CUDA tensor')
    if not x.is_contiguous():
        raise RuntimeError(f'{name} must be contiguous')


def _f32(x: torch.Tensor, name: str) -> None:
    # the reference's data_ptr<float>() throws for any other dtype
    if x.dtype != torch.float32:
        raise RuntimeError(f'expected scalar type Float but found {x.dtype} ({name})')


def _i32(x: torch.Tensor, name: str) -> None:
    if x.dtype != torch.int32:
        raise RuntimeError(f'expected scalar type Int but found {x.dtype} ({name})')


def _stream(x: torch.Tensor) -> int:
    return torch.cuda.current_stream(x.device).cuda_stream


def _sizes(set_d: torch.Tensor, set_q: torch.Tensor) -> tuple[int, int, int]:
    return set_d.size(0), set_d.size(1), set_q.size(1)


def ApproxMatch(set_d: torch.Tensor, set_q: torch.Tensor) -> list[torch.Tensor]:
    """-> [match[B,M,N], temp[B,2(N+M)]]   (structural_loss.cpp:24-38)."""
    b, n, m = _sizes(set_d, set_q)
    match = torch.empty((b, m, n), dtype=torch.float32, device=set_d.device)
    temp = torch.empty((b, (n + m) * 2), dtype=torch.float32, device=set_d.device)
    _check_input(set_d, 'set_d')
    _check_input(set_q, 'set_q')
    _f32(set_d, 'set_d')
    _f32(set_q, 'set_q')
    with torch.cuda.device(set_d.device):
        _lib.check(_L.pcc_approxmatch(b, n, m, set_d.data_ptr(), set_q.data_ptr(), match.data_ptr(),
                                      temp.data_ptr(), _stream(set_d)), 'ApproxMatch')
    return [match, temp]


def ApproxMatchCost(set_d: torch.Tensor, set_q: torch.Tensor) -> list[torch.Tensor]:
    """ApproxMatch + MatchCost in one pass over ``match`` -> [match, temp, cost[B]] (extension)."""
    b, n, m = _sizes(set_d, set_q)
    match = torch.empty((b, m, n), dtype=torch.float32, device=set_d.device)
    temp = torch.empty((b, (n + m) * 2), dtype=torch.float32, device=set_d.device)
    cost = torch.empty((b,), dtype=torch.float32, device=set_d.device)
    _check_input(set_d, 'set_d')
    _check_input(set_q, 'set_q')
    _f32(set_d, 'set_d')
    _f32(set_q, 'set_q')
    with torch.cuda.device(set_d.device):
        _lib.check(_L.pcc_approxmatch_cost(b, n, m, set_d.data_ptr(), set_q.data_ptr(), match.data_ptr(),
                                           temp.data_ptr(), cost.data_ptr(), _stream(set_d)), 'ApproxMatchCost')
    return [match, temp, cost]


def MatchCostImplicit(set_d: torch.Tensor, set_q: torch.Tensor, with_grad: bool) -> list[torch.Tensor]:
    """``match_cost`` without the match tensor (extension, ``pcc_match_cost``): -> [cost[B]] or, ``with_grad``,
    [cost[B], grad1[B,N,3], grad2[B,M,3]] where the gradients are those of ``MatchCostGrad`` (upstream gradient 1).
    Every match element is evaluated in registers and consumed on the spot; nothing of size B*M*N touches HBM."""
    b, n, m = _sizes(set_d, set_q)
    dev = set_d.device
    cost = torch.empty((b,), dtype=torch.float32, device=dev)
    out = [cost]
    g1 = g2 = None
    if with_grad:
        g1 = torch.empty((b, n, 3), dtype=torch.float32, device=dev)
        g2 = torch.empty((b, m, 3), dtype=torch.float32, device=dev)
        out += [g1, g2]
    _check_input(set_d, 'set_d')
    _check_input(set_q, 'set_q')
    _f32(set_d, 'set_d')
    _f32(set_q, 'set_q')
    with torch.cuda.device(dev):
        _lib.check(_L.pcc_match_cost(b, n, m, set_d.data_ptr(), set_q.data_ptr(), None, cost.data_ptr(),
                                     g1.data_ptr() if with_grad else None, g2.data_ptr() if with_grad else None,
                                     _stream(set_d)), 'MatchCostImplicit')
    return out


def MatchCost(set_d: torch.Tensor, set_q: torch.Tensor, match: torch.Tensor) -> torch.Tensor:
    """-> cost[B]   (structural_loss.cpp:40-53)."""
    b, n, m = _sizes(set_d, set_q)
    out = torch.empty((b,), dtype=torch.float32, device=set_d.device)
    _check_input(set_d, 'set_d')
    _check_input(set_q, 'set_q')
    _check_input(match, 'match')
    for t, name in ((set_d, 'set_d'), (set_q, 'set_q'), (match, 'match')):
        _f32(t, name)
    if match.numel() != b * n * m:
        raise RuntimeError(f'match has {match.numel()} elements, expected {b}x{m}x{n}')
    with torch.cuda.device(set_d.device):
        _lib.check(_L.pcc_matchcost(b, n, m, set_d.data_ptr(), set_q.data_ptr(), match.data_ptr(), out.data_ptr(),
                                    _stream(set_d)), 'MatchCost')
    return out


def MatchCostGrad(set_d: torch.Tensor, set_q: torch.Tensor, match: torch.Tensor) -> list[torch.Tensor]:
    """-> [grad1[B,N,3], grad2[B,M,3]]   (structural_loss.cpp:55-70)."""
    b, n, m = _sizes(set_d, set_q)
    grad1 = torch.empty((b, n, 3), dtype=torch.float32, device=set_d.device)
    grad2 = torch.empty((b, m, 3), dtype=torch.float32, device=set_d.device)
    _check_input(set_d, 'set_d')
    _check_input(set_q, 'set_q')
    _check_input(match, 'match')
    for t, name in ((set_d, 'set_d'), (set_q, 'set_q'), (match, 'match')):
        _f32(t, name)
    if match.numel() != b * n * m:
        raise RuntimeError(f'match has {match.numel()} elements, expected {b}x{m}x{n}')
    with torch.cuda.device(set_d.device):
        _lib.check(_L.pcc_matchcostgrad(b, n, m, set_d.data_ptr(), set_q.data_ptr(), match.data_ptr(),
                                        grad1.data_ptr(), grad2.data_ptr(), _stream(set_d)), 'MatchCostGrad')
    return [grad1, grad2]


def MatchCostGradScaled(set_d: torch.Tensor, set_q: torch.Tensor, match: torch.Tensor,
                        grad_cost: torch.Tensor) -> list[torch.Tensor]:
    """MatchCostGrad with the upstream gradient ``grad_cost[B]`` folded into the reduction (extension): equals
    ``grad * grad_cost[:, None, None]`` of the reference wrapper (match_cost.py:41-42) without the two extra passes."""
    b, n, m = _sizes(set_d, set_q)
    grad1 = torch.empty((b, n, 3), dtype=torch.float32, device=set_d.device)
    grad2 = torch.empty((b, m, 3), dtype=torch.float32, device=set_d.device)
    for t, name in ((set_d, 'set_d'), (set_q, 'set_q'), (match, 'match'), (grad_cost, 'grad_cost')):
        _check_input(t, name)
        _f32(t, name)
    if match.numel() != b * n * m or grad_cost.numel() != b:
        raise RuntimeError('MatchCostGradScaled: match / grad_cost shapes do not match the clouds')
    with torch.cuda.device(set_d.device):
        _lib.check(_L.pcc_matchcostgrad_scaled(b, n, m, set_d.data_ptr(), set_q.data_ptr(), match.data_ptr(),
                                               grad_cost.data_ptr(), grad1.data_ptr(), grad2.data_ptr(),
                                               _stream(set_d)), 'MatchCostGradScaled')
    return [grad1, grad2]


def NNDistance(set_d: torch.Tensor, set_q: torch.Tensor) -> list[torch.Tensor]:
    """-> [dist1[B,N] f32, idx1[B,N] i32, dist2[B,M] f32, idx2[B,M] i32]   (structural_loss.cpp:81-100)."""
    b, n, m = _sizes(set_d, set_q)
    dev = set_d.device
    dist1 = torch.empty((b, n), dtype=torch.float32, device=dev)
    idx1 = torch.empty((b, n), dtype=torch.int32, device=dev)
    dist2 = torch.empty((b, m), dtype=torch.float32, device=dev)
    idx2 = torch.empty((b, m), dtype=torch.int32, device=dev)
    _check_input(set_d, 'set_d')
    _check_input(set_q, 'set_q')
    _f32(set_d, 'set_d')
    _f32(set_q, 'set_q')
    with torch.cuda.device(dev):
        _lib.check(_L.pcc_nndistance(b, n, set_d.data_ptr(), m, set_q.data_ptr(), dist1.data_ptr(), idx1.data_ptr(),
                                     dist2.data_ptr(), idx2.data_ptr(), _stream(set_d)), 'NNDistance')
    return [dist1, idx1, dist2, idx2]


def ChamferLoss(set_d: torch.Tensor, set_q: torch.Tensor, mean: bool) -> list[torch.Tensor]:
    """NNDistance + the loss reduction in one call (extension, ``pcc_chamfer_loss``):
    -> [loss[B], dist1, idx1, dist2, idx2] with loss = mean_j dist1 + mean_k dist2 (``mean``) or the sums."""
    b, n, m = _sizes(set_d, set_q)
    dev = set_d.device
    loss = torch.empty((b,), dtype=torch.float32, device=dev)
    dist1 = torch.empty((b, n), dtype=torch.float32, device=dev)
    idx1 = torch.empty((b, n), dtype=torch.int32, device=dev)
    dist2 = torch.empty((b, m), dtype=torch.float32, device=dev)
    idx2 = torch.empty((b, m), dtype=torch.int32, device=dev)
    _check_input(set_d, 'set_d')
    _check_input(set_q, 'set_q')
    _f32(set_d, 'set_d')
    _f32(set_q, 'set_q')
    with torch.cuda.device(dev):
        _lib.check(_L.pcc_chamfer_loss(b, n, set_d.data_ptr(), m, set_q.data_ptr(), int(mean), loss.data_ptr(),
                                       dist1.data_ptr(), idx1.data_ptr(), dist2.data_ptr(), idx2.data_ptr(),
                                       _stream(set_d)), 'ChamferLoss')
    return [loss, dist1, idx1, dist2, idx2]


def ChamferLossGrad(set_d: torch.Tensor, set_q: torch.Tensor, idx1: torch.Tensor, idx2: torch.Tensor,
                    grad_loss: torch.Tensor, mean: bool) -> list[torch.Tensor]:
    """NNDistanceGrad with ``grad_dist1[b,:] = grad_loss[b] (/N)``, ``grad_dist2[b,:] = grad_loss[b] (/M)`` formed
    inside the kernel (extension, ``pcc_chamfer_loss_grad``) -> [grad1[B,N,3], grad2[B,M,3]]."""
    b, n, m = _sizes(set_d, set_q)
    grad1 = torch.empty((b, n, 3), dtype=torch.float32, device=set_d.device)
    grad2 = torch.empty((b, m, 3), dtype=torch.float32, device=set_d.device)
    for t, name in ((set_d, 'set_d'), (set_q, 'set_q'), (idx1, 'idx1'), (idx2, 'idx2')):
        _check_input(t, name)
    if grad_loss.device.type != 'cuda':
        raise RuntimeError('grad_loss must be a CUDA tensor')
    for t, name in ((set_d, 'set_d'), (set_q, 'set_q'), (grad_loss, 'grad_loss')):
        _f32(t, name)
    _i32(idx1, 'idx1')
    _i32(idx2, 'idx2')
    if idx1.numel() != b * n or idx2.numel() != b * m or grad_loss.numel() != b:
        raise RuntimeError('ChamferLossGrad: idx / grad_loss shapes do not match the clouds')
    # one scalar expanded over the batch (loss.sum().backward()) is read in place through stride 0: no copy kernel
    stride = grad_loss.stride(0) if grad_loss.dim() == 1 and b > 1 else 1
    if stride not in (0, 1):
        grad_loss = grad_loss.contiguous()
        stride = 1
    with torch.cuda.device(set_d.device):
        _lib.check(_L.pcc_chamfer_loss_grad(b, n, set_d.data_ptr(), m, set_q.data_ptr(), idx1.data_ptr(),
                                            idx2.data_ptr(), grad_loss.data_ptr(), int(stride), int(mean), grad1.data_ptr(),
                                            grad2.data_ptr(), _stream(set_d)), 'ChamferLossGrad')
    return [grad1, grad2]


def ChamferEMD(set_d: torch.Tensor, set_q: torch.Tensor, mean: bool, with_grad: bool,
               return_dist: bool = False) -> list[torch.Tensor]:
    """``ChamferLoss`` and ``MatchCostImplicit`` on the same pair of clouds in ONE call (extension, ``pcc_chamfer_emd``;
    the reference's ChamferEMD loss, metrics_and_losses.py:70-79) -> [chamfer[B], idx1, idx2, emd[B]] (+ [emd_grad1,
    emd_grad2] ``with_grad``) (+ [dist1, dist2] ``return_dist``).  Same bits as the two calls; the nearest-neighbour
    search runs on the clouds the approximate EMD has just Hilbert-sorted, with box culling instead of the exhaustive
    scan."""
    b, n, m = _sizes(set_d, set_q)
    dev = set_d.device
    _check_input(set_d, 'set_d')
    _check_input(set_q, 'set_q')
    _f32(set_d, 'set_d')
    _f32(set_q, 'set_q')
    loss = torch.empty((b,), dtype=torch.float32, device=dev)
    idx1 = torch.empty((b, n), dtype=torch.int32, device=dev)
    idx2 = torch.empty((b, m), dtype=torch.int32, device=dev)
    cost = torch.empty((b,), dtype=torch.float32, device=dev)
    out = [loss, idx1, idx2, cost]
    g1 = g2 = None
    if with_grad:
        g1 = torch.empty((b, n, 3), dtype=torch.float32, device=dev)
        g2 = torch.empty((b, m, 3), dtype=torch.float32, device=dev)
        out += [g1, g2]
    d1 = torch.empty((b, n), dtype=torch.float32, device=dev)  # per-point distances: scratch of the reduction
    d2 = torch.empty((b, m), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_L.pcc_chamfer_emd(b, n, set_d.data_ptr(), m, set_q.data_ptr(), int(mean), loss.data_ptr(), d1.data_ptr(),
                                      idx1.data_ptr(), d2.data_ptr(), idx2.data_ptr(), cost.data_ptr(),
                                      g1.data_ptr() if with_grad else None, g2.data_ptr() if with_grad else None,
                                      _stream(set_d)), 'ChamferEMD')
    if return_dist:
        out += [d1, d2]
    return out


def _batch_stride(g: torch.Tensor, b: int) -> tuple[torch.Tensor, int]:
    """An upstream gradient [B] as (tensor, stride): one scalar expanded over the batch (what ``loss.sum().backward()``
    hands down) is read in place through stride 0, no copy kernel."""
    stride = g.stride(0) if g.dim() == 1 and b > 1 else 1
    if stride not in (0, 1):
        return g.contiguous(), 1
    return g, stride


def ChamferEMDGrad(set_d: torch.Tensor, set_q: torch.Tensor, idx1: torch.Tensor, idx2: torch.Tensor,
                   grad_chamfer: torch.Tensor, mean: bool, emd_grad1: torch.Tensor, emd_grad2: torch.Tensor,
                   grad_emd: torch.Tensor) -> list[torch.Tensor]:
    """Backward of Chamfer + match_cost on the same clouds in ONE launch (extension, ``pcc_chamfer_emd_grad``):
    ``ChamferLossGrad(grad_chamfer) + emd_grad * grad_emd[:, None, None]`` -> [grad1[B,N,3], grad2[B,M,3]]."""
    b, n, m = _sizes(set_d, set_q)
    grad1 = torch.empty((b, n, 3), dtype=torch.float32, device=set_d.device)
    grad2 = torch.empty((b, m, 3), dtype=torch.float32, device=set_d.device)
    for t, name in ((set_d, 'set_d'), (set_q, 'set_q'), (idx1, 'idx1'), (idx2, 'idx2'), (emd_grad1, 'emd_grad1'),
                    (emd_grad2, 'emd_grad2')):
        _check_input(t, name)
    for t, name in ((set_d, 'set_d'), (set_q, 'set_q'), (grad_chamfer, 'grad_chamfer'), (grad_emd, 'grad_emd'),
                    (emd_grad1, 'emd_grad1'), (emd_grad2, 'emd_grad2')):
        if t.device.type != 'cuda':
            raise RuntimeError(f'{name} must be a CUDA tensor')
        _f32(t, name)
    _i32(idx1, 'idx1')
    _i32(idx2, 'idx2')
    if (idx1.numel() != b * n or idx2.numel() != b * m or grad_chamfer.numel() != b or grad_emd.numel() != b
            or emd_grad1.numel() != b * n * 3 or emd_grad2.numel() != b * m * 3):
        raise RuntimeError('ChamferEMDGrad: shapes do not match the clouds')
    grad_chamfer, sc = _batch_stride(grad_chamfer, b)
    grad_emd, se = _batch_stride(grad_emd, b)
    with torch.cuda.device(set_d.device):
        _lib.check(_L.pcc_chamfer_emd_grad(b, n, set_d.data_ptr(), m, set_q.data_ptr(), idx1.data_ptr(), idx2.data_ptr(),
                                           grad_chamfer.data_ptr(), int(sc), int(mean), emd_grad1.data_ptr(),
                                           emd_grad2.data_ptr(), grad_emd.data_ptr(), int(se), grad1.data_ptr(),
                                           grad2.data_ptr(), _stream(set_d)), 'ChamferEMDGrad')
    return [grad1, grad2]


def NNDistanceGrad(set_d: torch.Tensor, set_q: torch.Tensor, idx1: torch.Tensor, idx2: torch.Tensor,
                   grad_dist1: torch.Tensor, grad_dist2: torch.Tensor) -> list[torch.Tensor]:
    """-> [grad1[B,N,3], grad2[B,M,3]]   (structural_loss.cpp:102-125)."""
    b, n, m = _sizes(set_d, set_q)
    grad1 = torch.empty((b, n, 3), dtype=torch.float32, device=set_d.device)
    grad2 = torch.empty((b, m, 3), dtype=torch.float32, device=set_d.device)
    for t, name in ((set_d, 'set_d'), (set_q, 'set_q'), (idx1, 'idx1'), (idx2, 'idx2'),
                    (grad_dist1, 'grad_dist1'), (grad_dist2, 'grad_dist2')):
        _check_input(t, name)
    for t, name in ((set_d, 'set_d'), (set_q, 'set_q'), (grad_dist1, 'grad_dist1'), (grad_dist2, 'grad_dist2')):
        _f32(t, name)
    _i32(idx1, 'idx1')
    _i32(idx2, 'idx2')
    if idx1.numel() != b * n or grad_dist1.numel() != b * n or idx2.numel() != b * m or grad_dist2.numel() != b * m:
        raise RuntimeError('NNDistanceGrad: idx/grad_dist shapes do not match the clouds')
    with torch.cuda.device(set_d.device):
        _lib.check(_L.pcc_nndistancegrad(b, n, set_d.data_ptr(), m, set_q.data_ptr(), grad_dist1.data_ptr(),
                                         idx1.data_ptr(), grad_dist2.data_ptr(), idx2.data_ptr(), grad1.data_ptr(),
                                         grad2.data_ptr(), _stream(set_d)), 'NNDistanceGrad')
    return [grad1, grad2]
