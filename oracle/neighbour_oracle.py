"""Torch restatement of the reference's graph ops (TEST INFRASTRUCTURE ONLY): the formulas of
``src/utils/neighbour_ops.py:85-133`` written independently (index arithmetic instead of gather/expand) so that
they can be checked against the vectors generated from the reference's own functions
(``tests/golden/ref_neighbour_ops.npz``) and then serve as the expected values for the HIP kernels."""

from __future__ import annotations

import torch


def gather_neighbours(x: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """out[b,c,n,j] = x[b,c,idx[b,n,j]]   (get_neighbours, :85-94)."""
    b, c, n = x.shape
    bi = torch.arange(b, device=x.device)[:, None, None, None]
    ci = torch.arange(c, device=x.device)[None, :, None, None]
    return x[bi, ci, idx[:, None, :, :]]


def graph_features(x: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """cat([nbr - x, x], 1)   (get_graph_features, :113-119)."""
    nb = gather_neighbours(x, idx)
    xe = x[:, :, :, None].expand_as(nb)
    return torch.cat([nb - xe, xe], dim=1)


def graph_max_pooling(x: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """max over the k neighbours   (graph_max_pooling, :106-110)."""
    return gather_neighbours(x, idx).max(dim=-1)[0]


def graph_filtering(x: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """(1 + sum_j w_j) x - sum_j w_j nbr_j, w = exp(-dist/sigma), sigma = clamp(mean first-nbr dist, 0.005) (:122-133).
    ``idx`` is the k-NN index tensor (self first)."""
    nb = gather_neighbours(x, idx)[..., 1:]
    diff = x[..., None] - nb
    dist = (diff**2).sum(1).abs().sqrt()
    sigma = dist[..., 0:1].mean(1, keepdim=True).clamp(min=0.005)
    w = torch.exp(-(dist / sigma))
    return (1 + w.sum(2))[:, None, :] * x - (w[:, None] * nb).sum(-1)
