"""Batch sharding of the structural-loss path over the GPUs of one node.

Every kernel treats batch elements independently (reference: outer loop over ``i<b`` in each kernel,
``nndistance.cu:5``, ``approxmatch.cu:15``), so the path shards with NO data-path collective: each rank
(one process per GPU) takes a contiguous slice of the global batch, exactly the reference's
``batch_size_per_device`` rule (``src/config/specs.py:331-345``).  Collectives (RCCL through
``torch.distributed`` backend ``"nccl"`` on ROCm; ``gloo`` in the CPU tests) appear only where there is a real
exchange: the scalar loss/metric reduction and the optional gradient all-reduce of replicated model
parameters (the reference's DDP wrap, ``src/train/hooks.py:36``).
"""

from __future__ import annotations

from collections.abc import Iterable

import torch
import torch.distributed as dist


def batch_size_per_device(global_batch: int, world_size: int) -> int:
    """``specs.py:331-345``: the global batch must divide evenly over the devices."""
    if world_size <= 0:
        return global_batch
    if global_batch % world_size != 0:
        raise ValueError(f'Global batch size {global_batch} not divisible by number of devices {world_size}.')
    return global_batch // world_size


def shard_slice(global_batch: int, rank: int, world_size: int) -> slice:
    """Contiguous slice of the global batch owned by ``rank``."""
    per = batch_size_per_device(global_batch, world_size)
    return slice(rank * per, (rank + 1) * per)


def global_mean(per_sample: torch.Tensor, group: dist.ProcessGroup | None = None) -> torch.Tensor:
    """Mean of a per-sample loss ``[B_local]`` over the GLOBAL batch: one all-reduce of ``[sum, count]``."""
    acc = torch.stack([per_sample.double().sum(), torch.tensor(float(per_sample.numel()), dtype=torch.float64,
                                                               device=per_sample.device)])
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
    return (acc[0] / acc[1]).to(per_sample.dtype)


def allreduce_mean_(tensors: Iterable[torch.Tensor], group: dist.ProcessGroup | None = None,
                    bucket_bytes: int = 64 << 20) -> None:
    """In-place average of replicated gradients across ranks, flattened into few large buckets.

    xGMI is point-to-point (7 links per GPU): a ring all-reduce is bound by ONE link, so the message count
    matters more than on a switch; ~45 MB of autoencoder gradients go out as a single
    bucket by default.
    """
    if not (dist.is_available() and dist.is_initialized()):
        return
    world = dist.get_world_size(group)
    bucket: list[torch.Tensor] = []
    size = 0

    def flush() -> None:
        nonlocal bucket, size
        if not bucket:
            return
        flat = torch.cat([t.reshape(-1) for t in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat /= world
        off = 0
        for t in bucket:
            t.copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()
        bucket, size = [], 0

    for t in tensors:
        nbytes = t.numel() * t.element_size()
        if bucket and (size + nbytes > bucket_bytes or t.dtype != bucket[0].dtype):
            flush()
        bucket.append(t)
        size += nbytes
    flush()


def max_over_ranks(seconds: float, device: torch.device, group: dist.ProcessGroup | None = None) -> float:
    """Bench timing rule: the step time of a job is the slowest rank's."""
    if not (dist.is_available() and dist.is_initialized()):
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def broadcast_(tensor: torch.Tensor, src: int = 0, group: dist.ProcessGroup | None = None) -> torch.Tensor:
    """In-place broadcast of a replicated tensor from ``src`` (no-op without a process group)."""
    if dist.is_available() and dist.is_initialized():
        dist.broadcast(tensor, src=src, group=group)
    return tensor


@torch.no_grad()
def reseed_unused_codes_(codebook: torch.Tensor, usage_local: torch.Tensor, vq_noise: float, final_epoch: bool = False,
                         generator: torch.Generator | None = None, group: dist.ProcessGroup | None = None) -> int:
    """The reference's ``DiscreteSpaceOptimizer`` step (``src/train/hooks.py:47-77``) for a codebook replicated over
    the ranks: entries of ``codebook[n_books, book_size, dim]`` that no sample selected are re-seeded from a used entry
    of the same book, drawn with probability proportional to its usage, plus ``vq_noise`` Gaussian noise (``:74-77``);
    in the final epoch they are parked at 1000 instead (``:70-72``).

    Two deliberate differences from the reference, both on the multi-GPU side: the usage
    counts ``usage_local[n_books, book_size]`` of every rank's shard are SUMMED first (the reference looks at rank 0's
    shard only, ``:51-55``), and after rank 0 has rewritten the entries the codebook is BROADCAST -- the reference
    mutates ``codebook.data`` on rank 0 alone (``:51``) and its replicas silently diverge.  Returns the number of
    re-seeded entries (the same on every rank)."""
    usage = usage_local.detach().to(torch.float64).clone()
    distributed = dist.is_available() and dist.is_initialized()
    if distributed:
        dist.all_reduce(usage, op=dist.ReduceOp.SUM, group=group)
    rank = dist.get_rank(group) if distributed else 0
    unused = usage == 0
    n_reseeded = int(unused.sum().item())
    if rank == 0 and n_reseeded:
        u_cpu = usage.cpu()
        for book in range(codebook.shape[0]):
            total = float(u_cpu[book].sum())
            if total <= 0:  # nothing of this book was used: no template to draw from (the reference would divide by 0)
                continue
            probs = u_cpu[book] / total
            for entry in torch.nonzero(unused[book].cpu()).flatten().tolist():
                if final_epoch:
                    codebook[book, entry] = 1000
                    continue
                sampled = int(torch.multinomial(probs, 1, generator=generator).item())
                template = codebook[book, sampled]
                noise = torch.randn(template.shape, generator=generator).to(template)
                codebook[book, entry] = template + vq_noise * noise
    broadcast_(codebook, 0, group)
    return n_reseeded
