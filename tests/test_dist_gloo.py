"""world_size-2 ``gloo`` tests of the N>1 path on CPU: contiguous batch sharding with no data-path collective,
the [sum,count] loss reduction, the bucketed gradient all-reduce and the max-over-ranks timing rule."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.util import pair


def _free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, out_dir: str) -> None:
    from pointcloudcounterfactual_amd import sharding
    from pointcloudcounterfactual_amd.losses import torch_chamfer

    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        a, c = pair(99, 4, 128, 96)  # the same global batch on every rank
        sl = sharding.shard_slice(4, rank, world)
        t1 = torch.from_numpy(a[sl]).requires_grad_(True)
        loss = torch_chamfer(t1, torch.from_numpy(c[sl]))  # per-sample losses of the local shard: no collective
        gm = sharding.global_mean(loss.detach())
        # a replicated "model parameter" whose gradient must be averaged across ranks
        w = torch.ones(3, requires_grad=True)
        (loss * (t1.detach() * w).sum((1, 2))).sum().backward()
        grads = [w.grad.clone(), torch.full((5,), float(rank))]
        sharding.allreduce_mean_(grads, bucket_bytes=16)  # tiny bucket: exercises several flushes
        slow = sharding.max_over_ranks(1.0 + rank, torch.device('cpu'))
        # codebook re-seeding (hooks.py:47-77): replicated codebook, per-rank usage counts; entry (0,2) is used on rank 1
        # only, entries (0,3) and (1,0) by nobody -> exactly those two are rewritten, and every rank ends with rank 0's bits
        book = torch.arange(2 * 4 * 3, dtype=torch.float32).reshape(2, 4, 3).clone()
        usage = torch.tensor([[5, 1, 0, 0], [0, 2, 2, 1]]) if rank == 0 else torch.tensor([[1, 0, 7, 0], [0, 1, 0, 3]])
        n_re = sharding.reseed_unused_codes_(book, usage, vq_noise=0.01, generator=torch.Generator().manual_seed(5 + rank))
        book_final = torch.arange(2 * 4 * 3, dtype=torch.float32).reshape(2, 4, 3).clone()
        sharding.reseed_unused_codes_(book_final, usage, vq_noise=0.01, final_epoch=True)
        np.savez(os.path.join(out_dir, f'r{rank}.npz'), loss=loss.detach().numpy(), gm=gm.numpy(),
                 wgrad_local=w.grad.numpy(), wgrad_avg=grads[0].numpy(), other=grads[1].numpy(), slow=slow,
                 book=book.numpy(), n_re=n_re, book_final=book_final.numpy())
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding(tmp_path):
    from pointcloudcounterfactual_amd.losses import torch_chamfer

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / f'r{k}.npz') for k in range(world)]
    a, c = pair(99, 4, 128, 96)
    full = torch_chamfer(torch.from_numpy(a), torch.from_numpy(c)).numpy()
    np.testing.assert_array_equal(np.concatenate([r[0]['loss'], r[1]['loss']]), full)  # shards == unsharded, bit for bit
    for k in range(world):
        np.testing.assert_allclose(r[k]['gm'], full.mean(), rtol=1e-6)
        np.testing.assert_allclose(r[k]['wgrad_avg'], (r[0]['wgrad_local'] + r[1]['wgrad_local']) / 2, rtol=1e-6)
        np.testing.assert_allclose(r[k]['other'], 0.5)
        assert float(r[k]['slow']) == 2.0
    # codebook: both ranks hold the same bits; only the globally unused entries moved, each next to a used entry of its book
    orig = np.arange(24, dtype=np.float32).reshape(2, 4, 3)
    np.testing.assert_array_equal(r[0]['book'], r[1]['book'])
    assert int(r[0]['n_re']) == 2 and int(r[1]['n_re']) == 2
    changed = (r[0]['book'] != orig).any(-1)
    assert changed.tolist() == [[False, False, False, True], [True, False, False, False]]
    assert min(np.abs(r[0]['book'][0, 3] - orig[0, j]).max() for j in (0, 1, 2)) < 0.1
    assert min(np.abs(r[0]['book'][1, 0] - orig[1, j]).max() for j in (1, 2, 3)) < 0.1
    np.testing.assert_array_equal(r[0]['book_final'], r[1]['book_final'])
    assert (r[0]['book_final'][0, 3] == 1000).all() and (r[0]['book_final'][1, 0] == 1000).all()
    assert ((r[0]['book_final'] == 1000).any(-1) == changed).all()


def test_batch_must_divide():
    from pointcloudcounterfactual_amd import sharding

    assert sharding.batch_size_per_device(256, 8) == 32
    assert sharding.shard_slice(256, 3, 8) == slice(96, 128)
    with pytest.raises(ValueError, match='not divisible'):
        sharding.batch_size_per_device(30, 8)
