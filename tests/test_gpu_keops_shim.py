"""GPU parity tests of the ``pykeops`` drop-in: the reference's three PyKeOps call sites, restated here with the
shim's LazyTensor exactly as the reference writes them (the reference tree does not exist on the GPU box), against
float64 brute force.  No PyKeOps output exists anywhere to compare with: "parity unpinned" w.r.t. PyKeOps itself."""

import numpy as np
import pytest
import torch

from tests.util import pair

pytestmark = pytest.mark.gpu


def _square_distance(t1, t2):
    """pykeops_square_distance, src/utils/neighbour_ops.py:35-40."""
    from pykeops.torch import LazyTensor

    return ((LazyTensor(t1[:, :, None, :]) - LazyTensor(t2[:, None, :, :])) ** 2).sum(-1)


def _pykeops_chamfer(t1, t2):
    """pykeops_chamfer, src/train/metrics_and_losses.py:32-41."""
    dist = _square_distance(t1, t2)
    idx1 = dist.argmin(axis=1).expand(-1, -1, t1.shape[2])
    m1 = t1.gather(1, idx1)
    squared1 = ((t2 - m1) ** 2).sum(2).mean(1)
    idx2 = dist.argmin(axis=2).expand(-1, -1, t1.shape[2])
    m2 = t2.gather(1, idx2)
    squared2 = ((t1 - m2) ** 2).sum(2).mean(1)
    return squared1 + squared2


@pytest.mark.parametrize('b,n,m', [(2, 5, 7), (3, 257, 130), (2, 2048, 2048)])
def test_pykeops_chamfer_restated(cuda, b, n, m):
    from pointcloudcounterfactual_amd.losses import chamfer

    a, c = pair(600 + n, b, n, m)
    t1 = torch.from_numpy(a).to(cuda).requires_grad_(True)
    t2 = torch.from_numpy(c).to(cuda).requires_grad_(True)
    loss = _pykeops_chamfer(t1, t2)
    loss.sum().backward()
    D = ((torch.from_numpy(a).double()[:, :, None, :] - torch.from_numpy(c).double()[:, None, :, :]) ** 2).sum(-1)
    expect = D.min(2)[0].mean(1) + D.min(1)[0].mean(1)
    np.testing.assert_allclose(loss.detach().cpu().numpy(), expect.numpy(), rtol=1e-5)
    # the same loss and gradients as chamfer() / nn_distance
    u1 = torch.from_numpy(a).to(cuda).requires_grad_(True)
    u2 = torch.from_numpy(c).to(cuda).requires_grad_(True)
    ref = chamfer(u1, u2)
    ref.sum().backward()
    np.testing.assert_allclose(loss.detach().cpu().numpy(), ref.detach().cpu().numpy(), rtol=1e-5)
    np.testing.assert_allclose(t1.grad.cpu().numpy(), u1.grad.cpu().numpy(), rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(t2.grad.cpu().numpy(), u2.grad.cpu().numpy(), rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize('c,k', [(3, 4), (3, 20), (64, 25)])
def test_pykeops_knn_restated(cuda, c, k):
    """pykeops_knn, neighbour_ops.py:77-82: transpose, lazy self distance, argKmin."""
    from pointcloudcounterfactual_amd.neighbour_ops import hip_knn

    g = torch.Generator().manual_seed(7 + c)
    x = torch.randn(2, c, 300, generator=g).to(cuda)
    xt = x.transpose(2, 1).contiguous()
    idx = _square_distance(xt, xt).argKmin(k, dim=2)
    assert idx.shape == (2, 300, k) and idx.dtype == torch.int64
    assert torch.equal(idx, hip_knn(x, k))
    assert torch.equal(idx[:, :, 0], torch.arange(300, device=cuda).expand(2, -1))  # the point itself first
    D = torch.cdist(xt.double(), xt.double()) ** 2
    kth = D.gather(2, idx)  # ascending, and no outsider is closer than the k-th
    assert (kth[:, :, 1:] >= kth[:, :, :-1] - 1e-6).all()
    assert (D.topk(k, largest=False)[0][:, :, -1] - kth[:, :, -1]).abs().max() < 1e-4 * max(1.0, float(D.max()))


@pytest.mark.parametrize('batch,n_codes,book,d', [(4, 8, 16, 4), (32, 256, 16, 4), (3, 5, 7, 11)])
def test_vector_quantizer_restated(cuda, batch, n_codes, book, d):
    """VectorQuantizer.quantize, src/module/quantize.py:20-32, on the shim; gradients of dist_sum against torch."""
    g = torch.Generator().manual_seed(11)
    x = torch.randn(batch, n_codes * d, generator=g).to(cuda).requires_grad_(True)
    codebook = torch.randn(n_codes, book, d, generator=g).to(cuda).requires_grad_(True)
    x_flat = x.view(batch * n_codes, 1, d)
    book_repeated = codebook.repeat(batch, 1, 1)
    dist = _square_distance(x_flat, book_repeated)
    idx_flat = dist.argmin(axis=2)
    dist_sum = dist.sum(1).view(batch, n_codes, book)
    assert idx_flat.shape == (batch * n_codes, 1, 1) and idx_flat.dtype == torch.int64
    w = torch.rand(batch, n_codes, book, generator=g).to(cuda)
    (dist_sum * w).sum().backward()
    # dense float64 evaluation of the same quantities
    xd = x.detach().double().view(batch * n_codes, 1, d).requires_grad_(True)
    cd = codebook.detach().double().requires_grad_(True)
    dense = ((xd[:, :, None, :] - cd.repeat(batch, 1, 1)[:, None, :, :]) ** 2).sum(-1)  # [B', 1, book]
    assert torch.equal(idx_flat.view(-1), dense.argmin(2).view(-1))
    dense_sum = dense.sum(1).view(batch, n_codes, book)
    np.testing.assert_allclose(dist_sum.detach().cpu().numpy(), dense_sum.detach().cpu().numpy(), rtol=1e-5, atol=1e-6)
    (dense_sum * w.double()).sum().backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), xd.grad.view(batch, n_codes * d).cpu().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(codebook.grad.cpu().numpy(), cd.grad.cpu().numpy(), rtol=1e-4, atol=1e-4)


def test_general_dimension_argmin_and_sum_axes(cuda):
    """Non-3-D clouds take pcc_pair_argmin; both reduction axes; ties go to the lowest index."""
    g = torch.Generator().manual_seed(3)
    t1 = torch.randn(2, 40, 5, generator=g).to(cuda)
    t2 = torch.randn(2, 33, 5, generator=g)
    t2[:, 20] = t2[:, 4]  # duplicate candidate: index 4 must win over 20
    t2 = t2.to(cuda)
    dist = _square_distance(t1, t2)
    D = ((t1.double()[:, :, None, :] - t2.double()[:, None, :, :]) ** 2).sum(-1)
    a2, a1 = dist.argmin(axis=2), dist.argmin(dim=1)
    assert a2.shape == (2, 40, 1) and a1.shape == (2, 33, 1)
    assert not (a2 == 20).any()
    np.testing.assert_allclose(D.gather(2, a2).squeeze(-1).cpu().numpy(), D.min(2)[0].cpu().numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(D.gather(1, a1.transpose(1, 2)).squeeze(1).cpu().numpy(), D.min(1)[0].cpu().numpy(),
                               rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(dist.min(axis=2).squeeze(-1).cpu().numpy(), D.min(2)[0].cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(dist.sum(axis=2).squeeze(-1).cpu().numpy(), D.sum(2).cpu().numpy(), rtol=1e-5)
    np.testing.assert_allclose(dist.sum(axis=1).squeeze(-1).cpu().numpy(), D.sum(1).cpu().numpy(), rtol=1e-5)


def test_neighbour_ops_square_distance_exports(cuda):
    """``square_distance`` / ``pykeops_square_distance`` / ``pykeops_knn`` as the drop-in ``neighbour_ops`` exports them
    (reference neighbour_ops.py:27-40,77-82; imported by metrics_and_losses.py:18 and quantize.py:6): the lazy distance's
    reductions equal the dense float64 ones (indices exactly, away from ties; minima to 1e-6)."""
    from pointcloudcounterfactual_amd import neighbour_ops as ops

    a, c = pair(611, 2, 300, 257)
    t1, t2 = torch.from_numpy(a).to(cuda), torch.from_numpy(c).to(cuda)
    dist = ops.square_distance(t1, t2)
    assert type(dist) is type(ops.pykeops_square_distance(t1, t2)) and dist.shape == (2, 300, 257)
    D = ((torch.from_numpy(a).double()[:, :, None, :] - torch.from_numpy(c).double()[:, None, :, :]) ** 2).sum(-1)
    assert torch.equal(dist.argmin(axis=2).squeeze(-1).cpu(), D.argmin(2))
    assert torch.equal(dist.argmin(axis=1).squeeze(-1).cpu(), D.argmin(1))
    np.testing.assert_allclose(dist.min(axis=2).squeeze(-1).cpu().numpy(), D.min(2)[0].numpy(), rtol=1e-5, atol=1e-9)
    x = t1.transpose(1, 2).contiguous()
    assert torch.equal(ops.pykeops_knn(x, 5), ops.hip_knn(x, 5))
    # the CPU side of the dispatch is the reference's dense expanded form
    assert torch.allclose(ops.square_distance(t1.cpu(), t2.cpu()), D.float(), atol=1e-5)
