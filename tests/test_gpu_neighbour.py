"""GPU parity tests of the kNN-graph primitives (HIP through the C ABI) vs the oracle and the vectors
generated from the reference's own neighbour_ops.py functions."""

import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'ref_neighbour_ops.npz'), allow_pickle=False)


def _x(seed, b, c, n, kind='normal'):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(b, c, n, generator=g) if kind == 'normal' else torch.rand(b, c, n, generator=g)).contiguous()


@pytest.mark.parametrize('b,c,n,k', [(2, 3, 64, 4), (2, 3, 257, 20), (1, 3, 2048, 25), (2, 3, 2100, 25), (3, 1, 100, 8),
                                       (2, 2, 333, 16), (1, 3, 25, 25), (2, 3, 40, 32), (32, 3, 2048, 4)])
def test_knn_small_c_bit_exact(cuda, oracle_mod, b, c, n, k):
    from pointcloudcounterfactual_amd import neighbour_ops as ops

    x = _x(b * 1000 + n + k, b, c, n, 'uniform')
    idx = ops.knn(x.to(cuda), k).cpu().numpy()
    if b * n > 20000:  # oracle on a slice only (it sorts every row)
        sel = slice(0, 2)
        exp = oracle_mod.knn_diff(x[sel].numpy(), k)
        assert np.array_equal(idx[sel], exp)
        exp_last = oracle_mod.knn_diff(x[-1:].numpy(), k)
        assert np.array_equal(idx[-1:], exp_last)
    else:
        assert np.array_equal(idx, oracle_mod.knn_diff(x.numpy(), k))


@pytest.mark.parametrize('b,c,n,k', [(2, 3, 1000, 5), (2, 3, 777, 10), (1, 3, 3000, 23), (2, 3, 500, 30), (1, 3, 5000, 20),
                                       (2, 3, 17, 1), (1, 3, 16, 16), (3, 2, 49, 7), (1, 3, 17000, 8)])
def test_knn_sorted_search_shapes(cuda, oracle_mod, b, c, n, k):
    """The Hilbert-sorted search (c <= 3, n <= 16384): k between the list sizes, clouds that are not a multiple of the
    16-point boxes, several 128-box windows (n > 2048), surface-like clouds (points on a sphere, where the walk ends
    early), and the exhaustive kernel behind it for n > 16384 -- all bit-identical to the oracle's lists."""
    from pointcloudcounterfactual_amd import neighbour_ops as ops

    x = _x(n * 3 + k, b, c, n)
    if c == 3:
        x[0] = x[0] / x[0].norm(dim=0, keepdim=True)  # first sample on the unit sphere
    idx = ops.knn(x.to(cuda), k).cpu().numpy()
    stride = 1 if n <= 3000 else 7 if n <= 5000 else 37  # (the oracle sorts a whole row per query)
    exp = oracle_mod.knn_diff(x.numpy(), k, stride)
    assert np.array_equal(idx[:, ::stride], exp[:, ::stride])


def test_knn_degenerate_clouds(cuda, oracle_mod):
    """All points identical (every distance ties: the lists are 0..k-1 for every query), a single point, and a cloud with
    NaN / inf coordinates (a NaN distance never enters a list; no index outside the cloud is ever emitted)."""
    from pointcloudcounterfactual_amd import neighbour_ops as ops

    same = torch.full((2, 3, 100), 0.25)
    idx = ops.knn(same.to(cuda), 10).cpu().numpy()
    assert (idx == np.arange(10)[None, None, :]).all()
    one = torch.rand(3, 3, 1)
    assert (ops.knn(one.to(cuda), 1).cpu().numpy() == 0).all()
    x = _x(5, 2, 3, 200)
    x[0, 1, 17] = float('nan')
    x[1, 0, 3] = float('inf')
    idx = ops.knn(x.to(cuda), 8).cpu().numpy()
    assert idx.min() >= 0 and idx.max() < 200
    ok = np.ones(200, bool)
    ok[17] = False
    # queries other than the NaN point: the NaN candidate is never chosen, the rest as the oracle on the cloud without it
    assert not (idx[0, ok] == 17).any()
    clean = torch.cat([x[0, :, :17], x[0, :, 18:]], dim=1)[None]
    ref = oracle_mod.knn_diff(clean.numpy(), 8)[0]
    ref = ref + (ref >= 17)  # indices of the cloud with point 17 removed -> original numbering
    assert np.array_equal(idx[0, ok], ref)


def test_knn_sorted_search_random_sweep(cuda, oracle_mod):
    """Seeded sweep of the sorted search over awkward sizes: clouds around the 16-point box and 2048-point window
    boundaries, every list size and k in between, 1-3 channels, Gaussian / uniform / planar / clustered clouds."""
    from pointcloudcounterfactual_amd import neighbour_ops as ops

    rng = np.random.default_rng(77)
    sizes = [1, 2, 15, 16, 17, 31, 33, 63, 64, 65, 127, 129, 255, 1000, 2047, 2048, 2049, 2063, 2065, 4097]
    for trial in range(40):
        n = int(sizes[trial % len(sizes)] if trial < 20 else rng.integers(1, 2600))
        k = int(min(n, rng.integers(1, 33)))
        b = int(rng.integers(1, 4))
        c = int(rng.integers(1, 4))
        kind = trial % 4
        x = rng.standard_normal((b, c, n)).astype(np.float32)
        if kind == 1:
            x = rng.random((b, c, n)).astype(np.float32)
        elif kind == 2 and c == 3:
            x[:, 2] = 0.0  # planar
        elif kind == 3:
            x = (x * 0.01 + rng.integers(0, 3, (b, c, n))).astype(np.float32)  # tight clusters on a lattice
        idx = ops.knn(torch.from_numpy(x).to(cuda), k).cpu().numpy()
        stride = 1 if n <= 2600 else 5
        exp = oracle_mod.knn_diff(x, k, stride)
        assert np.array_equal(idx[:, ::stride], exp[:, ::stride]), (trial, b, c, n, k, kind)


def test_inputs_that_are_views_with_a_storage_offset(cuda):
    """Contiguous tensors that start 4 bytes into their storage (slices of a larger buffer): every 16-byte fast path must
    notice and every result must equal the one for an aligned copy."""
    from pointcloudcounterfactual_amd import backend
    from pointcloudcounterfactual_amd import neighbour_ops as ops

    def shifted(t):
        buf = torch.empty(t.numel() + 1, dtype=t.dtype, device=t.device)
        v = buf[1:].view(t.shape)
        v.copy_(t)
        assert v.is_contiguous() and v.data_ptr() % 16 != 0
        return v

    g = torch.Generator().manual_seed(5)
    x3 = torch.rand(3, 3, 1000, generator=g).to(cuda)
    x64 = torch.randn(3, 64, 1000, generator=g).to(cuda)
    a, c = torch.rand(3, 1000, 3, generator=g).to(cuda), torch.rand(3, 900, 3, generator=g).to(cuda)
    idx = ops.hip_knn(x64, 20)
    assert torch.equal(ops.hip_knn(shifted(x3), 20), ops.hip_knn(x3, 20))
    assert torch.equal(ops.hip_knn(shifted(x64), 20), idx)
    for fn in (lambda x, i: ops.get_graph_features(x, i, 20)[1], lambda x, i: ops.graph_max_pooling(x, i, 20),
               lambda x, i: ops.get_neighbours(x, i, 20)[1]):
        want = fn(x64, idx)
        assert torch.equal(fn(shifted(x64), idx), want) and torch.equal(fn(x64, shifted(idx)), want)
    want = ops.global_max_pool(x64)
    got = ops.global_max_pool(shifted(x64))
    assert all(torch.equal(p, q) for p, q in zip(got if isinstance(got, (tuple, list)) else [got],
                                                 want if isinstance(want, (tuple, list)) else [want]))
    for f in (lambda p, q: backend.NNDistance(p, q), lambda p, q: backend.MatchCostImplicit(p, q, True),
              lambda p, q: backend.ChamferEMD(p, q, True, True)):
        want = f(a, c)
        got = f(shifted(a), shifted(c))
        assert all(torch.equal(p, q) for p, q in zip(got, want))


def test_knn_ties_ascending_index(cuda, oracle_mod):
    from pointcloudcounterfactual_amd import neighbour_ops as ops

    base = _x(3, 1, 3, 50, 'uniform')
    x = torch.cat([base] * 6, dim=2).contiguous()  # every point 6 times -> 6-way exact ties everywhere
    idx = ops.knn(x.to(cuda), 8).cpu().numpy()
    assert np.array_equal(idx, oracle_mod.knn_diff(x.numpy(), 8))
    assert (idx[0, :, 0] == np.arange(300) % 50).all()  # lowest copy first


@pytest.mark.parametrize('b,c,n,k', [(2, 4, 64, 4), (2, 6, 200, 16), (2, 64, 257, 25), (1, 64, 2048, 25), (2, 128, 300, 20),
                                       (1, 128, 2048, 25), (2, 17, 131, 8), (2, 64, 2050, 25), (2, 32, 500, 30)])
@pytest.mark.parametrize('kernel', [1, 2])
def test_knn_mfma_vs_oracle(cuda, oracle_mod, b, c, n, k, kernel):
    """c >= 4: expanded form on the f32 MFMA pipe.  v_mfma_f32_32x32x2_f32 is a k-ordered fmaf chain, so the
    distances (and therefore the sorted index lists) must match the sequential-fma oracle bit for bit -- in the
    128-query kernel (1) and in the role-split kernel that large launches take (2)."""
    from pointcloudcounterfactual_amd import _lib, neighbour_ops as ops

    x = _x(c * 7 + n, b, c, n)
    _lib.set_tuning('knn_nosplit', kernel)
    try:
        idx = ops.knn(x.to(cuda), k).cpu().numpy()
    finally:
        _lib.set_tuning('knn_nosplit', 0)
    exp = oracle_mod.knn_expanded(x.numpy(), k)
    assert np.array_equal(idx, exp)


def test_knn_mfma_many_small_clouds_take_the_role_split_kernel(cuda, oracle_mod):
    """A batch of many small clouds fills the chip with 256-query workgroups that are mostly padding (40 of 256 queries,
    two stages): the product's own dispatch, no switch."""
    from pointcloudcounterfactual_amd import neighbour_ops as ops

    x = _x(99, 200, 8, 40)
    idx = ops.knn(x.to(cuda), 5).cpu().numpy()
    assert np.array_equal(idx, oracle_mod.knn_expanded(x.numpy(), 5))


@pytest.mark.parametrize('kernel', [1, 2])
def test_knn_mfma_adversarial_orders(cuda, oracle_mod, kernel):
    """Candidate orders that stress the selection: distances descending along the index (every candidate displaces an
    entry of every list), ascending (the first k stay), exact ties everywhere, and a query count that leaves the last
    workgroup mostly empty."""
    from pointcloudcounterfactual_amd import _lib, neighbour_ops as ops

    n, c = 700, 5
    t = torch.linspace(0, 1, n)
    line = torch.stack([t, 2 * t, -t, 0.5 * t, t * t], 0)[None]                      # points along a curve, in index order
    rev = line.flip(2)
    ties = torch.cat([_x(11, 1, c, 70)] * 10, dim=2)                                 # every point ten times
    for x, k in ((line, 25), (rev, 25), (ties, 16), (torch.cat([line, rev, ties], 0), 20)):
        x = x.contiguous()
        _lib.set_tuning('knn_nosplit', kernel)
        try:
            idx = ops.knn(x.to(cuda), k).cpu().numpy()
        finally:
            _lib.set_tuning('knn_nosplit', 0)
        assert np.array_equal(idx, oracle_mod.knn_expanded(x.numpy(), k))


@pytest.mark.parametrize('tag', ['c3', 'c3k20', 'c64'])
def test_knn_vs_reference_fixture(cuda, tag):
    """Against the reference's own torch_knn outputs: identical neighbour lists up to near ties."""
    from pointcloudcounterfactual_amd import neighbour_ops as ops
    from tests.test_neighbour_oracle import _near_tie_ok

    x, k = GOLD[f'{tag}_x'], int(GOLD[f'{tag}_k'])
    ref_idx, ref_dist = GOLD[f'{tag}_knn'], GOLD[f'{tag}_selfdist']
    idx = ops.knn(torch.from_numpy(x).to(cuda), k).cpu().numpy()
    bad = sum(not _near_tie_ok(ref_dist[b, q], idx[b, q], ref_idx[b, q], 1e-4)
              for b in range(x.shape[0]) for q in range(x.shape[2]))
    assert bad == 0
    exact = (idx == ref_idx).mean()
    assert exact > 0.995, exact


@pytest.mark.parametrize('b,c,n,k', [(2, 3, 64, 4), (2, 64, 257, 20), (1, 128, 512, 25), (3, 5, 100, 7)])
def test_graph_ops_forward_backward(cuda, b, c, n, k):
    from oracle import neighbour_oracle as no
    from pointcloudcounterfactual_amd import neighbour_ops as ops

    x = _x(n + c, b, c, n)
    idx = ops.knn(x.to(cuda), k)
    for name, fn_gpu, fn_ref in (
        ('gather', lambda t: ops.get_neighbours(t, idx, k)[1], lambda t: no.gather_neighbours(t, idx.cpu())),
        ('features', lambda t: ops.get_graph_features(t, idx, k)[1], lambda t: no.graph_features(t, idx.cpu())),
        ('maxpool', lambda t: ops.graph_max_pooling(t, idx, k), lambda t: no.graph_max_pooling(t, idx.cpu())),
        ('globalmax', ops.global_max_pool, lambda t: t.max(dim=2)[0]),
    ):
        tg = x.to(cuda).requires_grad_(True)
        tr = x.clone().requires_grad_(True)
        og, orf = fn_gpu(tg), fn_ref(tr)
        assert torch.equal(og.cpu(), orf), name  # pure copies / max: exact
        w = torch.randn(orf.shape, generator=torch.Generator().manual_seed(1))
        (og * w.to(cuda)).sum().backward()
        (orf * w).sum().backward()
        torch.testing.assert_close(tg.grad.cpu(), tr.grad, rtol=1e-5, atol=1e-5, msg=name)


@pytest.mark.parametrize('b,c,n,k,kind', [(2, 64, 2048, 25, 'knn'), (2, 5, 700, 7, 'random'), (1, 3, 300, 4, 'hub'),
                                            (2, 9, 1000, 20, 'invalid'), (1, 2, 64, 100, 'random'), (1, 3, 5000, 3, 'random')])
def test_gather_and_feature_backward_edge_stream(cuda, b, c, n, k, kind):
    """Backward of get_neighbours / get_graph_features (neighbour_ops.py:85-119: torch.gather's scatter_add) on the
    streaming kernels (edge_chunk_sort_kernel + edge_stream_bwd_kernel): several chunks of source points, ragged last
    chunk, arbitrary (non-kNN) graphs, a hub every point links to (segments that span many waves' runs), neighbour
    indices outside [0, n) (replaced by the point itself, as in the forward), k large enough for one-wave chunks --
    against float64 scatter_add."""
    from pointcloudcounterfactual_amd import neighbour_ops as ops

    g = torch.Generator().manual_seed(n + k)
    x = torch.randn(b, c, n, generator=g)
    if kind == 'knn':
        idx = ops.knn(x.to(cuda), k).cpu()
    else:
        idx = torch.randint(0, n, (b, n, k), generator=g)
        if kind == 'hub':
            idx[:, :, 0] = 7
            idx[:, : n // 2, 1:] = 3
        if kind == 'invalid':
            idx[:, ::5, 2] = n + 3
            idx[:, 1::7, 0] = -1
    self_idx = torch.arange(n).view(1, n, 1).expand(b, n, k)
    eff = torch.where((idx >= 0) & (idx < n), idx, self_idx)  # what the kernels do with an index outside the cloud
    for name, width in (('gather', c), ('features', 2 * c)):
        w = torch.randn(b, width, n, k, generator=g)
        tg = x.to(cuda).requires_grad_(True)
        out = (ops.get_neighbours(tg, idx.to(cuda), k)[1] if name == 'gather' else ops.get_graph_features(tg, idx.to(cuda), k)[1])
        out.backward(w.to(cuda))
        wd = w.double()
        ref = torch.zeros(b, c, n, dtype=torch.float64)
        ref.scatter_add_(2, eff.reshape(b, 1, n * k).expand(b, c, n * k), wd[:, :c].reshape(b, c, n * k))
        if name == 'features':
            ref += (wd[:, c:] - wd[:, :c]).sum(3)
        scale = ref.abs().max().item()
        assert (tg.grad.cpu().double() - ref).abs().max().item() <= 2e-6 * scale + 1e-6, (name, kind)


def test_graph_ops_vs_reference_fixture(cuda):
    from pointcloudcounterfactual_amd import neighbour_ops as ops

    for tag in ('c3', 'c3k20'):
        x = torch.from_numpy(GOLD[f'{tag}_x']).to(cuda)
        idx = torch.from_numpy(GOLD[f'{tag}_knn']).to(cuda)
        k = int(GOLD[f'{tag}_k'])
        assert torch.equal(ops.get_graph_features(x, idx, k)[1].cpu(), torch.from_numpy(GOLD[f'{tag}_graph_features']))
        assert torch.equal(ops.graph_max_pooling(x, idx, k).cpu(), torch.from_numpy(GOLD[f'{tag}_max_pool']))
    x = torch.from_numpy(GOLD['c64_x']).to(cuda)
    idx = torch.from_numpy(GOLD['c64_knn']).to(cuda)
    assert torch.equal(ops.graph_max_pooling(x, idx, 25).cpu(), torch.from_numpy(GOLD['c64_max_pool']))


def test_graph_filtering(cuda):
    from oracle import neighbour_oracle as no
    from pointcloudcounterfactual_amd import neighbour_ops as ops

    out = ops.graph_filtering(torch.from_numpy(GOLD['filt_x']).to(cuda), k=4)
    torch.testing.assert_close(out.cpu(), torch.from_numpy(GOLD['filt_out']), rtol=1e-5, atol=1e-6)
    # gradient flows through the gathered neighbours exactly as in the torch composition
    x = _x(5, 2, 3, 300, 'uniform')
    tg = x.to(cuda).requires_grad_(True)
    ops.graph_filtering(tg, 4).pow(2).sum().backward()
    tr = x.clone().requires_grad_(True)
    idx = ops.knn(x.to(cuda), 4).cpu()
    no.graph_filtering(tr, idx).pow(2).sum().backward()
    torch.testing.assert_close(tg.grad.cpu(), tr.grad, rtol=1e-4, atol=1e-5)


def test_global_max_mean_pool(cuda):
    from pointcloudcounterfactual_amd import neighbour_ops as ops

    x = _x(9, 4, 96, 1000)
    out = ops.global_max_mean_pool(x.to(cuda)).cpu()
    assert torch.equal(out[:, :96], x.max(dim=2)[0])
    torch.testing.assert_close(out[:, 96:], x.mean(dim=2), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('b,c,cout,n,k,act', [(2, 3, 16, 200, 8, False), (2, 16, 32, 257, 20, True), (1, 64, 64, 512, 25, True)])
def test_fused_edgeconv_matches_unfused_block(cuda, b, c, cout, n, k, act):
    """FusedEdgeConv == get_graph_features -> Conv2d -> BatchNorm2d(train) -> LeakyReLU -> max over k: outputs, gradients
    w.r.t. the input and every parameter, and the running statistics."""
    from pointcloudcounterfactual_amd import neighbour_ops as ops
    from pointcloudcounterfactual_amd.edgeconv import FusedEdgeConv, reference_edgeconv

    torch.manual_seed(c + n)
    x = _x(n + c, b, c, n).to(cuda)
    idx = ops.knn(x, k)
    fused = FusedEdgeConv(c, cout, act=act).to(cuda).train()
    with torch.no_grad():
        fused.bn.weight.copy_(torch.randn(cout, device=cuda))             # both signs of the BatchNorm scale
        fused.bn.bias.copy_(torch.randn(cout, device=cuda))
    import copy
    ref = copy.deepcopy(fused)
    xf = x.clone().requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    of = fused(xf, idx)
    orf = reference_edgeconv(xr, idx, ref.conv, ref.bn, ref.act)
    torch.testing.assert_close(of, orf, rtol=2e-4, atol=2e-4)
    w = torch.randn_like(of)
    (of * w).sum().backward()
    (orf * w).sum().backward()
    torch.testing.assert_close(xf.grad, xr.grad, rtol=2e-3, atol=2e-4)
    torch.testing.assert_close(fused.conv.weight.grad, ref.conv.weight.grad, rtol=2e-3, atol=2e-3)
    torch.testing.assert_close(fused.bn.weight.grad, ref.bn.weight.grad, rtol=2e-3, atol=2e-3)
    torch.testing.assert_close(fused.bn.bias.grad, ref.bn.bias.grad, rtol=2e-3, atol=2e-3)
    torch.testing.assert_close(fused.bn.running_mean, ref.bn.running_mean, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(fused.bn.running_var, ref.bn.running_var, rtol=1e-4, atol=1e-5)
    fused.eval(); ref.eval()
    with torch.no_grad():
        torch.testing.assert_close(fused(x, idx), reference_edgeconv(x, idx, ref.conv, ref.bn, ref.act), rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize('b,c,n,k', [(2, 5, 300, 7), (3, 64, 2048, 25), (1, 3, 64, 4), (2, 9, 1000, 20)])
def test_neighbour_sum_forward_backward(cuda, b, c, n, k):
    """neighbour_sum (sum_j y[b,c,idx[b,i,j]]) and its backward -- the sorted-edge schedule (edges counting-sorted by
    target, segmented sums, one LDS atomic per segment) -- against a dense torch gather / scatter_add, including hubs
    (every point also lists point 0, so bin 0 collects n edges)."""
    from pointcloudcounterfactual_amd.edgeconv import neighbour_sum

    g = torch.Generator().manual_seed(b * 1000 + n)
    y = torch.randn(b, c, n, generator=g).to(cuda).requires_grad_(True)
    idx = torch.randint(0, n, (b, n, k), generator=g)
    idx[:, :, 0] = 0  # a hub
    idx = idx.to(cuda)
    w = torch.randn(b, c, n, generator=g).to(cuda)
    out = neighbour_sum(y, idx)
    (out * w).sum().backward()
    yd = y.detach().double().requires_grad_(True)
    gathered = torch.gather(yd.unsqueeze(3).expand(-1, -1, -1, k), 2, idx.unsqueeze(1).expand(-1, c, -1, -1))
    ref = gathered.sum(3)
    (ref * w.double()).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().cpu().numpy(), rtol=1e-5, atol=1e-5)
    scale = float(yd.grad.abs().max())
    np.testing.assert_allclose(y.grad.cpu().numpy(), yd.grad.cpu().numpy(), rtol=1e-5, atol=1e-5 * scale)
