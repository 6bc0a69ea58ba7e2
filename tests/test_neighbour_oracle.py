"""CPU tests: the kNN / graph-op oracle against vectors produced by the REFERENCE's own functions
(tests/golden/ref_neighbour_ops.npz, generated from /root/reference/src/utils/neighbour_ops.py), and the
reference-API wrappers on CPU tensors."""

import os

import numpy as np
import pytest
import torch

GOLD = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'ref_neighbour_ops.npz'), allow_pickle=False)
TAGS = ['c3', 'c3k20', 'c64']


def _near_tie_ok(dist_row: np.ndarray, got: np.ndarray, exp: np.ndarray, rtol: float) -> bool:
    """Two k-lists may differ only by candidates whose distance ties the k-th within rounding, in any order."""
    if np.array_equal(got, exp):
        return True
    kth = max(dist_row[got].max(), dist_row[exp].max())
    tol = rtol * max(abs(kth), 1e-6) + 1e-6
    for a, e in zip(got, exp):
        if a != e and abs(dist_row[a] - dist_row[e]) > tol:
            return False
    return set(got) - set(exp) == set() or all(abs(dist_row[j] - kth) <= tol for j in set(got) ^ set(exp))


@pytest.mark.parametrize('tag', TAGS)
def test_oracle_knn_matches_reference_torch_knn(oracle_mod, tag):
    x, k = GOLD[f'{tag}_x'], int(GOLD[f'{tag}_k'])
    ref_idx, ref_dist = GOLD[f'{tag}_knn'], GOLD[f'{tag}_selfdist']
    idx, dist = oracle_mod.knn_expanded(x, k, return_dist=True)
    # our sequential-fma expanded distances vs torch's bmm: same numbers to f32 rounding of a C-term dot product
    scale = np.abs(ref_dist).max()
    assert np.abs(dist - ref_dist).max() <= 2e-6 * scale * max(1, x.shape[1] ** 0.5)
    bad = 0
    for b in range(x.shape[0]):
        for q in range(x.shape[2]):
            if not _near_tie_ok(ref_dist[b, q], idx[b, q], ref_idx[b, q], 1e-5):
                bad += 1
    assert bad == 0
    # difference form (the GPU reference's formula) selects the same neighbours up to near ties
    idx_d = oracle_mod.knn_diff(x, k)
    bad = sum(not _near_tie_ok(ref_dist[b, q], idx_d[b, q], ref_idx[b, q], 1e-4)
              for b in range(x.shape[0]) for q in range(x.shape[2]))
    assert bad == 0
    assert (idx_d[:, :, 0] == np.arange(x.shape[2])).all()  # the point itself comes first


@pytest.mark.parametrize('tag', TAGS)
def test_oracle_graph_ops_match_reference(tag):
    from oracle import neighbour_oracle as no

    x = torch.from_numpy(GOLD[f'{tag}_x'])
    idx = torch.from_numpy(GOLD[f'{tag}_knn'])
    if f'{tag}_graph_features' in GOLD:
        torch.testing.assert_close(no.graph_features(x, idx), torch.from_numpy(GOLD[f'{tag}_graph_features']), rtol=0, atol=0)
    torch.testing.assert_close(no.graph_max_pooling(x, idx), torch.from_numpy(GOLD[f'{tag}_max_pool']), rtol=0, atol=0)


def test_oracle_graph_filtering_matches_reference(oracle_mod):
    from oracle import neighbour_oracle as no

    x = torch.from_numpy(GOLD['filt_x'])
    idx = torch.from_numpy(oracle_mod.knn_expanded(GOLD['filt_x'], 4))
    torch.testing.assert_close(no.graph_filtering(x, idx), torch.from_numpy(GOLD['filt_out']), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('tag', TAGS)
def test_cpu_wrappers_reproduce_reference(tag):
    """The reference-API module on CPU tensors == the reference's outputs bit for bit (same torch formulas)."""
    from pointcloudcounterfactual_amd import neighbour_ops as ops

    x = torch.from_numpy(GOLD[f'{tag}_x'])
    k = int(GOLD[f'{tag}_k'])
    idx = ops.knn(x, k)
    assert torch.equal(idx, torch.from_numpy(GOLD[f'{tag}_knn']))
    torch.testing.assert_close(ops.self_square_distance(x), torch.from_numpy(GOLD[f'{tag}_selfdist']), rtol=0, atol=0)
    idx2, feat = ops.get_graph_features(x, torch.empty(0), k)
    assert torch.equal(idx2, idx)
    if f'{tag}_graph_features' in GOLD:
        assert torch.equal(feat, torch.from_numpy(GOLD[f'{tag}_graph_features']))
    assert torch.equal(ops.graph_max_pooling(x, idx, k), torch.from_numpy(GOLD[f'{tag}_max_pool']))


def test_cpu_graph_filtering_reproduces_reference():
    from pointcloudcounterfactual_amd import neighbour_ops as ops

    out = ops.graph_filtering(torch.from_numpy(GOLD['filt_x']).clone(), k=4)
    torch.testing.assert_close(out, torch.from_numpy(GOLD['filt_out']), rtol=0, atol=0)


def test_cpu_get_local_covariance_reproduces_reference():
    """get_local_covariance (reference neighbour_ops.py:97-103) on CPU tensors against the reference's own output
    (the reference subtracts the neighbourhood mean in place; ours out of place: same values)."""
    from pointcloudcounterfactual_amd import neighbour_ops as ops

    x = torch.from_numpy(GOLD['cov_x']).clone()
    out = ops.get_local_covariance(x, torch.empty(0), int(GOLD['cov_k']))
    assert torch.equal(x, torch.from_numpy(GOLD['cov_x']))  # the input is not modified
    torch.testing.assert_close(out, torch.from_numpy(GOLD['cov_out']), rtol=0, atol=0)
