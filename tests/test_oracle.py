"""CPU tests of the oracle (no GPU): the C restatement against float64 brute force, analytic known
answers, the committed golden fixtures, and the structural properties listed in SURVEY.md section 8(c)."""

import os

import numpy as np
import pytest

from tests.util import pair

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.mark.parametrize('n,m', [(1, 1), (3, 5), (64, 64), (257, 130), (512, 513), (513, 512), (1024, 1024), (2, 1500)])
def test_nndistance_vs_float64(oracle_mod, n, m):
    a, c = pair(n * 7 + m, 2, n, m, 'uniform')
    d1, i1, d2, i2 = oracle_mod.nndistance(a, c)
    e1, j1 = oracle_mod.nndistance_f64(a, c)
    e2, j2 = oracle_mod.nndistance_f64(c, a)
    # distances agree to f32 rounding; indices agree except at float64-certified near ties
    np.testing.assert_allclose(d1, e1, rtol=3e-7, atol=1e-12)
    np.testing.assert_allclose(d2, e2, rtol=3e-7, atol=1e-12)
    for (i, j, e, x, y) in ((i1, j1, e1, a, c), (i2, j2, e2, c, a)):
        bad = np.argwhere(i != j)
        for (bb, q) in bad:
            dq = ((y[bb, i[bb, q]].astype(np.float64) - x[bb, q]) ** 2).sum()
            assert abs(dq - e[bb, q]) <= 1e-6 * max(e[bb, q], 1e-12)  # a genuine near tie


def test_nndistance_tie_rule_and_chunk_boundaries(oracle_mod):
    """Lowest index wins, also across the 512-candidate chunks of nndistance.cu:6,116."""
    base = np.random.default_rng(3).random((1, 5, 3), dtype=np.float32)
    cand = np.concatenate([base] * 300, axis=1)  # 1500 candidates = 3 chunks, each point repeated 300x
    d1, i1, _, _ = oracle_mod.nndistance(base, cand)
    assert (d1 == 0).all() and (i1 == np.arange(5)).all()


def test_nndistance_contraction_modes_only_differ_at_near_ties(oracle_mod):
    a, c = pair(77, 2, 700, 900, 'uniform')
    ref = oracle_mod.nndistance(a, c)
    try:
        for mode in (1, 2):
            oracle_mod.set_contraction(mode)
            alt = oracle_mod.nndistance(a, c)
            np.testing.assert_allclose(alt[0], ref[0], rtol=3e-7, atol=1e-12)
            diff = np.argwhere(alt[1] != ref[1])
            for bb, q in diff:  # different winner only when the two candidates are within rounding
                da = ((c[bb, alt[1][bb, q]].astype(np.float64) - a[bb, q]) ** 2).sum()
                dr = ((c[bb, ref[1][bb, q]].astype(np.float64) - a[bb, q]) ** 2).sum()
                assert abs(da - dr) <= 1e-6 * max(dr, 1e-12)
    finally:
        oracle_mod.set_contraction(0)


def test_nndistancegrad_vs_dense_float64(oracle_mod):
    a, c = pair(5, 2, 120, 77)
    rng = np.random.default_rng(0)
    g1 = rng.standard_normal((2, 120)).astype(np.float32)
    g2 = rng.standard_normal((2, 77)).astype(np.float32)
    _, i1, _, i2 = oracle_mod.nndistance(a, c)
    r1, r2 = oracle_mod.nndistancegrad(a, c, i1, i2, g1, g2)
    e1 = np.zeros((2, 120, 3))
    e2 = np.zeros((2, 77, 3))
    for b in range(2):
        for j in range(120):
            t = 2.0 * g1[b, j] * (a[b, j].astype(np.float64) - c[b, i1[b, j]])
            e1[b, j] += t
            e2[b, i1[b, j]] -= t
        for k in range(77):
            t = 2.0 * g2[b, k] * (c[b, k].astype(np.float64) - a[b, i2[b, k]])
            e2[b, k] += t
            e1[b, i2[b, k]] -= t
    np.testing.assert_allclose(r1, e1, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(r2, e2, rtol=1e-5, atol=1e-6)


def test_approxmatch_single_point_known_answer(oracle_mod):
    p = np.array([[[0, 0, 0]]], np.float32)
    q = np.array([[[0.3, 0.4, 0]]], np.float32)
    match, temp = oracle_mod.approxmatch(p, q)
    assert match.shape == (1, 1, 1) and abs(match[0, 0, 0] - 1) < 1e-6
    np.testing.assert_allclose(oracle_mod.matchcost(p, q, match), [0.5], rtol=1e-6)
    g1, g2 = oracle_mod.matchcostgrad(p, q, match)
    np.testing.assert_allclose(g1, [[[-0.6, -0.8, 0]]], atol=1e-6)
    np.testing.assert_allclose(g2, [[[0.6, 0.8, 0]]], atol=1e-6)


@pytest.mark.parametrize('n,m,row_cap,col_cap', [(256, 256, 1, 1), (256, 128, 2, 1), (128, 256, 1, 2), (257, 130, 1, 1)])
def test_approxmatch_mass_conservation(oracle_mod, n, m, row_cap, col_cap):
    """sum_k match[l,k] <= multiR and sum_l match[l,k] <= multiL with integer-division multipliers
    (approxmatch.cu:6-12); total mass ~ the scarcer side's capacity for overlapping clouds."""
    a, c = pair(n + m, 2, n, m, 'uniform')
    match, temp = oracle_mod.approxmatch(a, c)
    assert match.min() >= 0
    assert match.sum(2).max() <= row_cap + 1e-5  # per query point l (set2)
    assert match.sum(1).max() <= col_cap + 1e-5  # per dataset point k (set1)
    cap = min(m * row_cap, n * col_cap)
    assert (match.sum((1, 2)) > 0.95 * cap).all() and (match.sum((1, 2)) <= cap * (1 + 1e-5)).all()
    # temp = remainL | remainR | ratioL | ratioR ; what is left + what was matched = initial capacity
    np.testing.assert_allclose(temp[:, :n] + match.sum(1), col_cap, atol=2e-5)


def test_approxmatch_identical_clouds_and_far_clouds(oracle_mod):
    a, _ = pair(4, 1, 200, 200, 'uniform')
    match, _ = oracle_mod.approxmatch(a, a)
    assert np.abs(np.diagonal(match[0]) - 1).max() < 1e-3  # all mass on the diagonal
    assert oracle_mod.matchcost(a, a, match)[0] < 1e-2
    far = a + np.float32(25.0)  # |d|^2 > 350: even exp(-0.25 d2) underflows -> no mass moves (reference quirk)
    match, _ = oracle_mod.approxmatch(a, far)
    assert match.sum() < 1e-20 and oracle_mod.matchcost(a, far, match)[0] < 1e-18


def test_approxmatch_permutation_equivariance(oracle_mod):
    a, c = pair(8, 1, 150, 100)
    match, _ = oracle_mod.approxmatch(a, c)
    perm = np.random.default_rng(1).permutation(150)
    match_p, _ = oracle_mod.approxmatch(a[:, perm], c)
    np.testing.assert_allclose(match_p, match[:, :, perm], atol=2e-4)
    np.testing.assert_allclose(oracle_mod.matchcost(a[:, perm], c, match_p), oracle_mod.matchcost(a, c, match), rtol=1e-5)


def test_matchcost_and_grad_vs_float64(oracle_mod):
    a, c = pair(21, 2, 300, 170)
    match, _ = oracle_mod.approxmatch(a, c)
    m64 = match.astype(np.float64)
    np.testing.assert_allclose(oracle_mod.matchcost(a, c, match), oracle_mod.matchcost_f64(a, c, m64), rtol=2e-6)
    g1, g2 = oracle_mod.matchcostgrad(a, c, match)
    h1, h2 = oracle_mod.matchcostgrad_f64(a, c, m64)
    np.testing.assert_allclose(g1, h1, rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(g2, h2, rtol=1e-4, atol=2e-6)


def test_matchcostgrad_is_the_gradient_of_cost_at_fixed_match(oracle_mod):
    """Finite differences of cost(set1) with match held fixed (what the reference differentiates)."""
    a, c = pair(2, 1, 20, 16)
    match, _ = oracle_mod.approxmatch(a, c)
    m64 = match.astype(np.float64)
    g1, _ = oracle_mod.matchcostgrad_f64(a, c, m64)
    eps = 1e-3
    for (j, ax) in ((0, 0), (7, 1), (19, 2)):
        ap, am = a.copy(), a.copy()
        ap[0, j, ax] += eps
        am[0, j, ax] -= eps
        fd = (oracle_mod.matchcost_f64(ap, c, m64)[0] - oracle_mod.matchcost_f64(am, c, m64)[0]) / (
            float(ap[0, j, ax]) - float(am[0, j, ax]))
        assert abs(fd - g1[0, j, ax]) < 1e-3 * max(1.0, abs(fd))


def test_oracle_matches_committed_fixtures(oracle_mod):
    """Regression pin: the committed vectors were produced by this oracle (tests/golden/make_golden.py)."""
    z = np.load(os.path.join(GOLD, 'oracle_structural.npz'), allow_pickle=False)
    for tag in 'abcde':
        s1, s2 = z[f'{tag}_set1'], z[f'{tag}_set2']
        d1, i1, d2, i2 = oracle_mod.nndistance(s1, s2)
        assert np.array_equal(d1, z[f'{tag}_dist1']) and np.array_equal(i1, z[f'{tag}_idx1'])
        assert np.array_equal(d2, z[f'{tag}_dist2']) and np.array_equal(i2, z[f'{tag}_idx2'])
        match, temp = oracle_mod.approxmatch(s1, s2)
        np.testing.assert_allclose(match, z[f'{tag}_match'], rtol=0, atol=1e-6)
        np.testing.assert_allclose(oracle_mod.matchcost(s1, s2, match), z[f'{tag}_cost'], rtol=1e-6)


def test_reference_torch_chamfer_fixture():
    """BASELINE config 1 (CPU sum-Chamfer): our restatement vs vectors produced by the reference's own
    torch_square_distance (src/utils/neighbour_ops.py:43-50) + torch_chamfer body (metrics_and_losses.py:46-47)."""
    import torch

    from pointcloudcounterfactual_amd.losses import torch_chamfer, torch_square_distance

    z = np.load(os.path.join(GOLD, 'ref_neighbour_ops.npz'), allow_pickle=False)
    t1, t2 = torch.from_numpy(z['cd_t1']), torch.from_numpy(z['cd_t2'])
    torch.testing.assert_close(torch_square_distance(t1, t2), torch.from_numpy(z['cd_dist']), rtol=0, atol=0)
    torch.testing.assert_close(torch_chamfer(t1, t2), torch.from_numpy(z['cd_chamfer_sum']), rtol=0, atol=0)


def test_config1_cpu_chamfer_plumbing(oracle_mod):
    """BASELINE configs[0]: N=1024 B=4 Chamfer-only on the CPU reference path, checked against the oracle's
    difference-form nearest neighbours (expanded vs difference form agree to ~1e-6 absolute)."""
    import torch

    from pointcloudcounterfactual_amd.losses import torch_chamfer

    a, c = pair(1234 + 1, 4, 1024, 1024)
    t1 = torch.from_numpy(a).requires_grad_(True)
    loss = torch_chamfer(t1, torch.from_numpy(c))
    loss.sum().backward()
    d1, i1, d2, i2 = oracle_mod.nndistance(a, c)
    np.testing.assert_allclose(loss.detach().numpy(), d1.sum(1) + d2.sum(1), rtol=2e-5)
    g1, _ = oracle_mod.nndistancegrad(a, c, i1, i2, np.ones_like(d1), np.ones_like(d2))
    np.testing.assert_allclose(t1.grad.numpy(), g1, rtol=1e-3, atol=2e-5)


def test_oracle_nndistance_pinned_by_reference_generated_vectors(oracle_mod):
    """The one piece of this path the reference can execute without CUDA: torch_square_distance
    (src/utils/neighbour_ops.py:43-50) + the torch_chamfer body (metrics_and_losses.py:46-47) and their autograd
    gradients, captured by tests/golden/make_golden.py from the imported reference at BASELINE configs[0]'s exact
    shape (B=4, N=1024) and at a small ragged shape.  The oracle's nearest neighbours (and therefore the HIP kernel's,
    which are bit-exact against the oracle) must be the reference's: equal indices outside float64-certified near
    ties, distances within 1e-5 relative + the reference's own cancellation floor (8 ulp of |p|^2 + |q|^2), loss
    1e-5, gradients 1e-5 of the largest component."""
    z = np.load(os.path.join(GOLD, 'ref_neighbour_ops.npz'), allow_pickle=False)
    cases = {
        'cfg1': (z['cfg1_t1'], z['cfg1_t2'], z['cfg1_dist1'], z['cfg1_idx1'], z['cfg1_dist2'], z['cfg1_idx2'],
                 z['cfg1_chamfer_sum']),
        'cd': (z['cd_t1'], z['cd_t2'], z['cd_dist'].min(2), z['cd_dist'].argmin(2), z['cd_dist'].min(1),
               z['cd_dist'].argmin(1), z['cd_chamfer_sum']),
    }
    for tag, (a, c, rd1, ri1, rd2, ri2, rloss) in cases.items():
        floor = 8 * np.finfo(np.float32).eps * float((a ** 2).sum(-1).max() + (c ** 2).sum(-1).max())
        d1, i1, d2, i2 = oracle_mod.nndistance(a, c)
        for (p, q, io, ir) in ((a, c, i1, ri1), (c, a, i2, ri2)):
            for b, j in np.argwhere(io != ir):
                do = ((p[b, j].astype(np.float64) - q[b, io[b, j]]) ** 2).sum()
                dr = ((p[b, j].astype(np.float64) - q[b, ir[b, j]]) ** 2).sum()
                assert abs(do - dr) <= floor, (tag, b, j)
        np.testing.assert_allclose(d1, rd1, rtol=1e-5, atol=floor)
        np.testing.assert_allclose(d2, rd2, rtol=1e-5, atol=floor)
        loss = d1.astype(np.float64).sum(1) + d2.astype(np.float64).sum(1)
        np.testing.assert_allclose(loss, rloss, rtol=1e-5)
        if tag == 'cfg1' and np.array_equal(i1, ri1) and np.array_equal(i2, ri2):
            g1, g2 = oracle_mod.nndistancegrad(a, c, i1, i2, np.ones_like(d1), np.ones_like(d2))
            scale = max(np.abs(z['cfg1_grad1']).max(), np.abs(z['cfg1_grad2']).max())
            np.testing.assert_allclose(g1, z['cfg1_grad1'], rtol=1e-5, atol=1e-5 * scale)
            np.testing.assert_allclose(g2, z['cfg1_grad2'], rtol=1e-5, atol=1e-5 * scale)
