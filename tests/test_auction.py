"""Auction EMD: CPU validity tests of the oracle and GPU parity (HIP vs oracle, bit-exact because every race
of the reference is resolved the same deterministic way in both)."""

import numpy as np
import pytest
import torch

from tests.util import pair


def _clouds(seed, b, n):
    a, c = pair(seed, b, n, n, 'uniform')  # the auction needs coordinates in [0,1]
    return a, c


def test_oracle_auction_validity(oracle_mod):
    a, c = _clouds(1, 2, 1024)
    dist, ass, price = oracle_mod.auction_forward(a, c, 0.005, 50)
    assert ass.min() >= 0 and ass.max() < 1024
    d_chk = ((a - np.take_along_axis(c, ass[..., None].astype(np.int64), 1)) ** 2).sum(-1)
    np.testing.assert_allclose(dist, d_chk, rtol=1e-5, atol=1e-9)
    # near-bijection: few targets are shared after 50 iterations with a forced last one
    for b in range(2):
        assert len(np.unique(ass[b])) > 0.9 * 1024
    # the auction cost is close to (and not below) the optimum given by the Hungarian algorithm
    from scipy.optimize import linear_sum_assignment

    for b in range(2):
        cost = np.sqrt(((a[b][:, None, :] - c[b][None, :, :]) ** 2).sum(-1))
        r, col = linear_sum_assignment(cost)
        opt = cost[r, col].mean()
        got = np.sqrt(dist[b]).mean()
        assert got <= opt * 1.25 + 0.005
        if len(np.unique(ass[b])) == 1024:
            assert got >= opt - 1e-6


def test_oracle_auction_converged_is_a_permutation(oracle_mod):
    a, c = _clouds(2, 1, 1024)
    dist, ass, price = oracle_mod.auction_forward(a, c, 0.01, 3000)
    assert len(np.unique(ass[0])) == 1024  # eps-optimal complete assignment
    from scipy.optimize import linear_sum_assignment

    cost = np.sqrt(((a[0][:, None, :] - c[0][None, :, :]) ** 2).sum(-1))
    r, col = linear_sum_assignment(cost)
    assert np.sqrt(dist[0]).sum() <= cost[r, col].sum() + 1024 * 0.01 + 1e-3  # auction bound: within n*eps


def test_oracle_auction_rejects_bad_sizes(oracle_mod):
    a, c = pair(0, 1, 100, 100, 'uniform')
    with pytest.raises(ValueError):
        oracle_mod.auction_forward(a, c, 0.005, 10)


def test_oracle_auction_backward(oracle_mod):
    a, c = _clouds(3, 1, 1024)
    dist, ass, _ = oracle_mod.auction_forward(a, c, 0.005, 20)
    g = np.random.default_rng(0).standard_normal((1, 1024)).astype(np.float32)
    out = oracle_mod.auction_backward(a, c, g, ass)
    exp = 2 * g[..., None] * (a - c[0][ass[0]][None])
    np.testing.assert_allclose(out, exp, rtol=1e-6, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize('b,n,eps,iters', [(2, 1024, 0.005, 50), (3, 2048, 0.005, 50), (1, 1024, 0.002, 300),
                                            (2, 1024, 0.01, 1), (1, 5120, 0.005, 20)])
def test_auction_gpu_matches_oracle(cuda, oracle_mod, b, n, eps, iters):
    from emd import emdModule

    a, c = _clouds(10 + n + iters, b, n)
    t1 = torch.from_numpy(a).to(cuda).requires_grad_(True)
    t2 = torch.from_numpy(c).to(cuda)
    dist, ass = emdModule()(t1, t2, eps, iters)
    od, oa, _ = oracle_mod.auction_forward(a, c, eps, iters)
    assert np.array_equal(ass.cpu().numpy(), oa)
    assert np.array_equal(dist.detach().cpu().numpy(), od)
    g = torch.randn(b, n, generator=torch.Generator().manual_seed(0))
    dist.backward(g.to(cuda))
    np.testing.assert_allclose(t1.grad.cpu().numpy(), oracle_mod.auction_backward(a, c, g.numpy(), oa), rtol=1e-6, atol=1e-7)


@pytest.mark.gpu
def test_auction_at_baseline_config2_workload(cuda, oracle_mod):
    """BASELINE configs[2], auction reading, at its full workload: B=32, n=2048, eps=0.005, iters=50
    (external/emd/README.md:7's training parameters), stress-set clouds (U[0,1]^3, seed 1234 + config id).
    Forward (emd_cuda.cu:227-281): assignment and dist bit-exact against the deterministic oracle; backward
    (emd_cuda.cu:283-315): grad_xyz1 = 2 g (p1 - p2[assignment]), grad_xyz2 = 0 (emd_module.py:76-79)."""
    from emd import emdModule

    a, c = pair(1234 + 2, 32, 2048, 2048, 'uniform')
    t1 = torch.from_numpy(a).to(cuda).requires_grad_(True)
    t2 = torch.from_numpy(c).to(cuda).requires_grad_(True)
    dist, ass = emdModule()(t1, t2, 0.005, 50)
    od, oa, _ = oracle_mod.auction_forward(a, c, 0.005, 50)
    assert ass.dtype == torch.int32 and tuple(ass.shape) == (32, 2048)
    assert np.array_equal(ass.cpu().numpy(), oa)
    assert np.array_equal(dist.detach().cpu().numpy().view(np.uint32), od.view(np.uint32))
    assert oa.min() >= 0 and oa.max() < 2048
    g = torch.randn(32, 2048, generator=torch.Generator().manual_seed(3))
    dist.backward(g.to(cuda))
    np.testing.assert_allclose(t1.grad.cpu().numpy(), oracle_mod.auction_backward(a, c, g.numpy(), oa), rtol=1e-6, atol=1e-7)
    assert t2.grad is None or float(t2.grad.abs().max()) == 0.0


@pytest.mark.gpu
def test_auction_many_samples_chunked_launches(cuda, oracle_mod):
    """More samples than one co-resident launch of 8-workgroup clusters holds (256 CUs / 8): consecutive launches."""
    from emd import emdModule

    a, c = _clouds(77, 40, 1024)
    dist, ass = emdModule()(torch.from_numpy(a).to(cuda), torch.from_numpy(c).to(cuda), 0.005, 10)
    od, oa, _ = oracle_mod.auction_forward(a, c, 0.005, 10)
    assert np.array_equal(ass.cpu().numpy(), oa) and np.array_equal(dist.cpu().numpy(), od)


@pytest.mark.gpu
@pytest.mark.parametrize('cluster', [1, 4, 16])
def test_auction_cluster_sizes_agree(cuda, oracle_mod, cluster):
    """One workgroup per sample (the schedule for batches that fill the chip on their own) and other cluster sizes
    (measurement switch `auction_cluster`, include/pcc_test_hooks.h) give the oracle's bits too."""
    from emd import emdModule
    from pointcloudcounterfactual_amd import _lib

    a, c = _clouds(5, 3, 2048)
    od, oa, _ = oracle_mod.auction_forward(a, c, 0.005, 30)
    _lib.set_tuning('auction_cluster', cluster)
    try:
        d, asg = emdModule()(torch.from_numpy(a).to(cuda), torch.from_numpy(c).to(cuda), 0.005, 30)
        torch.cuda.synchronize()
    finally:
        _lib.set_tuning('auction_cluster', 0)
    assert np.array_equal(asg.cpu().numpy(), oa) and np.array_equal(d.cpu().numpy(), od)


@pytest.mark.gpu
def test_auction_readme_relation(cuda):
    """external/README.md:20-41: sqrt(auction).mean(1) and match_cost/N are the same quantity, loosely."""
    from emd import emdModule
    from structural_losses import match_cost

    a, c = _clouds(42, 2, 2048)
    t1, t2 = torch.from_numpy(a).to(cuda), torch.from_numpy(c).to(cuda)
    emd1 = torch.sqrt(emdModule()(t1, t2, 0.01, 200)[0]).mean(1)
    emd2 = match_cost(t1, t2) / t1.shape[1]
    # two different approximations of the same transport cost: the auction is within n*eps of the optimum, the
    # multi-scale soft matching over-estimates it; the README prints them side by side without a tolerance.
    ratio = emd2 / emd1
    assert ((ratio > 0.9) & (ratio < 2.0)).all(), (emd1, emd2)


def test_emd_module_input_checks():
    from emd import emdModule

    with pytest.raises(ValueError, match='multiple of 1024'):
        emdModule()(torch.zeros(1, 100, 3), torch.zeros(1, 100, 3), 0.005, 10)
    with pytest.raises(ValueError, match='same number of points'):
        emdModule()(torch.zeros(1, 1024, 3), torch.zeros(1, 2048, 3), 0.005, 10)


@pytest.mark.gpu
def test_auction_failure_is_reported_not_silent(cuda):
    """A cluster launch whose sample barrier fails (here: injected through the test hook, which raises the kernel's
    error word before the launch) poisons its outputs AND surfaces as an error: the next call on the device raises
    instead of returning rc 0, the backward pass of a poisoned assignment (-1) reads nothing out of range and gives a
    zero gradient, and the device recovers afterwards."""
    from emd import emd_backend
    from pointcloudcounterfactual_amd import _lib

    a, c = _clouds(7, 4, 2048)
    t1, t2 = torch.from_numpy(a).to(cuda), torch.from_numpy(c).to(cuda)
    dist = torch.zeros(4, 2048, device=cuda)
    ass = torch.zeros(4, 2048, device=cuda, dtype=torch.int32)
    assert _lib.lib.pcc_test_inject_auction_failure() == 1  # armed: tests/conftest.py sets PCC_TEST_HOOKS=1
    assert emd_backend.forward(t1, t2, dist, ass, eps=0.005, iters=20) == 1  # the launch itself is asynchronous
    torch.cuda.synchronize()
    assert torch.isnan(dist).all() and (ass == -1).all()
    grad = torch.full((4, 2048, 3), 7.0, device=cuda)
    with pytest.raises(RuntimeError, match='did not complete'):
        emd_backend.backward(t1, t2, grad, torch.ones(4, 2048, device=cuda), ass)
    # the word is cleared by the report: the same backward now runs and an unassigned point gets a zero gradient
    emd_backend.backward(t1, t2, grad, torch.ones(4, 2048, device=cuda), ass)
    torch.cuda.synchronize()
    assert float(grad.abs().max()) == 0.0
    assert _lib.lib.pcc_auction_status() == 0
    emd_backend.forward(t1, t2, dist, ass, eps=0.005, iters=20)
    torch.cuda.synchronize()
    assert torch.isfinite(dist).all() and (ass >= 0).all()
    # and the other way round: a failed forward followed by a forward
    _lib.lib.pcc_test_inject_auction_failure()
    emd_backend.forward(t1, t2, dist, ass, eps=0.005, iters=20)
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match='did not complete'):
        emd_backend.forward(t1, t2, dist, ass, eps=0.005, iters=20)
    emd_backend.forward(t1, t2, dist, ass, eps=0.005, iters=20)
    torch.cuda.synchronize()
    assert torch.isfinite(dist).all()


@pytest.mark.gpu
def test_auction_cluster_launches_on_two_streams(cuda):
    """Two cluster launches enqueued on different streams are ordered by the library (each needs all its workgroups
    resident): both finish with the bits of a launch on its own."""
    from emd import emdModule

    a, c = _clouds(9, 8, 2048)
    t1, t2 = torch.from_numpy(a).to(cuda), torch.from_numpy(c).to(cuda)
    ref_d, ref_a = emdModule()(t1, t2, 0.005, 30)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for st in (s1, s2, s1, s2):
        with torch.cuda.stream(st):
            outs.append(emdModule()(t1, t2, 0.005, 30))
    torch.cuda.synchronize()
    for d, asg in outs:
        assert torch.equal(d, ref_d) and torch.equal(asg, ref_a)
