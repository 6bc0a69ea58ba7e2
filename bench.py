#!/usr/bin/env python3
"""Headline benchmark: clouds/sec, Chamfer + approximate-EMD forward+backward, N=2048, B=32 per GPU.

``python bench.py --gpus N --steps K --warmup W``.  One *step* = one pass of the structural-loss hot path over one
batch of B=32 synthetic cloud pairs already resident in HBM:

    loss = chamfer(recon, ref) + match_cost(recon, ref);  loss.sum().backward()

through the autograd surface, i.e. the HIP kernels behind the C ABI.  By default the two losses are taken as ONE
autograd node (``chamfer_emd``: the reference's ChamferEMD reconstruction loss, metrics_and_losses.py:70-79; same
kernels, same bits, the nearest-neighbour search scheduled in the shadow of the approximate-EMD launch chain);
``--separate`` calls ``chamfer`` and ``match_cost`` one after the other as round 1 did.

The batch shards trivially over ranks (weak scaling, no data-path collective).  Under ``torch.distributed.run`` (RANK /
WORLD_SIZE in the environment) the process is one rank; without it ``--gpus N`` (N > 1, or ``--via-launcher``) starts
its own N ranks -- ``python -m torch.distributed.run --nproc-per-node N bench.py ...`` as a CHILD process of a parent
that never touches the GPU (reference: ``src/utils/parallel.py:37-53`` spawns its ranks itself) -- and relays rank
0's line.  Rank 0 prints ONE JSON line (contract in the task statement; the ``roofline`` / ``cpu_baseline`` /
``parity_at_bench_size`` objects are described in DESIGN.md section 5).
"""

from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

B_PER_GPU = 32
N_POINTS = 2048
SEED = 1234 + 2  # seed = 1234 + config id

# Algorithmic work per pair-evaluation of one reference pass of approxmatch:
# 8 flop for the squared distance + level multiply + weight multiply + accumulate + 1 for the exp
# = 12 flop + 1 exp, counted as 13.
FLOP_PER_PAIR_PASS = 13.0
PHASE_LAUNCHES = 19        # am_phase_kernel launches per approxmatch (27 reference passes, 8 of them fused pairwise)
PEAK_LANE_OPS = 256 * 4 * 32 * 2.4e9  # vector lanes x clock (an FMA lane-op = 2 of the 157.3 TFLOP/s)
PEAK_F32_TFLOPS = 157.3   # MI355X_MICROARCH.md: FP32 vector peak == FP32 dense MFMA peak
PEAK_HBM_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (6.29 TB/s measured copy)
CHAMFER_ALGO_BYTES = 6_815_744  # fwd 2,621,440 + bwd 4,194,304 at B=32, N=M=2048
TRANS_SLOTS = 4.0         # a transcendental (v_exp_f32 / v_rsq_f32) occupies the vector pipe for 4 issue slots

SEPARATE = False


def parse() -> argparse.Namespace:
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--reps', type=int, default=7,
                    help='repetitions of the K timed steps; value = the median repetition, value_min/max beside it')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the timings of the rows outside the headline metric')
    ap.add_argument('--cpu-clouds', type=int, default=0, help='clouds in the CPU-baseline sample (0 = auto)')
    ap.add_argument('--separate', action='store_true',
                    help='chamfer() and match_cost() as two autograd nodes, one after the other (the round-1 step)')
    ap.add_argument('--via-launcher', action='store_true',
                    help='start the ranks through torch.distributed.run even for --gpus 1 (exercises RCCL at world 1)')
    ap.add_argument('--phase-probe', action='store_true', help=argparse.SUPPRESS)  # child of the no-skip measurement
    ap.add_argument('--no-dist-init', action='store_true', help=argparse.SUPPRESS)  # diagnosis: launcher without RCCL
    ap.add_argument('--no-noskip', action='store_true', help='skip the child process of the no-skip roofline (profiling runs)')
    ap.add_argument('--kind', choices=['recon', 'uniform'], default='recon', help=argparse.SUPPRESS)
    ap.add_argument('--emd-mode', choices=['implicit', 'fused', 'reference'], default='implicit',
                    help="how match_cost carries out ApproxMatch -> MatchCost / MatchCostGrad (losses.MatchCostFunction.mode): "
                         "'implicit' never stores match; 'fused' / 'reference' materialise the [B,M,N] tensor (with --separate)")
    return ap.parse_args()


def self_launch(args: argparse.Namespace, script: str | None = None) -> int:
    """Start the N ranks of this benchmark (or of ``script``: bench_train.py shares the launcher) as a child
    ``torch.distributed.run`` (this process has not initialised HIP and never does: it only relays the child's output and
    exit code)."""
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    argv = [a for a in sys.argv[1:] if a != '--via-launcher']
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(script or __file__)] + argv
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    proc = subprocess.run(cmd, env=env)
    return proc.returncode


def make_inputs(rank: int, dev, kind: str = 'recon'):
    import torch

    from tests.util import pair

    recon, ref = pair(SEED + 1000 * rank, B_PER_GPU, N_POINTS, N_POINTS, kind)
    return recon, ref, torch.from_numpy(recon).to(dev), torch.from_numpy(ref).to(dev)


def step(recon_t, ref_t):
    from pointcloudcounterfactual_amd import chamfer, chamfer_emd, match_cost

    recon_t.grad = None
    if SEPARATE:
        loss = chamfer(recon_t, ref_t) + match_cost(recon_t, ref_t)
    else:
        lc, le = chamfer_emd(recon_t, ref_t)
        loss = lc + le
    loss.sum().backward()
    return loss


def timed_steps(recon_t, ref_t, steps: int, warmup: int) -> float:
    """Seconds per step on this rank (warm-up, synchronise, K steps, synchronise)."""
    import torch

    for _ in range(warmup):
        step(recon_t, ref_t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(recon_t, ref_t)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def kernel_breakdown(recon_t, ref_t, steps: int) -> dict[str, float]:
    """Average duration (us) of each C-ABI launch sequence: HIP events on torch's current stream (the stream every kernel
    of this library is enqueued on) around `steps` back-to-back calls of the SAME entry point, so that the host runs
    ahead of the GPU as it does in the timed loop.  (Rounds 1-2 bracketed single calls of different entry points one after
    the other: after a 50 us Chamfer call the GPU idles while the host enqueues the ~45 launches of an EMD call, and the
    interval measured the host -- match_cost read 40 us above its steady-state time.)"""
    import torch

    from pointcloudcounterfactual_amd import backend

    b, n = recon_t.shape[0], recon_t.shape[1]
    g = torch.full((b, n), 1.0 / n, device=recon_t.device)
    d1, i1, d2, i2 = backend.NNDistance(recon_t, ref_t)
    match = backend.ApproxMatchCost(recon_t, ref_t)[0]

    def ev(fn) -> float:
        fn()
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(steps):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / steps * 1e3

    out = {
        'nndistance': ev(lambda: backend.NNDistance(recon_t, ref_t)),
        'nndistancegrad': ev(lambda: backend.NNDistanceGrad(recon_t, ref_t, i1, i2, g, g)),
        'match_cost_implicit_fwd_bwd': ev(lambda: backend.MatchCostImplicit(recon_t, ref_t, True)),
        'match_cost_implicit_fwd_only': ev(lambda: backend.MatchCostImplicit(recon_t, ref_t, False)),
        'matchcostgrad': ev(lambda: backend.MatchCostGrad(recon_t, ref_t, match)),
    }
    del match
    out['approxmatch_cost'] = ev(lambda: backend.ApproxMatchCost(recon_t, ref_t))
    return out


def phase_kernel_time_us(recon_t, ref_t, steps: int) -> tuple[float, int, int]:
    """Average duration of ONE am_phase_kernel launch (the dominant kernel), measured live with HIP events on
    the launch stream: the library brackets the 19 back-to-back phase launches of every approxmatch with one event
    before the first and one after the last (pcc_profile_enable(2); no event between the kernels, so nothing but
    the kernels themselves is in the interval) and the interval is divided by 19.  A large batch runs as two
    half-batch sequences on two streams at the same time: both are bracketed, each on its own stream.
    Returns (us per launch, launches timed, concurrent sequences per call)."""
    import ctypes

    import torch

    from pointcloudcounterfactual_amd import _lib, backend

    L = _lib.lib
    L.pcc_profile_enable(2)
    for _ in range(steps):
        backend.MatchCostImplicit(recon_t, ref_t, True)
    torch.cuda.synchronize()
    us = ctypes.c_double(0)
    cnt = ctypes.c_int(0)
    L.pcc_profile_read(b'am_phase_sequence', ctypes.byref(us), ctypes.byref(cnt))
    L.pcc_profile_enable(0)
    return us.value / PHASE_LAUNCHES, cnt.value * PHASE_LAUNCHES, max(1, cnt.value // steps)


def phase_probe(args: argparse.Namespace) -> int:
    """Child process of the no-skip measurement: every exact-zero skip off (measurement switch `am_nocull` of
    include/pcc_test_hooks.h, armed by PCC_TEST_HOOKS=1 in this child's environment: the kernels execute the
    reference's full arithmetic); its own process, one JSON line."""
    import torch

    from pointcloudcounterfactual_amd import _lib, backend

    _lib.set_tuning('am_nocull', 1)

    dev = torch.device('cuda', 0)
    _, _, recon_t, ref_t = make_inputs(0, dev, args.kind)
    for _ in range(3):
        backend.MatchCostImplicit(recon_t, ref_t, True)
    torch.cuda.synchronize()
    k = max(3, min(args.steps, 20))
    us, cnt, lanes = phase_kernel_time_us(recon_t, ref_t, k)
    s = torch.cuda.Event(enable_timing=True)
    e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(k):
        backend.MatchCostImplicit(recon_t, ref_t, True)
    e.record()
    torch.cuda.synchronize()
    print(json.dumps({'phase_us': us, 'launches': cnt, 'lanes': lanes, 'emd_fwd_bwd_us': s.elapsed_time(e) / k * 1e3}))
    return 0


def noskip_roofline(steps: int) -> dict:
    """am_phase_kernel with every exact-zero skip switched off (child process with the `am_nocull` switch): the kernels then
    execute exactly the reference's arithmetic, algorithmic == executed, and algorithmic flops / time / peak is a true
    fraction of the vector rate."""
    env = dict(os.environ, PCC_TEST_HOOKS='1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), '--phase-probe', '--steps', str(steps)], env=env,
                       capture_output=True, text=True, timeout=600)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    if r.returncode != 0 or not line:
        return {'error': (r.stderr or r.stdout)[-300:]}
    p = json.loads(line[-1])
    pairs = B_PER_GPU * N_POINTS * N_POINTS
    flop_per_launch = 27.0 / PHASE_LAUNCHES * (pairs / p['lanes']) * FLOP_PER_PAIR_PASS
    achieved = p['lanes'] * flop_per_launch / (p['phase_us'] * 1e-6) / 1e12
    return {'avg_launch_us': p['phase_us'], 'concurrent_launches': p['lanes'], 'achieved': achieved, 'peak': PEAK_F32_TFLOPS,
            'unit': 'TFLOP/s', 'frac': achieved / PEAK_F32_TFLOPS, 'emd_fwd_bwd_us': p['emd_fwd_bwd_us'],
            'note': 'measurement switch am_nocull: no term is skipped, the kernels execute the 27 reference passes in full '
                    '(13 flop-equivalents per pair per pass), so this fraction is algorithmic AND executed'}


def other_rows_us(dev) -> dict[str, float]:
    """Durations (us, HIP events) of the other SURVEY.md 8(a) rows at the same batch: auction EMD (A12-A13, eps 0.005,
    50 iterations) and the encoder primitives (A14-A17) at the DGCNN layer shapes.  Informational: not part of `value`."""
    import torch

    from emd import emdModule
    from pointcloudcounterfactual_amd import neighbour_ops as ops

    def ev(fn, iters=12, warm=3) -> float:
        for _ in range(warm):
            fn()
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / iters * 1e3

    g = torch.Generator().manual_seed(SEED)
    out: dict[str, float] = {}
    a = torch.rand(B_PER_GPU, N_POINTS, 3, generator=g).to(dev).requires_grad_(True)
    b = torch.rand(B_PER_GPU, N_POINTS, 3, generator=g).to(dev)

    def auction():
        a.grad = None
        emdModule()(a, b, 0.005, 50)[0].sum().backward()

    out['auction_emd_fwd_bwd_eps0.005_iters50'] = ev(auction, iters=3, warm=1)
    # k-NN on xyz: the bench's own reference clouds (what the encoder's first layer sees: the sorted search ends early on
    # them) and, as `_gaussian`, an i.i.d. Gaussian volume, where a 25-neighbour ball still touches most 16-point boxes
    cloud = make_inputs(0, dev)[3].transpose(1, 2).contiguous()
    for c, k in ((3, 25), (64, 25), (128, 25), (3, 4)):
        x = torch.randn(B_PER_GPU, c, N_POINTS, generator=g).to(dev)
        if c == 3:
            out[f'knn_c{c}_k{k}'] = ev(lambda: ops.hip_knn(cloud, k))
            out[f'knn_c{c}_k{k}_gaussian'] = ev(lambda: ops.hip_knn(x, k))
        else:
            out[f'knn_c{c}_k{k}'] = ev(lambda: ops.hip_knn(x, k))
    x = torch.randn(B_PER_GPU, 64, N_POINTS, generator=g).to(dev)
    idx = ops.hip_knn(x, 25)
    out['graph_features_c64_k25_fwd'] = ev(lambda: ops.get_graph_features(x, idx, 25))
    out['graph_max_pooling_c64_k25_fwd'] = ev(lambda: ops.graph_max_pooling(x, idx, 25))
    # backward of the two gathers on a gradient tensor already resident in HBM (the C call alone: autograd's
    # `.sum().backward()` would first materialise its expanded scalar -- an 838 MB copy that is not this kernel's)
    from pointcloudcounterfactual_amd import _lib

    L, st = _lib.lib, torch.cuda.current_stream().cuda_stream
    gx = torch.empty(B_PER_GPU, 64, N_POINTS, device=dev)
    g2 = torch.randn(B_PER_GPU, 128, N_POINTS, 25, device=dev)
    out['graph_features_c64_k25_bwd'] = ev(lambda: L.pcc_graph_features_bwd(B_PER_GPU, 64, N_POINTS, 25, idx.data_ptr(), g2.data_ptr(),
                                                                           gx.data_ptr(), st))
    g1 = g2[:, :64].contiguous()
    del g2
    out['get_neighbours_c64_k25_bwd'] = ev(lambda: L.pcc_gather_neighbours_bwd(B_PER_GPU, 64, N_POINTS, 25, idx.data_ptr(), g1.data_ptr(),
                                                                            gx.data_ptr(), st))
    del g1
    x2 = torch.randn(B_PER_GPU, 1024, N_POINTS, generator=g).to(dev)
    out['global_max_pool_c1024'] = ev(lambda: ops.global_max_pool(x2))
    out['global_max_pool_c1024_torch'] = ev(lambda: x2.max(dim=2))
    return out


def cpu_baseline_and_parity(recon, ref, recon_t, ref_t, clouds: int) -> tuple[dict, dict]:
    """(cpu_baseline, parity_at_bench_size).

    CPU baseline, timed on this host for the same step on a bounded sample of the bench batch:
    Chamfer = the reference's CPU path restated (expanded-form torch ops of src/utils/neighbour_ops.py:43-50 and the
    mean-form loss, with autograd backward); approximate EMD = the oracle (C restatement of approxmatch.cu, OpenMP over
    the batch) -- the reference has NO CPU EMD (metrics_and_losses.py:76-79 drops it on CPU).  kind = "port".

    Parity at the bench size: the oracle outputs of that sample are kept and compared with what the HIP path returns
    for the same clouds."""
    import numpy as np
    import torch

    import oracle
    from pointcloudcounterfactual_amd import backend
    from pointcloudcounterfactual_amd.losses import torch_square_distance

    threads = max(1, min(oracle.max_threads(), os.cpu_count() or 1, clouds))
    oracle.set_threads(threads)
    a, c = recon[:clouds], ref[:clouds]
    n = a.shape[1]
    # reference CPU Chamfer (A5), mean form, forward + autograd backward
    torch.set_num_threads(max(1, min(os.cpu_count() or 1, 32)))
    t_threads = torch.get_num_threads()
    ta = torch.from_numpy(a).requires_grad_(True)
    tc = torch.from_numpy(c)
    t0 = time.perf_counter()
    dist = torch_square_distance(ta, tc)
    loss = torch.min(dist, dim=-1)[0].mean(1) + torch.min(dist, dim=-2)[0].mean(1)
    loss.sum().backward()
    t_torch = time.perf_counter() - t0
    del dist
    # oracle: Chamfer (kept for the parity check) and approximate EMD (the reference's data flow: match materialised)
    t0 = time.perf_counter()
    d1, i1, d2, i2 = oracle.nndistance(a, c)
    g = np.full_like(d1, 1.0 / n)
    og1, og2 = oracle.nndistancegrad(a, c, i1, i2, g, g)
    t1 = time.perf_counter()
    match, _ = oracle.approxmatch(a, c)
    ocost = oracle.matchcost(a, c, match)
    om1, om2 = oracle.matchcostgrad(a, c, match)
    t2 = time.perf_counter()
    del match
    cpu = {
        'value': clouds / (t_torch + (t2 - t1)),
        'unit': 'clouds/s',
        'cores': max(threads, t_threads),
        'kind': 'port',
        'sample': f'{clouds} of the {B_PER_GPU} clouds of the bench batch (N={n}): Chamfer fwd+bwd = the reference\'s CPU path '
                  f'restated (torch expanded form + autograd, {t_threads} threads) {t_torch:.3f}s; approx-EMD fwd+bwd = oracle C '
                  f'restatement, OpenMP over the batch ({threads} threads) {t2 - t1:.3f}s; [oracle Chamfer fwd+bwd {t1 - t0:.3f}s]',
    }
    # ---- parity of the node that is TIMED: backend.ChamferEMD (pcc_chamfer_emd: nn_sorted_kernel + the approximate-EMD
    # chain + am_pair_kernel) and backend.ChamferEMDGrad (pcc_chamfer_emd_grad), i.e. exactly what chamfer_emd() calls
    with torch.no_grad():
        x1, x2 = recon_t[:clouds].contiguous(), ref_t[:clouds].contiguous()
        lc, j1, j2, cost, e1, e2, h1, h2 = backend.ChamferEMD(x1, x2, True, True, return_dist=True)
        ones = torch.ones((clouds,), device=x1.device)
        zeros = torch.zeros((clouds,), device=x1.device)
        # the two upstream scalars separately, so that each term of the one backward launch is checked on its own
        r1, r2 = backend.ChamferEMDGrad(x1, x2, j1, j2, ones, True, e1, e2, zeros)
        s1, s2 = backend.ChamferEMDGrad(x1, x2, j1, j2, zeros, True, e1, e2, ones)
    # checker for the approximate EMD: the float64 recurrence, and the spread of the oracle's own legitimate float32
    # evaluations around it (exp modes 0-3) -- the bars of test_parity_at_bench_point_size
    om64, _ = oracle.approxmatch_f64(a, c)
    oc64 = oracle.matchcost_f64(a, c, om64)
    f64_1, f64_2 = oracle.matchcostgrad_f64(a, c, om64)
    del om64
    scale = float(max(np.abs(f64_1).max(), np.abs(f64_2).max()))
    spread = float(max(np.abs(om1 - f64_1).max(), np.abs(om2 - f64_2).max()))
    try:
        for mode in (1, 2, 3):
            oracle.set_exp_mode(mode)
            mm, _ = oracle.approxmatch(a, c)
            q1, q2 = oracle.matchcostgrad(a, c, mm)
            del mm
            spread = max(spread, float(np.abs(q1 - f64_1).max()), float(np.abs(q2 - f64_2).max()))
    finally:
        oracle.set_exp_mode(0)
    rel = lambda got, exp: float(np.abs(got - exp).max() / max(np.abs(exp).max(), 1e-30))  # noqa: E731
    chamfer_loss_oracle = d1.astype(np.float64).mean(1) + d2.astype(np.float64).mean(1)
    emd_grad_err = float(max(np.abs(s1.cpu().numpy() - f64_1).max(), np.abs(s2.cpu().numpy() - f64_2).max()))
    emd_grad_bar = max(3.0 * spread, 1e-5 * scale)
    parity = {
        'clouds': clouds, 'n_points': n,
        'node': 'backend.ChamferEMD(return_dist=True) + backend.ChamferEMDGrad: the calls chamfer_emd() -- the timed step -- makes',
        'checker': 'oracle on the same bits: float32 C restatement of nndistance.cu (indices, distances, Chamfer gradients); '
                   'float64 recurrence of approxmatch.cu / matchcost (EMD cost and gradients)',
        'nn_idx_mismatches': int((j1.cpu().numpy() != i1).sum() + (j2.cpu().numpy() != i2).sum()),
        'nn_dist_bit_mismatches': int((h1.cpu().numpy().view(np.uint32) != d1.view(np.uint32)).sum()
                                      + (h2.cpu().numpy().view(np.uint32) != d2.view(np.uint32)).sum()),
        'chamfer_loss_max_rel_err': float(np.abs(lc.cpu().numpy() - chamfer_loss_oracle).max() / np.abs(chamfer_loss_oracle).max()),
        'chamfer_grad_max_err_rel_to_largest': max(rel(r1.cpu().numpy(), og1), rel(r2.cpu().numpy(), og2)),
        'chamfer_grad_vs_reference_torch_cpu_path_rel_to_largest': rel(r1.cpu().numpy(), ta.grad.numpy()),
        'emd_cost_max_rel_err_vs_f64': float((np.abs(cost.cpu().numpy() - oc64) / np.abs(oc64)).max()),
        'emd_cost_max_rel_err_vs_f32_oracle': float((np.abs(cost.cpu().numpy() - ocost) / np.abs(ocost)).max()),
        'emd_grad_max_abs_err_vs_f64': emd_grad_err,
        'emd_grad_bar': emd_grad_bar,
        'emd_grad_f32_oracle_spread_vs_f64': spread,
        'emd_grad_largest_component': scale,
    }
    parity['bars'] = {
        'nn_idx_mismatches': 0, 'nn_dist_bit_mismatches': 0, 'chamfer_loss_max_rel_err': 1e-5,
        'chamfer_grad_max_err_rel_to_largest': 1e-5, 'emd_cost_max_rel_err_vs_f64': 1e-5,
        'emd_grad_max_abs_err_vs_f64': 'max(3 x spread of the four legitimate float32 evaluations of the oracle, 1e-5 of the '
                                       'largest component) -- tests/test_gpu_reference_pins.py::test_parity_at_bench_point_size',
    }
    parity['bars_ok'] = bool(
        parity['nn_idx_mismatches'] == 0 and parity['nn_dist_bit_mismatches'] == 0
        and parity['chamfer_loss_max_rel_err'] <= 1e-5 and parity['chamfer_grad_max_err_rel_to_largest'] <= 1e-5
        and parity['emd_cost_max_rel_err_vs_f64'] <= 1e-5 and emd_grad_err <= emd_grad_bar)
    return cpu, parity


def main() -> int:
    global SEPARATE
    args = parse()
    if args.phase_probe:
        return phase_probe(args)
    launched = 'RANK' in os.environ and 'WORLD_SIZE' in os.environ
    if not launched and (args.gpus > 1 or args.via_launcher):
        return self_launch(args)

    import torch

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a line for another job size')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the HIP path has no CPU fallback)')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist = None
    if launched and not args.no_dist_init:  # one process per GPU over RCCL (backend "nccl" on ROCm), also at world size 1
        import torch.distributed as dist_mod

        dist = dist_mod
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=dev)

    import pointcloudcounterfactual_amd  # noqa: F401  (raises if the HIP library is missing)
    from pointcloudcounterfactual_amd.losses import MatchCostFunction

    SEPARATE = bool(args.separate) or args.emd_mode != 'implicit'
    MatchCostFunction.mode = args.emd_mode
    recon, ref, recon_t, ref_t = make_inputs(rank, dev)
    recon_t.requires_grad_(True)

    def sync() -> None:
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(recon_t, ref_t)
    # R repetitions of the K timed steps, each bracketed by barrier + synchronize on both sides and reduced with MAX
    # over the ranks; `value` is the MEDIAN repetition (K=20 steps are a 10 ms window: one repetition is one sample
    # of the clock, the spread is reported as value_min / value_max)
    rep_s: list[float] = []
    for _ in range(max(1, args.reps)):
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(recon_t, ref_t)
        sync()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        rep_s.append(el)
    rep_sorted = sorted(rep_s)
    elapsed = rep_sorted[len(rep_sorted) // 2] if len(rep_sorted) % 2 else 0.5 * (rep_sorted[len(rep_sorted) // 2 - 1]
                                                                                 + rep_sorted[len(rep_sorted) // 2])

    clouds = B_PER_GPU * world * args.steps
    result = {
        'metric': 'clouds/sec Chamfer+EMD fwd+bwd, N=2048 B=32',
        'value': clouds / elapsed,
        'unit': 'clouds/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': elapsed / args.steps * 1e3,
        'repetitions': len(rep_s),
        'value_min': B_PER_GPU * world * args.steps / max(rep_s),
        'value_max': B_PER_GPU * world * args.steps / min(rep_s),
        'ms_per_step_all_repetitions': [r / args.steps * 1e3 for r in rep_s],
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': 'f32',
        'data': 'synthetic',
        'config': {
            'workload': 'BASELINE configs[1]+[2]: N=2048 B=32 per GPU, nn_distance (Chamfer, mean) fwd+bwd + '
                        'match_cost (approxmatch+matchcost) fwd+bwd through the autograd surface; inputs = "recon" clouds '
                        '(reference permuted + N(0, 0.02^2): a trained autoencoder, SURVEY 8(d)); the 8(d) stress set '
                        '(both clouds i.i.d. U[0,1]^3) is value_stress',
            'step': 'separate autograd nodes chamfer() + match_cost()' if SEPARATE else
                    'one autograd node chamfer_emd() (the reference\'s ChamferEMD loss; same kernels and bits as the two '
                    'separate nodes, nearest-neighbour search overlapped with the EMD launch chain)',
            'emd_mode': args.emd_mode,
            'batch_per_gpu': B_PER_GPU,
            'n_points': N_POINTS,
            'global_batch': B_PER_GPU * world,
            'parallelism': f'batch-sharded x{world}, no data-path collective',
            'launcher': 'torch.distributed.run + RCCL' if launched else 'single process',
        },
    }

    # stress set of SURVEY.md 8(d) (both clouds uniform in the unit cube: an untrained autoencoder): same step, same K
    _, _, su1, su2 = make_inputs(rank, dev, 'uniform')
    su1.requires_grad_(True)
    k_stress = max(3, min(args.steps, 20))
    sps = timed_steps(su1, su2, k_stress, 2)
    if dist is not None:
        t = torch.tensor([sps], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        sps = float(t.item())
    result['value_stress'] = B_PER_GPU * world / sps
    result['ms_per_step_stress'] = sps * 1e3
    del su1, su2

    if rank == 0:
        k2 = max(3, min(args.steps, 20))
        try:  # the other ways of taking the same step (informational: never lose the headline line)
            keep = SEPARATE
            alt: dict[str, object] = {}
            SEPARATE = not keep
            alt['separate_nodes' if keep is False else 'one_node'] = B_PER_GPU / timed_steps(recon_t, ref_t, k2, 2)
            if args.emd_mode == 'implicit':
                SEPARATE = True
                MatchCostFunction.mode = 'fused'
                alt['separate_nodes_match_materialised'] = B_PER_GPU / timed_steps(recon_t, ref_t, k2, 2)
            alt['unit'] = 'clouds/s on this rank'
            alt['note'] = ("'separate_nodes': chamfer() then match_cost() (the round-1 step); '..._match_materialised': "
                           "match_cost mode 'fused', match[B,M,N] written once and read once (the reference's data flow)")
            result['other_step_forms'] = alt
        except Exception as e:
            result['other_step_forms'] = {'error': repr(e)}
        finally:
            SEPARATE = bool(args.separate) or args.emd_mode != 'implicit'
            MatchCostFunction.mode = args.emd_mode
        with torch.no_grad():
            br = kernel_breakdown(recon_t.detach(), ref_t, k2)
            phase_us, phase_cnt, lanes = phase_kernel_time_us(recon_t.detach(), ref_t, k2)
        pairs = B_PER_GPU * N_POINTS * N_POINTS
        # The dominant kernel: am_phase_kernel (19 launches per approxmatch).  `lanes` launches (one per half batch, on
        # two streams) run at the same time: the rate of the chip is the work of all of them over the duration of one.
        ok = phase_us == phase_us and phase_us > 0
        algo_flop_per_launch = 27.0 / PHASE_LAUNCHES * (pairs / lanes) * FLOP_PER_PAIR_PASS
        algorithmic = lanes * algo_flop_per_launch / (phase_us * 1e-6) / 1e12 if ok else None
        traffic = executed = None
        pmc = os.path.join(ROOT, 'profiles', 'pmc_summary.json')
        if os.path.exists(pmc) and ok:
            try:
                fam = json.load(open(pmc)).get('am_phase+am_fine', {})
                traffic = fam.get('hbm_bytes_per_launch')
                insts = fam.get('valu_insts_per_launch')
                trans = fam.get('trans_insts_per_launch')
                if insts:
                    # what the kernel EXECUTES (committed PMC pass of the same command: wave-level VALU instructions per
                    # launch x 64 lanes; a transcendental occupies the pipe for 4 issue slots), over the live launch duration
                    slots = (insts - trans) + TRANS_SLOTS * trans if trans else insts
                    rate = lanes * slots * 64 / (phase_us * 1e-6)
                    executed = {'issue_slots_per_launch': slots * 64, 'valu_insts_per_launch': insts,
                                'trans_insts_per_launch': trans, 'lane_slots_per_s': rate,
                                'achieved_tflops': 2 * rate / 1e12,
                                'source': 'profiles/pmc_summary.json (SQ_INSTS_VALU, SQ_INSTS_VALU_TRANS_F32)'}
            except Exception:
                traffic = executed = None
        roof = {
            'kernel': 'am_fine_kernel / am_phase_kernel (approxmatch passes A/B/C: 19 launches per forward, 7 of them am_fine_kernel)',
            'bound': 'mfma',
            'bound_detail': 'f32 VALU + transcendental pipe, priced at the f32 vector rate 157.3 TFLOP/s (= 78.6 T lane '
                            'issue slots/s x 2; numerically the f32 dense MFMA peak); no MFMA is used: the kernel is an '
                            'all-pairs exp-sum on difference-form distances',
            'peak': PEAK_F32_TFLOPS,
            'unit': 'TFLOP/s',
            'traffic': traffic,
            'avg_launch_us': phase_us,
            'launches_timed': phase_cnt,
            'concurrent_launches': lanes,
        }
        if executed:
            roof['achieved'] = executed['achieved_tflops']
            roof['frac'] = executed['achieved_tflops'] / PEAK_F32_TFLOPS
            roof['definition'] = ('achieved = vector issue slots the kernel EXECUTES per launch (PMC) x 2 flop / live launch '
                                  'duration: a fraction of the machine, <= 1 by construction.  The kernels skip terms that are '
                                  'exactly zero in float32, so the ALGORITHMIC rate (every pair of every reference pass / time) '
                                  'is reported beside it as `algorithmic` -- a speed-up over the reference\'s arithmetic at peak, '
                                  'not a utilisation -- and `noskip` repeats the measurement with the skips off, where '
                                  'algorithmic == executed')
        else:
            roof['achieved'] = None
            roof['frac'] = None
        roof['executed'] = executed
        roof['algorithmic'] = {'achieved': algorithmic, 'ratio_to_peak': (algorithmic / PEAK_F32_TFLOPS) if algorithmic else None,
                               'flop_per_launch': algo_flop_per_launch,
                               'note': 'counts the exactly-zero terms the kernels skip; may exceed 1'}
        try:
            roof['noskip'] = noskip_roofline(k2) if not args.no_noskip else {'skipped': '--no-noskip'}
            if roof['frac'] is None and 'frac' in roof['noskip']:
                roof['achieved'], roof['frac'] = roof['noskip']['achieved'], roof['noskip']['frac']
        except Exception as e:
            roof['noskip'] = {'error': repr(e)}
        result['roofline'] = roof
        ch_us = br['nndistance'] + br['nndistancegrad']
        result['roofline_chamfer'] = {
            'kernel': 'nn_fwd_kernel + nn_bwd_range_kernel (Chamfer fwd+bwd, BASELINE configs[1])',
            'hbm': {'achieved': CHAMFER_ALGO_BYTES / (ch_us * 1e-6) / 1e9, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                    'frac': CHAMFER_ALGO_BYTES / (ch_us * 1e-6) / 1e9 / PEAK_HBM_GBS},
            'valu_f32': {'achieved': 2 * pairs * 8 / (br['nndistance'] * 1e-6) / 1e12, 'peak': PEAK_F32_TFLOPS,
                         'unit': 'TFLOP/s', 'frac': 2 * pairs * 8 / (br['nndistance'] * 1e-6) / 1e12 / PEAK_F32_TFLOPS},
            'clouds_per_s': B_PER_GPU / (ch_us * 1e-6),
        }
        result['breakdown_us'] = br
        result['emd_clouds_per_s'] = B_PER_GPU / (br['match_cost_implicit_fwd_bwd'] * 1e-6)
        result['emd_clouds_per_s_materialised'] = B_PER_GPU / ((br['approxmatch_cost'] + br['matchcostgrad']) * 1e-6)
        if not args.no_extras:
            try:
                result['other_rows_us'] = other_rows_us(dev)
            except Exception as e:  # informational only: never lose the headline line
                result['other_rows_us'] = {'error': repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            ncl = args.cpu_clouds or min(B_PER_GPU, max(2, os.cpu_count() or 2))
            try:
                result['cpu_baseline'], result['parity_at_bench_size'] = cpu_baseline_and_parity(
                    recon, ref, recon_t.detach(), ref_t, ncl)
            except Exception as e:
                result['cpu_baseline'] = {'error': repr(e)}
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == '__main__':
    sys.exit(main())
