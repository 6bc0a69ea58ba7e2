#!/usr/bin/env python3
"""End-to-end harness benchmarks for BASELINE configs[3] and configs[4].

  python bench_train.py --mode train --gpus N --steps K --warmup W   autoencoder training step, B=32 per GPU
  python bench_train.py --mode infer --gpus N ...                    classifier -> counterfactual (encoder, w-AE, codes, decoder) -> classifier + metric

One rank per GPU; gradients averaged by DDP over RCCL (training) / no collective at all (inference).  Under
``torch.distributed.run`` (RANK / WORLD_SIZE in the environment) the process is one rank; without it ``--gpus N`` (N > 1,
or ``--via-launcher`` at N = 1) starts its own N ranks as a CHILD of a parent that never touches the GPU -- bench.py's
launcher (reference: ``src/utils/parallel.py:37-53`` spawns its ranks itself).  Whenever it runs under a launcher --
also at world size 1 -- the process group is RCCL and the model is wrapped in DistributedDataParallel, so DDP's
autograd-hook threads, the library's lane streams and its private memory pool execute together.  A line whose
``n_gpus`` differs from ``--gpus`` is refused.  Prints one JSON line on rank 0 in the bench.py format.  The headline
metric of the repository stays bench.py; this file measures the rows SURVEY.md marks "next".
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument('--mode', choices=['train', 'infer'], default='train')
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch-per-gpu', type=int, default=32)
    ap.add_argument('--points', type=int, default=2048)
    ap.add_argument('--unfused', action='store_true', help='compose EdgeConv as the reference does ([B,2C,N,k] tensor)')
    ap.add_argument('--graph', action='store_true',
                    help='capture one step into a hipGraph (torch.cuda.CUDAGraph) and replay it: the step is ~1500 launches, '
                         'a third of its wall time is launch gaps')
    ap.add_argument('--via-launcher', action='store_true',
                    help='start the ranks through torch.distributed.run even for --gpus 1 (RCCL + DDP at world size 1)')
    args = ap.parse_args()
    launched = 'RANK' in os.environ and 'WORLD_SIZE' in os.environ
    if not launched and (args.gpus > 1 or args.via_launcher):
        import bench

        return bench.self_launch(args, script=__file__)

    import torch

    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a line for another job size')
    if not torch.cuda.is_available():
        raise SystemExit('bench_train.py needs a GPU (the HIP path has no CPU fallback)')
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist = None
    if launched:  # one process per GPU over RCCL (backend "nccl" on ROCm), also at world size 1
        import torch.distributed as dist_mod

        dist = dist_mod
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=dev)

    from pointcloudcounterfactual_amd import harness
    from pointcloudcounterfactual_amd.losses import chamfer_emd
    from tests.util import pair

    torch.manual_seed(1234 + 4 + rank)
    _, ref = pair(1234 + 4 + 1000 * rank, args.batch_per_gpu, args.points, args.points, 'recon')
    ref_t = torch.from_numpy(ref).to(dev)
    model_cls = harness.VQAutoencoder if args.mode == 'train' else harness.CounterfactualVQVAE
    model = model_cls(n_points=args.points, fused=not args.unfused).to(dev)
    clf = harness.DGCNNClassifier().to(dev)
    if args.mode == 'train':
        model.train()
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local]) if dist is not None else model
        opt = harness.make_optimizer(model)
        if args.graph:  # the optimiser state must live on the device to be captured
            opt = torch.optim.AdamW(model.parameters(), lr=4e-3, weight_decay=1e-3, capturable=True)

        def step() -> None:
            opt.zero_grad(set_to_none=True)
            out = net(ref_t)
            harness.autoencoder_loss(out, ref_t).mean().backward()
            opt.step()
    else:
        model.eval()
        clf.eval()

        @torch.inference_mode()
        def step() -> None:
            # evaluate_counterfactuals.py:61-88: classifier logits of the input -> counterfactual towards a target class
            # (encoder -> w-autoencoder latent step -> nearest codes -> decoder) -> classifier on the result + the
            # Chamfer / EMD metric between input and counterfactual
            logits = clf(ref_t)
            out = model.generate_counterfactual(ref_t, logits, target_dim=3, target_value=1.0)
            cham, emd = chamfer_emd(out['recon'], ref_t)
            _metric = cham + emd
            _flipped = clf(out['recon']).argmax(1) == 3

    def sync() -> None:
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    if args.graph:
        if dist is not None:
            raise SystemExit('--graph is a single-GPU measurement')
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step()
        step = graph.replay  # noqa: F811
    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    el = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    if rank == 0:
        print(json.dumps({
            'metric': 'clouds/sec ' + ('autoencoder train step (DGCNN enc + PCGen dec + Chamfer + EMD + AdamW)'
                                       if args.mode == 'train' else
                                       'counterfactual inference step (classifier + encoder + w-autoencoder + nearest codes + decoder + classifier + Chamfer/EMD metric)'),
            'value': args.batch_per_gpu * world * args.steps / el, 'unit': 'clouds/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': el / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'BASELINE configs[{3 if args.mode == "train" else 4}] harness, N={args.points}, '
                                   f'B={args.batch_per_gpu} per GPU, random-init weights',
                       'global_batch': args.batch_per_gpu * world,
                       'parallelism': f'dp{world}' + (' (DDP all-reduce over RCCL)' if args.mode == 'train' else ' (no collective)')},
            'edgeconv': 'unfused (reference composition)' if args.unfused else 'fused (no [B,2C,N,k] tensor)',
            'hipgraph': bool(args.graph),
            'launcher': 'torch.distributed.run + RCCL' + (' + DistributedDataParallel' if args.mode == 'train' else '')
                        if launched else 'single process',
            'peak_mem_gib': torch.cuda.max_memory_allocated() / 2**30,
        }), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == '__main__':
    sys.exit(main())
