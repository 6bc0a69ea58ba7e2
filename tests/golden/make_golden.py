#!/usr/bin/env python3
"""Generate the committed golden fixtures (run in the build container only; needs /root/reference).

Two families, both small ``.npz`` files of plain arrays (loaded with ``allow_pickle=False``):

``ref_neighbour_ops.npz``   inputs and outputs of the reference's OWN pure-torch functions in
    ``/root/reference/src/utils/neighbour_ops.py`` (torch_square_distance :43-50, self_square_distance :53-60,
    torch_knn :71-74, get_neighbours :85-94, graph_max_pooling :106-110, get_graph_features :113-119,
    graph_filtering :122-133), imported by file path and run on CPU, plus the two-line body of
    ``torch_chamfer`` (``src/train/metrics_and_losses.py:46-47``) evaluated over the imported
    ``torch_square_distance``.  The module imports ``pykeops`` at top level, which is not installed here and is
    never touched by the CPU functions; an empty placeholder module is registered for the import to succeed.
    These vectors pin the kNN / graph-op oracle and BASELINE config 1 (CPU sum-Chamfer).

``oracle_structural.npz``   outputs of ``oracle/structural_oracle.c`` for nndistance / nndistancegrad /
    approxmatch / matchcost / matchcostgrad on small seeded inputs.  The reference's CUDA kernels cannot be
    run here (no nvcc / NVIDIA GPU) and it ships no vectors, so these pin the restatement against regressions
    and give the GPU tests committed expected values; they are NOT outputs of the reference ("parity unpinned").
"""

import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = '/root/reference'


def load_reference_neighbour_ops():
    for name in ('pykeops', 'pykeops.torch'):
        if name not in sys.modules:
            mod = types.ModuleType(name)
            mod.set_verbose = lambda *_a, **_k: None  # type: ignore[attr-defined]
            mod.LazyTensor = object  # type: ignore[attr-defined]
            sys.modules[name] = mod
    spec = importlib.util.spec_from_file_location('ref_neighbour_ops', os.path.join(REF, 'src/utils/neighbour_ops.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def make_reference_fixture():
    ref = load_reference_neighbour_ops()
    out = {}
    g = torch.Generator().manual_seed(20261004)
    # Chamfer (config 1 flavour): [B,N,3] x [B,M,3]
    t1 = torch.rand(2, 96, 3, generator=g)
    t2 = torch.rand(2, 80, 3, generator=g)
    d = ref.torch_square_distance(t1, t2)
    out['cd_t1'], out['cd_t2'], out['cd_dist'] = t1.numpy(), t2.numpy(), d.numpy()
    out['cd_chamfer_sum'] = (torch.min(d, dim=-1)[0].sum(1) + torch.min(d, dim=-2)[0].sum(1)).numpy()
    # kNN / graph ops: x[B,C,N]
    for tag, (b, c, n, k) in {'c3': (2, 3, 64, 4), 'c3k20': (2, 3, 257, 20), 'c64': (2, 64, 200, 25)}.items():
        x = torch.randn(b, c, n, generator=g)
        out[f'{tag}_x'] = x.numpy()
        out[f'{tag}_k'] = np.int64(k)
        out[f'{tag}_selfdist'] = ref.self_square_distance(x).numpy()
        idx = ref.torch_knn(x, k)
        out[f'{tag}_knn'] = idx.numpy()
        idx2, feat = ref.get_graph_features(x, indices=torch.empty(0), k=k)
        assert torch.equal(idx, idx2)
        if c <= 3:  # the C=64 edge-feature tensor is 6.6 MB; its content is x[idx]-x | x, covered at C=3
            out[f'{tag}_graph_features'] = feat.numpy()
        out[f'{tag}_max_pool'] = ref.graph_max_pooling(x, indices=idx, k=k).numpy()
    xf = torch.rand(2, 3, 200, generator=g)
    out['filt_x'] = xf.numpy()
    out['filt_out'] = ref.graph_filtering(xf.clone(), k=4).numpy()
    # get_local_covariance (:97-103) on the kNN the reference computes itself
    xc = torch.randn(2, 3, 150, generator=g)
    out['cov_x'] = xc.numpy()
    out['cov_k'] = np.int64(16)
    out['cov_out'] = ref.get_local_covariance(xc.clone(), indices=torch.empty(0), k=16).numpy()
    # BASELINE configs[0] at its exact shape (ModelNet40-like N=1024, B=4, Chamfer only, CPU reference path):
    # torch_square_distance (:43-50) + the torch_chamfer body (metrics_and_losses.py:46-47) and its autograd
    # gradients.  The [4,1024,1024] matrix itself (16 MiB) is not stored: its row / column minima and argmins are.
    from tests.util import pair

    a, c = pair(1234 + 1, 4, 1024, 1024, 'recon')
    t1 = torch.from_numpy(a).requires_grad_(True)
    t2 = torch.from_numpy(c).requires_grad_(True)
    d = ref.torch_square_distance(t1, t2)
    m1, m2 = torch.min(d, dim=-1), torch.min(d, dim=-2)
    loss = m1[0].sum(1) + m2[0].sum(1)
    loss.sum().backward()
    out['cfg1_t1'], out['cfg1_t2'] = a, c
    out['cfg1_dist1'], out['cfg1_idx1'] = m1[0].detach().numpy(), m1[1].numpy().astype(np.int32)
    out['cfg1_dist2'], out['cfg1_idx2'] = m2[0].detach().numpy(), m2[1].numpy().astype(np.int32)
    out['cfg1_chamfer_sum'] = loss.detach().numpy()
    out['cfg1_grad1'], out['cfg1_grad2'] = t1.grad.numpy(), t2.grad.numpy()
    np.savez_compressed(os.path.join(HERE, 'ref_neighbour_ops.npz'), **out)
    print('wrote ref_neighbour_ops.npz', {k: v.shape for k, v in out.items()})


def make_oracle_fixture():
    import oracle
    from tests.util import pair

    oracle.set_threads(4)
    out = {}
    for tag, (b, n, m, kind) in {
        'a': (2, 3, 5, 'uniform'), 'b': (2, 64, 64, 'recon'), 'c': (2, 257, 130, 'uniform'),
        'd': (1, 513, 512, 'recon'), 'e': (2, 128, 256, 'recon'),
    }.items():
        s1, s2 = pair(ord(tag), b, n, m, kind)
        d1, i1, d2, i2 = oracle.nndistance(s1, s2)
        rng = np.random.default_rng(ord(tag))
        g1 = rng.standard_normal((b, n)).astype(np.float32)
        g2 = rng.standard_normal((b, m)).astype(np.float32)
        gr1, gr2 = oracle.nndistancegrad(s1, s2, i1, i2, g1, g2)
        match, temp = oracle.approxmatch(s1, s2)
        cost = oracle.matchcost(s1, s2, match)
        mg1, mg2 = oracle.matchcostgrad(s1, s2, match)
        for k, v in dict(set1=s1, set2=s2, dist1=d1, idx1=i1, dist2=d2, idx2=i2, gdist1=g1, gdist2=g2, nngrad1=gr1,
                         nngrad2=gr2, match=match.astype(np.float32), temp=temp, cost=cost, mgrad1=mg1,
                         mgrad2=mg2).items():
            out[f'{tag}_{k}'] = v
    np.savez_compressed(os.path.join(HERE, 'oracle_structural.npz'), **out)
    print('wrote oracle_structural.npz', len(out), 'arrays')


if __name__ == '__main__':
    make_reference_fixture()
    make_oracle_fixture()
