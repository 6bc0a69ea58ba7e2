"""CPU tests of the on-disk format readers on synthetic files laid out like the reference's datasets."""

import numpy as np
import pytest
import torch


def _make_tree(tmp_path, rng):
    root = tmp_path / 'ShapeNetCore.v2.PC15k'
    for synset, count in (('02691156', 3), ('04379243', 2)):
        for part, k in (('train', count), ('val', 1), ('test', 1)):
            d = root / synset / part
            d.mkdir(parents=True)
            for i in range(k):
                np.save(d / f'{part}{i}.npy', (rng.standard_normal((600, 3)) * [1.0, 0.5, 2.0] + [3.0, -1.0, 0.5]))
    return root


def test_shapenet_npy_split(tmp_path):
    from pointcloudcounterfactual_amd.data import ShapeNetNpySplit, normalise, shapenet_split_paths

    rng = np.random.default_rng(0)
    root = _make_tree(tmp_path, rng)
    paths = shapenet_split_paths(root, 'train')
    assert len(paths) == 5 and len(shapenet_split_paths(root, 'train_val')) == 7
    assert len(shapenet_split_paths(root, 'test', synsets=['04379243'])) == 1
    ds = ShapeNetNpySplit(paths, n_input_points=128, resample=True, seed=1)
    assert len(ds) == 5 and sorted(set(ds.labels)) == [0, 1]
    inp, ref, label = ds[0]
    assert inp.shape == (128, 3) and ref.shape == (128, 3) and inp.dtype == torch.float32 and int(label) == 0
    # disjoint draws from the same normalised cloud
    rows = {tuple(r) for r in inp.numpy().tolist()}
    assert not any(tuple(r) in rows for r in ref.numpy().tolist())
    full = ds.pcd[0]
    np.testing.assert_allclose(full.mean(0), 0.0, atol=1e-6)
    np.testing.assert_allclose(np.linalg.norm(full, axis=1).max(), 1.0, rtol=1e-6)
    c, s = normalise(np.array([[0.0, 0.0, 0.0], [2.0, 0.0, 0.0]]))
    np.testing.assert_allclose(c, [[-1, 0, 0], [1, 0, 0]]) and np.testing.assert_allclose(s, 1.0)
    same = ShapeNetNpySplit(paths, n_input_points=128, resample=False, seed=1)[3]
    assert same[0] is same[1] and int(same[2]) == 1
    again = ShapeNetNpySplit(paths, n_input_points=128, resample=True, seed=1)[0]
    assert torch.equal(again[0], inp) and torch.equal(again[1], ref)  # seeded draws are reproducible
    with pytest.raises(ValueError):
        ShapeNetNpySplit(paths, n_input_points=400, resample=True)[0]


def test_modelnet_h5_needs_h5py(tmp_path):
    from pointcloudcounterfactual_amd.data import load_modelnet_h5

    try:
        import h5py  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError, match='h5py'):
            load_modelnet_h5(tmp_path, '*train*.h5', 1024, 20)
