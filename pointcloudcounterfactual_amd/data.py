"""On-disk formats either side of the hot path: what the reference's dataset classes read
and the samples they hand to the model, restated without the reference's config / drytorch machinery.

* ShapeNet (PointFlow split): ``ShapeNetCore.v2.PC15k/<synset>/<train|val|test>/<id>.npy``, float arrays ``[15000, 3]``
  (``src/data/shapenet.py:21-59,79-106``): each cloud is normalised once to the unit sphere
  (``src/data/augmentations.py:13-18``), the label is the rank of the synset folder name, and an item draws
  ``n_input_points`` points without replacement for the input cloud and -- with ``resample`` -- a disjoint second draw
  for the reference cloud.
* ModelNet40 (``modelnet40_hdf5_2048/*.h5``): datasets ``data [M,2048,3]``, ``label [M,1]`` and a cached kNN index
  ``index_<k> [M,N,k]`` int16 (``src/data/modelnet.py:130-165``).  Reading needs ``h5py``, which this image does not have:
  ``load_modelnet_h5`` imports it lazily and says so.

Augmentations (random rotation, jitter) are the caller's business; nothing here touches the accelerator.
"""

from __future__ import annotations

import pathlib
from collections.abc import Sequence
from typing import Any

import numpy as np
import torch


def normalise(cloud: np.ndarray) -> tuple[np.ndarray, float]:
    """Centre the cloud and scale its farthest point to distance 1 (``augmentations.py:13-18``); returns the scale."""
    cloud = cloud - cloud.mean(axis=0)
    std = float(np.max(np.sqrt(np.sum(cloud**2, axis=1))))
    return cloud / std, std


class ShapeNetNpySplit(torch.utils.data.Dataset):
    """``ShapenetFlowSplit`` (``shapenet.py:18-59``): items are ``(input_cloud[n,3], ref_cloud[n,3], label)``."""

    def __init__(self, paths: Sequence[pathlib.Path | str], n_input_points: int = 2048, resample: bool = False,
                 seed: int | None = None) -> None:
        self.paths = [pathlib.Path(p) for p in paths]
        self.n_input_points, self.resample = int(n_input_points), bool(resample)
        self.rng = np.random.default_rng(seed)
        self.pcd = [normalise(np.load(p, allow_pickle=False).astype(np.float64))[0].astype(np.float32) for p in self.paths]
        self.folder_id_list = [p.parent.parent.name for p in self.paths]
        mapping = {folder_id: i for i, folder_id in enumerate(sorted(set(self.folder_id_list)))}
        self.labels = [mapping[f] for f in self.folder_id_list]

    def __len__(self) -> int:
        return len(self.pcd)

    def __getitem__(self, index: int) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        cloud = self.pcd[index]
        n = self.n_input_points
        need = 2 * n if self.resample else n
        if cloud.shape[0] < need:
            raise ValueError(f'{self.paths[index]} has {cloud.shape[0]} points, {need} are needed')
        sampling = self.rng.choice(cloud.shape[0], size=need, replace=False)
        inp = torch.from_numpy(cloud[sampling[:n]])
        ref = torch.from_numpy(cloud[sampling[n:]]) if self.resample else inp
        return inp, ref, torch.tensor(self.labels[index])


def shapenet_split_paths(root: pathlib.Path | str, split: str, synsets: Sequence[str] | None = None) -> list[pathlib.Path]:
    """Files of one partition (``shapenet.py:84-101``); ``split`` in ``train | val | test | train_val``."""
    root = pathlib.Path(root)
    folders = sorted(f for f in root.glob('*') if f.is_dir() and (synsets is None or f.name in synsets))
    parts = ('train', 'val') if split == 'train_val' else (split,)
    return [p for folder in folders for part in parts for p in sorted((folder / part).glob('*.npy'))]


def load_modelnet_h5(path: pathlib.Path | str, wild_str: str, input_points: int, k: int) -> tuple[Any, Any, Any]:
    """``ModelNet40.load_h5`` (``modelnet.py:130-165``): ``(pcd [M,n,3] f32, index_k [M,n,k] int16, labels [M] int64)``.
    A missing ``index_<k>`` dataset is computed with this package's kNN on the accelerator when one is present."""
    try:
        import h5py  # type: ignore
    except ImportError as e:  # not installable here: no network
        raise ImportError('reading ModelNet40 .h5 files needs h5py, which is not installed in this environment') from e
    pcd_list, idx_list, label_list = [], [], []
    for h5_name in sorted(pathlib.Path(path).glob(wild_str)):
        with h5py.File(h5_name, 'r') as f:
            pcs = f['data'][:].astype('float32')[:, :input_points, :]
            label = f['label'][:].astype('int64')
            key = f'index_{k}'
            if key in f:
                index = f[key][:].astype(np.short)
            else:
                x = torch.from_numpy(pcs).transpose(1, 2).contiguous()
                if torch.cuda.is_available():
                    from pointcloudcounterfactual_amd.neighbour_ops import knn

                    index = knn(x.cuda(), k).cpu().numpy().astype(np.short)
                else:
                    from pointcloudcounterfactual_amd.neighbour_ops import torch_knn

                    index = torch_knn(x, k).numpy().astype(np.short)
        pcd_list.append(pcs)
        idx_list.append(index)
        label_list.append(label)
    return np.concatenate(pcd_list, 0), np.concatenate(idx_list, 0), np.concatenate(label_list, 0).ravel()
