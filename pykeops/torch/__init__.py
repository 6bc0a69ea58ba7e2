"""``from pykeops.torch import LazyTensor`` (reference ``src/utils/neighbour_ops.py:11``)."""

from pointcloudcounterfactual_amd.keops_shim import LazyTensor

__all__ = ['LazyTensor']
