"""``nn_distance`` (reference ``structural_losses/nn_distance.py:9-43``)."""

from pointcloudcounterfactual_amd.losses import NNDistanceFunction, nn_distance

__all__ = ['NNDistanceFunction', 'nn_distance']
