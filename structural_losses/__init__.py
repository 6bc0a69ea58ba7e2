"""Drop-in for the reference's ``structural_losses`` package
(``external/pytorch_structural_losses/structural_losses/__init__.py:1-5``):
``from structural_losses import match_cost`` (``src/train/metrics_and_losses.py:10``) resolves here when
the repository root is on ``sys.path``.  Backed by ``libpcc_structural.so`` (HIP, gfx950)."""

from structural_losses.match_cost import match_cost
from structural_losses.nn_distance import nn_distance

__all__ = ['match_cost', 'nn_distance']
