"""The five backend functions the reference builds as a pybind module
(``external/pytorch_structural_losses/src/structural_loss.cpp:129-135``; typing stub
``structural_losses_backend.pyi:5-18``), served by the HIP C-ABI library."""

from pointcloudcounterfactual_amd.backend import ApproxMatch, MatchCost, MatchCostGrad, NNDistance, NNDistanceGrad

__all__ = ['ApproxMatch', 'MatchCost', 'MatchCostGrad', 'NNDistance', 'NNDistanceGrad']
