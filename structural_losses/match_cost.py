"""``match_cost`` (reference ``structural_losses/match_cost.py:11-50``)."""

from pointcloudcounterfactual_amd.losses import MatchCostFunction, match_cost

__all__ = ['MatchCostFunction', 'match_cost']
