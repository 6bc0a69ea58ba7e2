"""MI355X-native structural losses for PointCloudCounterfactual's hot path.

Importing the package loads ``lib/libpcc_structural.so`` (hand-written HIP for gfx950); it raises
``ImportError`` if the library has not been built -- there is no CPU / PyTorch fallback.
"""

from pointcloudcounterfactual_amd import _lib, backend  # noqa: F401
from pointcloudcounterfactual_amd.losses import (  # noqa: F401
    MatchCostFunction,
    NNDistanceFunction,
    chamfer,
    chamfer_emd,
    match_cost,
    nn_distance,
    torch_chamfer,
)

__all__ = ['match_cost', 'nn_distance', 'chamfer', 'chamfer_emd', 'torch_chamfer', 'MatchCostFunction', 'NNDistanceFunction',
           'backend']
