"""Drop-in ``pykeops`` for ROCm: the reference imports ``pykeops`` (``src/utils/neighbour_ops.py:5,11,13``) for its
accelerator path and PyKeOps has no ROCm backend.  This package exposes the slice the reference uses
(``set_verbose``, ``pykeops.torch.LazyTensor``) on the hand-written HIP kernels of
``pointcloudcounterfactual_amd`` -- see ``pointcloudcounterfactual_amd/keops_shim.py`` for what is covered."""

from pointcloudcounterfactual_amd.keops_shim import set_verbose

__all__ = ['set_verbose']
