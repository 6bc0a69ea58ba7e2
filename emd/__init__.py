"""Drop-in for the reference's ``emd`` package (``external/emd/emd/``): auction-based EMD on MI355X.

``from emd import emdModule`` as in ``external/README.md:27-39``.  (The reference's own ``emd_module.py:9`` does
``import emd_backend`` at top level while ``setup.py:14`` builds ``emd.emd_backend``; here the backend is
``emd.emd_backend`` and the module imports it from the package.)"""

from emd.emd_module import emdFunction, emdModule

__all__ = ['emdFunction', 'emdModule']
