"""ctypes binding of ``libpcc_structural.so`` (the C ABI declared in ``include/pcc_structural.h``).

The library is hand-written HIP for gfx950; there is no CPU or PyTorch fallback.  If the shared
object has not been built (``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C pointcloudcounterfactual_amd/csrc``) importing this module raises ``ImportError``.
"""

from __future__ import annotations

import ctypes
import os

# torch FIRST: libpcc_structural.so needs libamdhip64.so.7, and the process must end up with ONE HIP runtime.  With
# torch imported before the library is loaded, the loader resolves that name to the runtime torch already brought in
# (the copy bundled in torch/lib); loaded the other way round, the library pulls /opt/rocm's copy, torch then loads
# its own, and launches through the first runtime on memory of the second fail with hipErrorNoDevice.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('PCC_LIB_OVERRIDE') or os.path.join(_HERE, 'lib', 'libpcc_structural.so')  # override: A/B builds only

_vp = ctypes.c_void_p
_int = ctypes.c_int

# name -> (restype, argtypes); must list every symbol include/pcc_structural.h declares.
ABI: dict[str, tuple[object, list[object]]] = {
    'pcc_version': (ctypes.c_char_p, []),
    'pcc_last_error': (ctypes.c_char_p, []),
    'pcc_last_status': (_int, []),
    'pcc_profile_enable': (None, [_int]),
    'pcc_profile_reset': (None, []),
    'pcc_profile_read': (_int, [ctypes.c_char_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]),
    'nndistance': (None, [_int, _int, _vp, _int, _vp, _vp, _vp, _vp, _vp, _vp]),
    'pcc_nndistance': (_int, [_int, _int, _vp, _int, _vp, _vp, _vp, _vp, _vp, _vp]),
    'nndistancegrad': (None, [_int, _int, _vp, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'pcc_nndistancegrad': (_int, [_int, _int, _vp, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'pcc_chamfer_loss': (_int, [_int, _int, _vp, _int, _vp, _int, _vp, _vp, _vp, _vp, _vp, _vp]),
    'pcc_chamfer_loss_grad': (_int, [_int, _int, _vp, _int, _vp, _vp, _vp, _vp, _int, _int, _vp, _vp, _vp]),
    'pcc_chamfer_emd': (_int, [_int, _int, _vp, _int, _vp, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'pcc_chamfer_emd_grad': (_int, [_int, _int, _vp, _int, _vp, _vp, _vp, _vp, _int, _int, _vp, _vp, _vp, _int, _vp, _vp, _vp]),
    'approxmatch': (None, [_int, _int, _int, _vp, _vp, _vp, _vp, _vp]),
    'pcc_approxmatch': (_int, [_int, _int, _int, _vp, _vp, _vp, _vp, _vp]),
    'pcc_approxmatch_workspace_bytes': (ctypes.c_size_t, [_int, _int, _int]),
    'pcc_approxmatch_ws': (_int, [_int, _int, _int, _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t, _vp]),
    'pcc_approxmatch_cost': (_int, [_int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp]),
    'matchcost': (None, [_int, _int, _int, _vp, _vp, _vp, _vp, _vp]),
    'pcc_matchcost': (_int, [_int, _int, _int, _vp, _vp, _vp, _vp, _vp]),
    'matchcostgrad': (None, [_int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp]),
    'pcc_matchcostgrad': (_int, [_int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp]),
    'pcc_matchcostgrad_scaled': (_int, [_int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'pcc_match_cost': (_int, [_int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    # include/pcc_neighbour.h
    'pcc_knn': (_int, [_int, _int, _int, _int, _vp, _vp, _vp]),
    'pcc_gather_neighbours': (_int, [_int, _int, _int, _int, _vp, _vp, _vp, _vp]),
    'pcc_gather_neighbours_bwd': (_int, [_int, _int, _int, _int, _vp, _vp, _vp, _vp]),
    'pcc_graph_features': (_int, [_int, _int, _int, _int, _vp, _vp, _vp, _vp]),
    'pcc_graph_features_bwd': (_int, [_int, _int, _int, _int, _vp, _vp, _vp, _vp]),
    'pcc_graph_max_pool': (_int, [_int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp]),
    'pcc_graph_max_pool_bwd': (_int, [_int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp]),
    'pcc_global_pool': (_int, [_int, _int, _int, _vp, _vp, _vp, _vp, _vp]),
    'pcc_neighbour_sum': (_int, [_int, _int, _int, _int, _vp, _vp, _vp, _vp]),
    'pcc_neighbour_sum_bwd': (_int, [_int, _int, _int, _int, _vp, _vp, _vp, _vp]),
    'pcc_neighbour_minmax_target': (_int, [_int, _int, _int, _int, _vp, _vp, _vp, _vp]),
    'pcc_pair_argmin': (_int, [_int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp]),
    'pcc_pair_sqdist_sum': (_int, [_int, _int, _int, _int, _vp, _vp, _vp, _vp]),
    'pcc_pair_sqdist_sum_bwd': (_int, [_int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp]),
    'pcc_bn_stats': (_int, [_int, _int, _int, _vp, _vp, _vp, _vp]),
    'pcc_bn_relu_res_fwd': (_int, [_int, _int, _int, _vp, _vp, _vp, ctypes.c_float, _vp, _vp, _vp, _int, _int, _vp, _vp]),
    'pcc_bn_relu_bwd': (_int, [_int, _int, _int, _vp, _vp, _vp, ctypes.c_float, _vp, _vp, _vp, _int, _vp, _vp, _vp, _vp]),
    # include/pcc_emd.h
    'pcc_auction_forward': (_int, [_int, _int, _vp, _vp, ctypes.c_float, _int, _vp, _vp, _vp]),
    'pcc_auction_status': (_int, []),
    'pcc_auction_backward': (_int, [_int, _int, _vp, _vp, _vp, _vp, _vp, _vp]),
    # include/pcc_test_hooks.h (inert without PCC_TEST_HOOKS=1)
    'pcc_test_inject_auction_failure': (_int, []),
    'pcc_test_set_tuning': (_int, [_int, _int]),
}


def _load() -> ctypes.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f'{LIB_PATH} is missing: the HIP extension has not been built '
            '(run `make -C pointcloudcounterfactual_amd/csrc`); there is no fallback path.'
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in ABI.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    return lib


lib = _load()


def check(status: int, what: str) -> None:
    """Raise the error the reference raises from its launchers (approxmatch.cu:303-306)."""
    if status != 0:
        msg = lib.pcc_last_error().decode() or f'HIP kernel failed : {status}'
        raise RuntimeError(f'{what}: {msg}')


# include/pcc_test_hooks.h: measurement / bit-identity switches (inert unless PCC_TEST_HOOKS=1 is in the environment)
TUNING = {'pair_plain_order': 1, 'am_nocull': 2, 'am_nosplit': 3, 'am_noresident': 4, 'edge_scatter': 5,
          'nbrsum_scatter': 6, 'auction_cluster': 7, 'knn_nosplit': 8, 'am_lanes': 9, 'nn_head': 11}


def set_tuning(name: str, value: int) -> None:
    """Set a measurement switch of the library (0 = the product's behaviour); raises if the hooks are not armed."""
    if lib.pcc_test_set_tuning(TUNING[name], int(value)) != 1:
        raise RuntimeError('test hooks are not armed: set PCC_TEST_HOOKS=1 before the library is loaded')
