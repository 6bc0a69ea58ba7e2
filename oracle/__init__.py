"""CPU oracle for the structural-loss hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package, and only as the checker.  The product path (``pointcloudcounterfactual_amd``,
``structural_losses``, ``emd``) never imports it.

``structural_oracle.c`` / ``auction_oracle.c`` are line-by-line C restatements of the reference's CUDA
kernels (file:line cited per function there); ``neighbour_oracle.py`` restates the reference's pure-torch
kNN / graph ops.  Parity pinning status: see the header of ``structural_oracle.c`` and DESIGN.md.
"""

from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, '_build', 'liboracle.so')
_lib: ctypes.CDLL | None = None

_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)
_i32p = ctypes.POINTER(ctypes.c_int)


def build(force: bool = False) -> str:
    """Compile the C oracle with gcc (``make -C oracle``)."""
    if force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
        for f in ('structural_oracle.c', 'auction_oracle.c', 'neighbour_oracle.c', 'Makefile')
    ):
        subprocess.check_call(['make', '-C', _HERE, '-s'] + (['-B'] if force else []))
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def _f(a: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _i(a: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a: np.ndarray, t=_f32p):
    return a.ctypes.data_as(t)


def set_threads(t: int) -> None:
    lib().oracle_set_threads(int(t))


def max_threads() -> int:
    return int(lib().oracle_max_threads())


def set_contraction(mode: int) -> None:
    """0 = fma(z,z,fma(x,x,y*y)) (canonical), 1 = no contraction, 2 = fma(z,z,fma(y,y,x*x))."""
    lib().oracle_set_contraction(int(mode))


def set_exp_mode(mode: int) -> None:
    """0 = libm expf(level*d2); 1 = exp2f((level*log2e)*d2) (the arrangement the HIP kernels use)."""
    lib().oracle_set_exp_mode(int(mode))


def _dims(set1: np.ndarray, set2: np.ndarray) -> tuple[int, int, int]:
    assert set1.ndim == 3 and set2.ndim == 3 and set1.shape[2] == 3 and set2.shape[2] == 3
    assert set1.shape[0] == set2.shape[0]
    return set1.shape[0], set1.shape[1], set2.shape[1]


def nndistance(set1, set2):
    """-> dist1[B,N] f32, idx1[B,N] i32, dist2[B,M] f32, idx2[B,M] i32 (nndistance.cu:125-128)."""
    set1, set2 = _f(set1), _f(set2)
    b, n, m = _dims(set1, set2)
    d1 = np.zeros((b, n), np.float32)
    i1 = np.zeros((b, n), np.int32)
    d2 = np.zeros((b, m), np.float32)
    i2 = np.zeros((b, m), np.int32)
    lib().oracle_nndistance(b, n, _p(set1), m, _p(set2), _p(d1), _p(i1, _i32p), _p(d2), _p(i2, _i32p))
    return d1, i1, d2, i2


def nndistance_f64(set1, set2):
    """float64 brute force, one direction: for each point of set1 the nearest of set2."""
    set1, set2 = _f(set1), _f(set2)
    b, n, m = _dims(set1, set2)
    d = np.zeros((b, n), np.float64)
    i = np.zeros((b, n), np.int32)
    lib().oracle_nndistance_f64(b, n, _p(set1), m, _p(set2), _p(d, _f64p), _p(i, _i32p))
    return d, i


def nndistancegrad(set1, set2, idx1, idx2, grad_dist1, grad_dist2):
    """-> grad1[B,N,3], grad2[B,M,3] (nndistance.cu:149-154)."""
    set1, set2 = _f(set1), _f(set2)
    b, n, m = _dims(set1, set2)
    idx1, idx2, g1, g2 = _i(idx1), _i(idx2), _f(grad_dist1), _f(grad_dist2)
    o1 = np.zeros((b, n, 3), np.float32)
    o2 = np.zeros((b, m, 3), np.float32)
    lib().oracle_nndistancegrad(b, n, _p(set1), m, _p(set2), _p(g1), _p(idx1, _i32p), _p(g2), _p(idx2, _i32p),
                                _p(o1), _p(o2))
    return o1, o2


def approxmatch(set1, set2):
    """-> match[B,M,N] f32, temp[B,2(N+M)] f32 (approxmatch.cu:299-307)."""
    set1, set2 = _f(set1), _f(set2)
    b, n, m = _dims(set1, set2)
    match = np.zeros((b, m, n), np.float32)
    temp = np.zeros((b, 2 * (n + m)), np.float32)
    lib().oracle_approxmatch(b, n, m, _p(set1), _p(set2), _p(match), _p(temp))
    return match, temp


def approxmatch_f64(set1, set2):
    set1, set2 = _f(set1), _f(set2)
    b, n, m = _dims(set1, set2)
    match = np.zeros((b, m, n), np.float64)
    temp = np.zeros((b, 2 * (n + m)), np.float64)
    lib().oracle_approxmatch_f64(b, n, m, _p(set1), _p(set2), _p(match, _f64p), _p(temp, _f64p))
    return match, temp


def matchcost(set1, set2, match):
    """-> cost[B] (approxmatch.cu:309-316)."""
    set1, set2, match = _f(set1), _f(set2), _f(match)
    b, n, m = _dims(set1, set2)
    assert match.shape == (b, m, n)
    out = np.zeros((b,), np.float32)
    lib().oracle_matchcost(b, n, m, _p(set1), _p(set2), _p(match), _p(out))
    return out


def matchcost_f64(set1, set2, match):
    set1, set2 = _f(set1), _f(set2)
    match = np.ascontiguousarray(match, dtype=np.float64)
    b, n, m = _dims(set1, set2)
    out = np.zeros((b,), np.float64)
    lib().oracle_matchcost_f64(b, n, m, _p(set1), _p(set2), _p(match, _f64p), _p(out, _f64p))
    return out


def matchcostgrad(set1, set2, match):
    """-> grad1[B,N,3], grad2[B,M,3] (approxmatch.cu:318-326)."""
    set1, set2, match = _f(set1), _f(set2), _f(match)
    b, n, m = _dims(set1, set2)
    g1 = np.zeros((b, n, 3), np.float32)
    g2 = np.zeros((b, m, 3), np.float32)
    lib().oracle_matchcostgrad(b, n, m, _p(set1), _p(set2), _p(match), _p(g1), _p(g2))
    return g1, g2


def matchcostgrad_f64(set1, set2, match):
    set1, set2 = _f(set1), _f(set2)
    match = np.ascontiguousarray(match, dtype=np.float64)
    b, n, m = _dims(set1, set2)
    g1 = np.zeros((b, n, 3), np.float64)
    g2 = np.zeros((b, m, 3), np.float64)
    lib().oracle_matchcostgrad_f64(b, n, m, _p(set1), _p(set2), _p(match, _f64p), _p(g1, _f64p), _p(g2, _f64p))
    return g1, g2


# ---- kNN (neighbour_oracle.c) -----------------------------------------------------------------------------------
_i64p = ctypes.POINTER(ctypes.c_int64)


def knn_diff(x, k: int, query_stride: int = 1):
    """Difference-form kNN of x[B,C,N] -> idx[B,N,k] int64 (pykeops_knn formula, neighbour_ops.py:77-82).
    query_stride > 1: only the rows q % query_stride == 0 are computed (the others stay 0)."""
    x = _f(x)
    b, c, n = x.shape
    idx = np.zeros((b, n, k), np.int64)
    lib().oracle_knn_diff_strided(b, c, n, int(k), _p(x), _p(idx, _i64p), int(query_stride))
    return idx


def knn_expanded(x, k: int, return_dist: bool = False):
    """Expanded-form kNN (torch_knn formula, neighbour_ops.py:53-74) with sequential-fma rounding."""
    x = _f(x)
    b, c, n = x.shape
    idx = np.zeros((b, n, k), np.int64)
    dist = np.zeros((b, n, n), np.float32) if return_dist else None
    lib().oracle_knn_expanded(b, c, n, int(k), _p(x), _p(idx, _i64p), _p(dist) if return_dist else None)
    return (idx, dist) if return_dist else idx


# ---- auction EMD (auction_oracle.c) ---------------------------------------------------------------------------------


def auction_forward(xyz1, xyz2, eps: float, iters: int):
    """-> dist[B,n] f32, assignment[B,n] i32, price[B,n] f32 (emd_cuda.cu:227-281, deterministic interpretation)."""
    xyz1, xyz2 = _f(xyz1), _f(xyz2)
    b, n, m = _dims(xyz1, xyz2)
    assert n == m
    dist = np.zeros((b, n), np.float32)
    ass = np.zeros((b, n), np.int32)
    price = np.zeros((b, n), np.float32)
    lib().oracle_auction_forward.restype = ctypes.c_int
    rc = lib().oracle_auction_forward(b, n, _p(xyz1), _p(xyz2), ctypes.c_float(eps), int(iters), _p(dist),
                                      _p(ass, _i32p), _p(price))
    if rc != 1:
        raise ValueError('auction: invalid input (n % 1024, b <= 512, iters >= 1)')
    return dist, ass, price


def auction_backward(xyz1, xyz2, grad_dist, assignment):
    xyz1, xyz2, g, a = _f(xyz1), _f(xyz2), _f(grad_dist), _i(assignment)
    b, n, _ = _dims(xyz1, xyz2)
    out = np.zeros((b, n, 3), np.float32)
    lib().oracle_auction_backward(b, n, _p(xyz1), _p(xyz2), _p(g), _p(a, _i32p), _p(out))
    return out
