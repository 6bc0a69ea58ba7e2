"""The slice of ``pykeops.torch.LazyTensor`` the reference uses, on the HIP kernels of this package.

PyKeOps (``pykeops>=2.3``, ``pyproject.toml:15`` of the reference; un-vendored, no ROCm backend) does three jobs in
the reference, all on the lazy matrix ``D[b,i,j] = ((x_i - y_j) ** 2).sum(-1)`` built by
``pykeops_square_distance`` (``src/utils/neighbour_ops.py:35-40``):

* ``D.argmin(axis=1)`` / ``D.argmin(axis=2)``  -- ``pykeops_chamfer`` (``src/train/metrics_and_losses.py:32-36``)
  and ``VectorQuantizer.quantize`` (``src/module/quantize.py:26-28``);
* ``D.argKmin(k, dim=2)`` on a cloud against itself -- ``pykeops_knn`` (``neighbour_ops.py:77-82``);
* ``D.sum(1)`` (differentiable)                  -- ``quantize.py:31``.

``LazyTensor`` below accepts exactly that expression shape -- ``LazyTensor(x[:, :, None, :])``,
``LazyTensor(y[:, None, :, :])``, ``-``, ``** 2``, ``.sum(-1)`` -- and maps the reductions onto
``pcc_nndistance`` (3-D clouds, both argmins from one launch), ``pcc_knn``, ``pcc_pair_argmin`` and
``pcc_pair_sqdist_sum``.  Anything else raises ``NotImplementedError`` naming what is missing; nothing falls back to
dense torch math.  Registered as the drop-in packages ``pykeops`` / ``pykeops.torch`` at the repo root, so the
reference's ``src/utils/neighbour_ops.py`` imports it unchanged.  No PyKeOps output exists to compare against
(SURVEY.md section 8c): parity at this boundary is pinned against float64 brute force only ("parity unpinned" with
respect to PyKeOps itself).
"""

from __future__ import annotations

from typing import Any

import torch
from torch.autograd import Function

from pointcloudcounterfactual_amd import _lib, backend

_L = _lib.lib


def _stream(x: torch.Tensor) -> int:
    return torch.cuda.current_stream(x.device).cuda_stream


def _need_device(x: torch.Tensor, name: str) -> None:
    if x.device.type != 'cuda':
        raise RuntimeError(f'{name} must be a CUDA tensor (the reference only reaches PyKeOps on the accelerator, '
                           'neighbour_ops.py:29,65)')
    if x.dtype != torch.float32:
        raise RuntimeError(f'expected scalar type Float but found {x.dtype} ({name})')


class LazyTensor:
    """Symbolic point variable: ``x[B,N,1,D]`` indexes rows (i), ``x[B,1,M,D]`` columns (j); ``x[B,1,1,D]`` is a
    one-point cloud that takes whichever role its partner leaves free."""

    def __init__(self, x: torch.Tensor, axis: int | None = None) -> None:
        if not isinstance(x, torch.Tensor) or x.dim() != 4:
            raise NotImplementedError('LazyTensor shim: expected a [B,N,1,D] or [B,1,M,D] torch tensor '
                                      '(pykeops_square_distance, neighbour_ops.py:37-38)')
        b, n, m, d = x.shape
        if axis is not None and axis not in (0, 1):
            raise ValueError('axis must be 0 (i) or 1 (j)')
        if n == 1 and m == 1:
            self.role = {None: 'p', 0: 'i', 1: 'j'}[axis]
        elif m == 1:
            self.role = 'i'
        elif n == 1:
            self.role = 'j'
        else:
            raise NotImplementedError('LazyTensor shim: dense [B,N,M,D] operands are not used by the reference')
        if axis is not None and self.role != 'p' and self.role != 'ij'[axis]:
            raise ValueError('axis contradicts the singleton dimension of the tensor')
        self.data = x.reshape(b, max(n, m), d)  # [B, points, D]; a view whenever x is one

    def __sub__(self, other: Any) -> '_Difference':
        if not isinstance(other, LazyTensor):
            raise NotImplementedError('LazyTensor shim: only variable - variable is used by the reference')
        roles = (self.role, other.role)
        if roles in (('i', 'j'), ('i', 'p'), ('p', 'j'), ('p', 'p')):
            return _Difference(self.data, other.data)
        if roles in (('j', 'i'), ('j', 'p'), ('p', 'i')):
            return _Difference(other.data, self.data)  # (y_j - x_i): the square is the same
        raise NotImplementedError(f'LazyTensor shim: difference of two {roles[0]}-variables')


class _Difference:
    def __init__(self, rows: torch.Tensor, cols: torch.Tensor) -> None:
        if rows.shape[0] != cols.shape[0] or rows.shape[2] != cols.shape[2]:
            raise ValueError(f'incompatible operands {tuple(rows.shape)} and {tuple(cols.shape)}')
        self.rows, self.cols = rows, cols

    def __pow__(self, p: int) -> '_SquaredDifference':
        if p != 2:
            raise NotImplementedError('LazyTensor shim: only ** 2 is used by the reference (neighbour_ops.py:39)')
        return _SquaredDifference(self.rows, self.cols)


class _SquaredDifference(_Difference):
    def sum(self, axis: int | None = None, dim: int | None = None) -> 'SquareDistance':
        ax = dim if axis is None else axis
        if ax not in (-1, 3):
            raise NotImplementedError('LazyTensor shim: the channel sum .sum(-1) must come first')
        return SquareDistance(self.rows, self.cols)


class _PairSqdistSum(Function):
    """``out[b,i] = sum_j |p_i - q_j|^2`` with the gradients KeOps' autograd gives for ``D.sum(axis)``."""

    @staticmethod
    def forward(ctx: Any, *args: Any, **kwargs: Any) -> torch.Tensor:
        p, q = args
        p, q = p.contiguous(), q.contiguous()
        b, n_p, d = p.shape
        n_q = q.shape[1]
        out = torch.empty((b, n_p), dtype=torch.float32, device=p.device)
        with torch.cuda.device(p.device):
            _lib.check(_L.pcc_pair_sqdist_sum(b, n_p, n_q, d, p.data_ptr(), q.data_ptr(), out.data_ptr(), _stream(p)),
                       'pair_sqdist_sum')
        ctx.save_for_backward(p, q)
        return out

    @staticmethod
    def backward(ctx: Any, *grad_outputs: Any) -> Any:
        p, q = ctx.saved_tensors
        g = grad_outputs[0].contiguous().float()
        b, n_p, d = p.shape
        n_q = q.shape[1]
        gp = torch.empty_like(p) if ctx.needs_input_grad[0] else None
        gq = torch.empty_like(q) if ctx.needs_input_grad[1] else None
        with torch.cuda.device(p.device):
            _lib.check(_L.pcc_pair_sqdist_sum_bwd(b, n_p, n_q, d, p.data_ptr(), q.data_ptr(), g.data_ptr(),
                                                  gp.data_ptr() if gp is not None else None,
                                                  gq.data_ptr() if gq is not None else None, _stream(p)),
                       'pair_sqdist_sum_bwd')
        return gp, gq


class SquareDistance:
    """Lazy ``D[b,i,j]``; ``rows[B,N,D]`` index i, ``cols[B,M,D]`` index j.  Reductions follow KeOps' output
    shapes: reducing ``axis=2`` (j) gives ``[B,N,1]`` (``[B,N,K]`` for argKmin), ``axis=1`` (i) gives ``[B,M,1]``."""

    def __init__(self, rows: torch.Tensor, cols: torch.Tensor) -> None:
        self.rows, self.cols = rows, cols
        self._nn: list[torch.Tensor] | None = None  # both directions of the 3-D nearest-neighbour search

    @property
    def shape(self) -> tuple[int, int, int]:
        return self.rows.shape[0], self.rows.shape[1], self.cols.shape[1]

    @staticmethod
    def _axis(axis: int | None, dim: int | None) -> int:
        ax = dim if axis is None else axis
        if ax not in (1, 2):
            raise NotImplementedError('LazyTensor shim: reductions run over axis 1 (i) or 2 (j)')
        return ax

    def _prepared(self) -> tuple[torch.Tensor, torch.Tensor]:
        _need_device(self.rows, 'x_i')
        _need_device(self.cols, 'y_j')
        return self.rows.contiguous(), self.cols.contiguous()

    def _nearest(self) -> list[torch.Tensor]:
        if self._nn is None:
            rows, cols = self._prepared()
            with torch.no_grad():
                self._nn = backend.NNDistance(rows.detach(), cols.detach())  # dist1, idx1 (over j), dist2, idx2 (over i)
        return self._nn

    def _general(self, ax: int) -> tuple[torch.Tensor, torch.Tensor]:
        rows, cols = self._prepared()
        p, q = (rows, cols) if ax == 2 else (cols, rows)
        b, n_p, d = p.shape
        idx = torch.empty((b, n_p), dtype=torch.int64, device=p.device)
        val = torch.empty((b, n_p), dtype=torch.float32, device=p.device)
        with torch.cuda.device(p.device):
            _lib.check(_L.pcc_pair_argmin(b, n_p, q.shape[1], d, p.data_ptr(), q.data_ptr(), idx.data_ptr(),
                                          val.data_ptr(), _stream(p)), 'pair_argmin')
        return idx, val

    def argmin(self, axis: int | None = None, dim: int | None = None) -> torch.Tensor:
        """Index of the nearest partner, int64 ``[B,N,1]`` (axis=2) or ``[B,M,1]`` (axis=1); lowest index on ties."""
        ax = self._axis(axis, dim)
        if self.rows.shape[2] == 3 and self.rows.shape[1] and self.cols.shape[1]:
            _d1, i1, _d2, i2 = self._nearest()
            return (i1 if ax == 2 else i2).long().unsqueeze(-1)
        return self._general(ax)[0].unsqueeze(-1)

    def min(self, axis: int | None = None, dim: int | None = None) -> torch.Tensor:
        """Smallest squared distance ``[B,N,1]`` / ``[B,M,1]`` (a constant of the graph: the reference never
        differentiates it, ``metrics_and_losses.py:22-30``)."""
        ax = self._axis(axis, dim)
        if self.rows.shape[2] == 3 and self.rows.shape[1] and self.cols.shape[1]:
            d1, _i1, d2, _i2 = self._nearest()
            return (d1 if ax == 2 else d2).unsqueeze(-1)
        return self._general(ax)[1].unsqueeze(-1)

    def argKmin(self, K: int, axis: int | None = None, dim: int | None = None) -> torch.Tensor:
        """``[B,N,K]`` int64 nearest neighbours, ascending; implemented for a cloud against itself
        (``pykeops_knn``, neighbour_ops.py:77-82)."""
        ax = self._axis(axis, dim)
        same = (self.rows.data_ptr() == self.cols.data_ptr() and self.rows.shape == self.cols.shape
                and self.rows.stride() == self.cols.stride())
        if not same:
            raise NotImplementedError('LazyTensor shim: argKmin between two different clouds is not used by the reference')
        del ax  # D is symmetric
        from pointcloudcounterfactual_amd.neighbour_ops import hip_knn

        rows, _ = self._prepared()
        return hip_knn(rows.detach().transpose(1, 2).contiguous(), int(K))

    def sum(self, axis: int | None = None, dim: int | None = None) -> torch.Tensor:
        """``sum_i D`` -> ``[B,M,1]`` (axis=1) or ``sum_j D`` -> ``[B,N,1]`` (axis=2); differentiable."""
        ax = self._axis(axis, dim)
        _need_device(self.rows, 'x_i')
        _need_device(self.cols, 'y_j')
        p, q = (self.rows, self.cols) if ax == 2 else (self.cols, self.rows)
        return _PairSqdistSum.apply(p, q).unsqueeze(-1)


def set_verbose(*_args: Any, **_kwargs: Any) -> None:
    """``pykeops.set_verbose`` (neighbour_ops.py:13): nothing is compiled at run time here."""
