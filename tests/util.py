"""Synthetic clouds shared by tests, smoke and bench."""

from __future__ import annotations

import numpy as np


def ref_cloud(rng: np.random.Generator, b: int, n: int) -> np.ndarray:
    """Surface-like cloud: directions uniform on the sphere, radius U(0.3,1)^(1/3), centred, max-norm 1."""
    v = rng.standard_normal((b, n, 3))
    v /= np.linalg.norm(v, axis=2, keepdims=True) + 1e-12
    r = rng.uniform(0.3, 1.0, (b, n, 1)) ** (1.0 / 3.0)
    p = v * r
    p -= p.mean(axis=1, keepdims=True)
    p /= np.maximum(np.linalg.norm(p, axis=2).max(axis=1), 1e-12)[:, None, None]
    if n == 1:
        p = v * r  # a single point cannot be centred; keep it off the origin
    return p.astype(np.float32)


def recon_cloud(rng: np.random.Generator, ref: np.ndarray, m: int | None = None, sigma: float = 0.02) -> np.ndarray:
    """A 'trained autoencoder' output: row-permuted reference (optionally resampled to m points) + noise."""
    b, n, _ = ref.shape
    m = n if m is None else m
    out = np.empty((b, m, 3), np.float32)
    for i in range(b):
        perm = rng.permutation(n)
        idx = perm[np.arange(m) % n]
        out[i] = ref[i, idx] + rng.normal(0.0, sigma, (m, 3)).astype(np.float32)
    return out


def uniform_cloud(rng: np.random.Generator, b: int, n: int) -> np.ndarray:
    return rng.random((b, n, 3), dtype=np.float32)


def pair(seed: int, b: int, n: int, m: int | None = None, kind: str = 'recon'):
    rng = np.random.default_rng(seed)
    m = n if m is None else m
    if kind == 'uniform':
        return uniform_cloud(rng, b, n), uniform_cloud(rng, b, m)
    ref = ref_cloud(rng, b, m)
    return recon_cloud(rng, ref, n), ref
