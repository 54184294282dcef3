"""Synthetic corpora / queries shared by tests and bench.

CPU generators are seeded numpy Philox (counter-based, reproducible); the
reference's own deterministic test generators are restated for the structured /
tie-stress fixtures.  Rows are L2-normalised with the reference formula
(`normalize_l2`, src/embedder/pooling.rs:60-67: f32 sum of squares, scale by
1/sqrt) so self-dots may exceed 1 by ~1e-6 exactly as the reference observes
(src/hnsw/mod.rs:276-279).
"""
from __future__ import annotations

import numpy as np

SEED_CORPUS = 0xC950001
SEED_QUERY = 0xC950002
DIM = 768


def _normalize_rows(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.float32, copy=False)
    nsq = np.einsum("ij,ij->i", x, x, dtype=np.float32)
    inv = np.where(nsq > 0, np.float32(1.0) / np.sqrt(nsq, dtype=np.float32), np.float32(1.0)).astype(np.float32)
    return (x * inv[:, None]).astype(np.float32)


def gaussian_unit(n: int, dim: int = DIM, seed: int = SEED_CORPUS) -> np.ndarray:
    """i.i.d. N(0,1) components, row-normalised."""
    rng = np.random.Generator(np.random.Philox(key=seed))
    return _normalize_rows(rng.standard_normal((n, dim), dtype=np.float32))


def sin_embedding(seed: int, dim: int = DIM, scale: float = 0.1) -> np.ndarray:
    """`make_test_embedding(seed)` = sin(seed*0.1 + i*0.001), normalised (src/hnsw/mod.rs:928-940);
    scale=10.0 gives the CAGRA test variant (src/cagra.rs:1815-1825)."""
    i = np.arange(dim, dtype=np.float32)
    v = np.sin(np.float32(seed) * np.float32(scale) + i * np.float32(0.001)).astype(np.float32)
    norm = np.sqrt(np.sum(v * v, dtype=np.float32), dtype=np.float32)
    return (v / norm).astype(np.float32) if norm > 0 else v


def sin_corpus(n: int, dim: int = DIM, scale: float = 0.1) -> np.ndarray:
    return np.stack([sin_embedding(s, dim, scale) for s in range(n)])


def mock_embedding(seed: float, dim: int = DIM) -> np.ndarray:
    """`mock_embedding(seed)`: constant vector, normalised (src/test_helpers.rs:20-29)."""
    v = np.full((dim,), np.float32(seed), dtype=np.float32)
    norm = np.sqrt(np.sum(v * v, dtype=np.float32), dtype=np.float32)
    return (v / norm).astype(np.float32) if norm > 0 else v


def embedding_with_lead(lead: float, dim: int = DIM) -> np.ndarray:
    """src/search/query.rs:2092-2100."""
    v = np.full((dim,), np.float32(0.05), dtype=np.float32)
    v[0] = np.float32(lead)
    norm = np.sqrt(np.sum(v * v, dtype=np.float32), dtype=np.float32)
    return (v / norm).astype(np.float32)
