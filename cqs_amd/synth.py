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


def sparse_corpus(n: int, vocab: int = 30522, nnz_lo: int = 64, nnz_hi: int = 128, seed: int = 0x5BA2DE, power: float = 2.5):
    """Synthetic SPLADE document vectors as a forward CSR (doc_off u64 [n + 1], tokens u32, weights f32): per chunk about
    nnz_lo..nnz_hi DISTINCT token ids in ascending order (what `encode` emits: ids ascending, weight > threshold,
    src/splade/mod.rs:1049-1062; trained models keep 100-300, :44), drawn as vocab * u^power of sorted uniforms - a few
    tokens in most chunks, a long tail of rare ones - weights in (0.01, 2.5)."""
    rng = np.random.default_rng(seed)
    L = nnz_hi
    want = rng.integers(nnz_lo, nnz_hi + 1, size=n)
    step = max(1, min(n, (64 << 20) // (L * 8)))                # 64 MB of f64 per slab
    ar = np.arange(L, dtype=np.int64)
    parts, lens = [], np.zeros(n, dtype=np.int64)
    for a in range(0, n, step):
        b = min(n, a + step)
        e = rng.standard_exponential((b - a, L + 1))
        u = np.cumsum(e[:, :L], axis=1) / e.sum(axis=1, keepdims=True)      # sorted uniforms (order statistics)
        ids = (vocab * u ** power).astype(np.int64)
        ids = np.maximum.accumulate(ids - ar, axis=1) + ar       # strictly increasing inside a chunk
        np.minimum(ids, vocab - L + ar, out=ids)                 # ... and below vocab
        keep = rng.random((b - a, L)) < (want[a:b, None] / float(L))         # ~want of the L candidates, still ascending
        keep[:, 0] |= ~keep.any(axis=1)                          # never an empty chunk
        lens[a:b] = keep.sum(axis=1)
        parts.append(ids[keep].astype(np.uint32))
    off = np.zeros(n + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    tok = np.concatenate(parts) if parts else np.zeros(0, np.uint32)
    w = (rng.random(tok.size, dtype=np.float32) * np.float32(2.49) + np.float32(0.01)).astype(np.float32)
    return off, tok, w


def sparse_queries(count: int, terms: int, vocab: int = 30522, seed: int = 0x5BA2DF, power: float = 2.5):
    """Query vectors: `terms` distinct token ids each (same skew as the corpus), ascending like the encoder's output."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        t = np.unique(np.minimum((vocab * rng.random(terms * 2) ** power).astype(np.uint32), vocab - 1))
        t = np.sort(rng.permutation(t)[:terms]).astype(np.uint32)
        out.append((t, (rng.random(t.size, dtype=np.float32) * np.float32(1.9) + np.float32(0.1)).astype(np.float32)))
    return out
