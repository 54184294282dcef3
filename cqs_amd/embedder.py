"""Host-side mirror of the reference's `Embedder` over the C ABI (embed section).

Reference surface mirrored (src/embedder/core.rs unless noted):
  Embedder::new / embed_documents :718 / embed_query :768 / embed_batch :994-1273 /
  embedding_dim :961 / warm :933; ModelConfig::embeddinggemma_300m (models.rs:455-470: prefixes,
  dim 768, max_seq 2048, pad_id 0, inputs `input_ids` + `attention_mask`); embed_batch_size
  (models.rs:789-817 -> 32); normalize_l2 (pooling.rs:60-67); pad_2d_i64_from_encodings
  (pooling.rs:40-57); EmbedderError (mod.rs:36-60).

Only the `session.run` block is replaced by the HIP engine; tokenisation, prefixes, chunking by
`embed_batch_size`, truncation to max_seq and the LRU query cache are host logic as in the
reference.  The tokenizer is injected: the real model ships `tokenizer.json` (loaded with the
`tokenizers` package when a model directory is given); tests use a deterministic stand-in.
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
from typing import Callable, List, Optional, Sequence

import numpy as np

from . import _lib
from .index import HipError

DOC_PREFIX = "title: none | text: "                 # models.rs:458
QUERY_PREFIX = "task: search result | query: "      # models.rs:457
MAX_QUERY_BYTES = 32 * 1024                         # core.rs:765


class EmbedderError(RuntimeError):
    """src/embedder/mod.rs:36-60 (EmptyQuery, InferenceFailed, Tokenizer...)."""


def embed_batch_size(dim: int, max_seq_length: int) -> int:
    """models.rs:802-817."""
    d = float(max(dim, 1))
    s = float(max(max_seq_length, 1))
    scaled = int(max(64.0 * (1024.0 / d) * max(512.0 / s, 0.25), 1.0))
    p = 1
    while p < scaled:
        p <<= 1
    return max(2, min(p, 256))


def normalize_l2(v: np.ndarray) -> np.ndarray:
    """pooling.rs:60-67: f32 left-to-right sum of squares; zero stays zero."""
    v = np.asarray(v, dtype=np.float32)
    norm_sq = np.float32(0.0)
    for x in v:
        norm_sq = np.float32(norm_sq + x * x)
    if norm_sq > 0:
        v = v * (np.float32(1.0) / np.sqrt(norm_sq, dtype=np.float32))
    return v.astype(np.float32)


def pad_2d_i64(rows: Sequence[Sequence[int]], max_len: int, pad_value: int) -> np.ndarray:
    """pooling.rs:40-57: right-pad / truncate to [batch, max_len] i64."""
    arr = np.full((len(rows), max_len), pad_value, dtype=np.int64)
    for i, r in enumerate(rows):
        r = list(r)[:max_len]
        arr[i, :len(r)] = r
    return arr


def default_config() -> _lib.EmbedConfig:
    cfg = _lib.EmbedConfig()
    _lib.load().cqs_hip_embed_config_default(C.byref(cfg))
    return cfg


class HipEmbedEngine:
    """Thin owner of a `cqs_hip_embedder*` (the ORT-session replacement)."""

    def __init__(self, cfg: Optional[_lib.EmbedConfig] = None, device: int = 0, handle: Optional[int] = None):
        self._lib = _lib.load()
        self.cfg = cfg or default_config()
        if handle is not None:
            self._h = C.c_void_p(handle)
            return
        h = C.c_void_p()
        rc = self._lib.cqs_hip_embedder_create(C.byref(self.cfg), device, C.byref(h))
        if rc != _lib.OK:
            raise HipError(rc, "cqs_hip_embedder_create failed")
        self._h = h

    @classmethod
    def load_dir(cls, model_dir: str, cfg: Optional[_lib.EmbedConfig] = None, device: int = 0) -> "HipEmbedEngine":
        lib = _lib.load()
        cfg = cfg or default_config()
        h = C.c_void_p()
        rc = lib.cqs_hip_embedder_load_dir(model_dir.encode(), C.byref(cfg), device, C.byref(h))
        if rc != _lib.OK:
            raise HipError(rc, f"cqs_hip_embedder_load_dir({model_dir}) failed")
        return cls(cfg, device, handle=h.value)

    def set_tensor(self, name: str, data: np.ndarray) -> None:
        a = np.ascontiguousarray(data, dtype=np.float32)
        rc = self._lib.cqs_hip_embedder_set_tensor(self._h, name.encode(), a.ctypes.data_as(C.c_void_p), a.size)
        if rc != _lib.OK:
            raise HipError(rc, self.last_error())

    def set_weights(self, weights: dict) -> None:
        for k, v in weights.items():
            self.set_tensor(k, v)
        rc = self._lib.cqs_hip_embedder_finalize(self._h)
        if rc != _lib.OK:
            raise HipError(rc, self.last_error())

    def last_error(self) -> str:
        buf = C.create_string_buffer(512)
        self._lib.cqs_hip_embedder_last_error(self._h, buf, 512)
        return buf.value.decode("utf-8", "replace")

    def dim(self) -> int:
        return int(self._lib.cqs_hip_embedder_dim(self._h))

    def max_seq(self) -> int:
        return int(self._lib.cqs_hip_embedder_max_seq(self._h))

    def last_ms(self) -> float:
        return float(self._lib.cqs_hip_embedder_last_ms(self._h))

    def warm(self, max_tokens: int = 128) -> None:
        """`cqs_hip_embedder_warm`: build and replay the search-time chain's graphs for every length 1..max_tokens."""
        rc = self._lib.cqs_hip_embedder_warm(self._h, int(max_tokens))
        if rc != _lib.OK:
            raise HipError(rc, self.last_error())

    def query_graph_stats(self) -> dict:
        """`cqs_hip_embedder_query_graph_stats`: {captured, failed, replays, eager} of the search-time chain."""
        v = [C.c_uint64() for _ in range(4)]
        self._lib.cqs_hip_embedder_query_graph_stats(self._h, *[C.byref(x) for x in v])
        return dict(zip(("captured", "failed", "replays", "eager"), (int(x.value) for x in v)))

    def run(self, input_ids: np.ndarray, attention_mask: np.ndarray) -> np.ndarray:
        """`session.run`: i64 [B, L] x2 -> f32 [B, dim] (`sentence_embedding`, not normalised)."""
        ids = np.ascontiguousarray(input_ids, dtype=np.int64)
        mask = np.ascontiguousarray(attention_mask, dtype=np.int64)
        if ids.shape != mask.shape or ids.ndim != 2:
            raise EmbedderError("InferenceFailed: input_ids / attention_mask shape mismatch")
        out = np.zeros((ids.shape[0], self.dim()), dtype=np.float32)
        rc = self._lib.cqs_hip_embed(self._h, ids.ctypes.data_as(C.c_void_p), mask.ctypes.data_as(C.c_void_p),
                                     ids.shape[0], ids.shape[1], out.ctypes.data_as(C.c_void_p))
        if rc != _lib.OK:
            raise EmbedderError(f"InferenceFailed: {self.last_error()} ({rc})")
        return out

    def submit(self, input_ids: np.ndarray, attention_mask: np.ndarray) -> int:
        """`cqs_hip_embed_submit`: enqueue a padded [B, L] batch, return its ticket (no device wait)."""
        ids = np.ascontiguousarray(input_ids, dtype=np.int64)
        mask = np.ascontiguousarray(attention_mask, dtype=np.int64)
        if ids.shape != mask.shape or ids.ndim != 2:
            raise EmbedderError("InferenceFailed: input_ids / attention_mask shape mismatch")
        t = C.c_uint64()
        rc = self._lib.cqs_hip_embed_submit(self._h, ids.ctypes.data_as(C.c_void_p), mask.ctypes.data_as(C.c_void_p),
                                            ids.shape[0], ids.shape[1], C.byref(t))
        if rc != _lib.OK:
            raise EmbedderError(f"InferenceFailed: {self.last_error()} ({rc})")
        return int(t.value)

    def submit_ragged(self, tokens: np.ndarray, lens: np.ndarray) -> int:
        """`cqs_hip_embed_submit_ragged`: sequences back to back (i32) + their lengths (u32)."""
        tok = np.ascontiguousarray(tokens, dtype=np.int32)
        ln = np.ascontiguousarray(lens, dtype=np.uint32)
        if int(ln.sum()) != tok.size:
            raise EmbedderError("InferenceFailed: lens do not add up to the token count")
        t = C.c_uint64()
        rc = self._lib.cqs_hip_embed_submit_ragged(self._h, tok.ctypes.data_as(C.c_void_p), ln.ctypes.data_as(C.c_void_p),
                                                   ln.size, C.byref(t))
        if rc != _lib.OK:
            raise EmbedderError(f"InferenceFailed: {self.last_error()} ({rc})")
        return int(t.value)

    def collect(self, ticket: int, batch: int, out: Optional[np.ndarray] = None) -> np.ndarray:
        """`cqs_hip_embed_collect`: wait for `ticket`, return its f32 [batch, dim] rows (not normalised)."""
        if out is None:
            out = np.empty((batch, self.dim()), dtype=np.float32)
        assert out.flags["C_CONTIGUOUS"] and out.dtype == np.float32 and out.shape == (batch, self.dim())
        rc = self._lib.cqs_hip_embed_collect(self._h, ticket, out.ctypes.data_as(C.c_void_p))
        if rc != _lib.OK:
            raise EmbedderError(f"InferenceFailed: {self.last_error()} ({rc})")
        return out

    def abandon(self, ticket: int) -> None:
        """`cqs_hip_embed_collect(ticket, NULL)`: wait for the ticket, release its submission slot, drop the rows."""
        rc = self._lib.cqs_hip_embed_collect(self._h, ticket, None)
        if rc != _lib.OK:
            raise EmbedderError(f"InferenceFailed: {self.last_error()} ({rc})")

    def run_hidden(self, input_ids: np.ndarray, attention_mask: np.ndarray) -> np.ndarray:
        ids = np.ascontiguousarray(input_ids, dtype=np.int64)
        mask = np.ascontiguousarray(attention_mask, dtype=np.int64)
        out = np.zeros((ids.shape[0], ids.shape[1], self.dim()), dtype=np.float32)
        rc = self._lib.cqs_hip_embed_hidden(self._h, ids.ctypes.data_as(C.c_void_p), mask.ctypes.data_as(C.c_void_p),
                                            ids.shape[0], ids.shape[1], out.ctypes.data_as(C.c_void_p))
        if rc != _lib.OK:
            raise EmbedderError(f"InferenceFailed: {self.last_error()} ({rc})")
        return out

    def close(self) -> None:
        if self._h:
            self._lib.cqs_hip_embedder_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Embedder:
    """`Embedder` (core.rs:34): documents / queries -> unit-norm 768-d vectors.

    tokenizer: callable(list[str]) -> list[list[int]] with special tokens already added
    (`encode_batch(inputs, add_special_tokens=true)`, core.rs:1012).
    """

    def __init__(self, engine: HipEmbedEngine, tokenizer: Callable[[List[str]], List[List[int]]],
                 pad_id: int = 0, query_cache_size: int = 1024, batch_size: Optional[int] = None):
        self.engine = engine
        self.tokenizer = tokenizer
        self.pad_id = pad_id
        self.max_seq_length = engine.max_seq()
        self.batch = batch_size or embed_batch_size(engine.dim(), self.max_seq_length)
        self._cache: "OrderedDict[str, np.ndarray]" = OrderedDict()
        self._cache_cap = query_cache_size

    def embedding_dim(self) -> int:  # core.rs:961
        return self.engine.dim()

    def embed_batch(self, texts: List[str]) -> List[np.ndarray]:
        """core.rs:994-1273: tokenize -> pad to min(longest, max_seq) -> run -> normalize_l2 per row."""
        if not texts:
            return []
        enc = self.tokenizer(list(texts))
        max_len = min(max(len(e) for e in enc), self.max_seq_length)   # core.rs:1020-1025
        max_len = max(max_len, 1)
        ids = pad_2d_i64(enc, max_len, self.pad_id)
        mask = pad_2d_i64([[1] * len(e) for e in enc], max_len, 0)
        out = self.engine.run(ids, mask)
        if out.shape != (len(texts), self.embedding_dim()):
            raise EmbedderError("InferenceFailed: unexpected output shape")
        return [normalize_l2(r) for r in out]

    def embed_documents(self, texts: List[str]) -> List[np.ndarray]:
        """core.rs:718-751: doc prefix, chunks of embed_batch_size(), order preserved."""
        pref = [DOC_PREFIX + t for t in texts]
        out: List[np.ndarray] = []
        for i in range(0, len(pref), self.batch):
            out.extend(self.embed_batch(pref[i:i + self.batch]))
        return out

    def embed_query(self, text: str) -> np.ndarray:
        """core.rs:768-856: trim, EmptyQuery, truncate to 32 KiB at a char boundary, LRU cache, query prefix."""
        t = text.strip()
        if not t:
            raise EmbedderError("EmptyQuery")
        b = t.encode("utf-8")
        if len(b) > MAX_QUERY_BYTES:
            t = b[:MAX_QUERY_BYTES].decode("utf-8", "ignore")
        hit = self._cache.get(t)
        if hit is not None:
            self._cache.move_to_end(t)
            return hit
        v = self.embed_batch([QUERY_PREFIX + t])[0]
        self._cache[t] = v
        if len(self._cache) > self._cache_cap:
            self._cache.popitem(last=False)
        return v

    def warm(self, max_tokens: int = 128) -> None:  # core.rs:933-957
        """One dummy inference like the reference - plus the search-time chain's graphs for every query length up to
        `max_tokens`, which would otherwise be built on the first query of each length."""
        self.engine.warm(max_tokens)
        self.embed_query("warmup")
