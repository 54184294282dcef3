"""Host-side mirror of `SpladeIndex` (src/splade/index.rs:177-306) over the C ABI's sparse index, and of the hybrid
fusion that consumes it (`search_hybrid_inner`, src/search/query.rs:898-1010).

Same names, argument meaning and failure behaviour as the reference: `build` takes `(chunk_id, sparse_vector)` pairs,
`search` / `search_with_filter` return `IndexResult`s best first and never raise for device trouble (a failed search is
an empty list + `last_error`, like a backend that logs and falls through).  No CPU fallback: without libcqs_hip.so the
import of the library raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .index import HipError, IndexResult

SparseVector = Sequence[Tuple[int, float]]      # `pub type SparseVector = Vec<(u32, f32)>` (src/splade/mod.rs)


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def forward_csr(vectors: Sequence[SparseVector]):
    """The chunks' sparse vectors as the flat arrays `cqs_hip_sparse_index_create` takes."""
    off = np.zeros(len(vectors) + 1, dtype=np.uint64)
    for i, v in enumerate(vectors):
        off[i + 1] = off[i] + len(v)
    total = int(off[-1])
    tok = np.zeros(total, dtype=np.uint32)
    w = np.zeros(total, dtype=np.float32)
    at = 0
    for v in vectors:
        for t, x in v:
            tok[at] = t
            w[at] = x
            at += 1
    return off, tok, w


def id_ranks(ids: Sequence[str]) -> np.ndarray:
    """rank of every chunk id in ascending byte order (Rust `String` ordering), equal ids in chunk order: the order
    `BoundedScoreHeap` breaks score ties in (src/search/scoring/candidate.rs:299-334)."""
    enc = [s.encode("utf-8") for s in ids]
    order = sorted(range(len(enc)), key=lambda i: (enc[i], i))
    rank = np.zeros(len(enc), dtype=np.uint32)
    for r, i in enumerate(order):
        rank[i] = r
    return rank


class HipSpladeIndex:
    """`SpladeIndex`: in-HBM inverted index for SPLADE sparse vectors."""

    def __init__(self, handle: int, id_map: Optional[List[str]]):
        self._lib = _lib.load()
        self._h = C.c_void_p(handle)
        self.id_map = id_map
        self.last_error = ""

    # ---- construction ---------------------------------------------------------
    @classmethod
    def build(cls, chunks: Sequence[Tuple[str, SparseVector]], device: int = 0) -> "HipSpladeIndex":
        """`SpladeIndex::build(chunks: Vec<(String, SparseVector)>)` (index.rs:191-212)."""
        ids = [cid for cid, _v in chunks]
        off, tok, w = forward_csr([v for _cid, v in chunks])
        return cls.build_from_csr(ids, off, tok, w, device=device)

    @classmethod
    def build_from_csr(cls, id_map: Optional[List[str]], doc_off: np.ndarray, tokens: np.ndarray, weights: np.ndarray,
                       id_rank: Optional[np.ndarray] = None, device: int = 0) -> "HipSpladeIndex":
        """From flat arrays (what `cqs_hip_splade_encode_sparse` hands back, batch after batch).  `id_map=None`: the
        chunk index is the id (ties by chunk index); else ties follow the ids' byte order unless `id_rank` is given."""
        lib = _lib.load()
        doc_off = np.ascontiguousarray(doc_off, dtype=np.uint64)
        tokens = np.ascontiguousarray(tokens, dtype=np.uint32)
        weights = np.ascontiguousarray(weights, dtype=np.float32)
        n = doc_off.size - 1
        if n < 0 or tokens.size != weights.size or (n >= 0 and int(doc_off[-1]) != tokens.size):
            raise ValueError("doc_off / tokens / weights disagree")
        if id_map is not None and len(id_map) != n:
            raise ValueError("id_map length != chunks")
        if id_rank is None and id_map is not None:
            id_rank = id_ranks(id_map)
        if id_rank is not None:
            id_rank = np.ascontiguousarray(id_rank, dtype=np.uint32)
        h = C.c_void_p()
        rc = lib.cqs_hip_sparse_index_create(_ptr(doc_off), _ptr(tokens), _ptr(weights), n, _ptr(id_rank), device, C.byref(h))
        if rc != _lib.OK:
            raise HipError(rc, "cqs_hip_sparse_index_create failed")
        return cls(h.value, None if id_map is None else list(id_map))

    @classmethod
    def build_from_postings(cls, id_map: Optional[List[str]], postings, n_chunks: Optional[int] = None,
                            id_rank: Optional[np.ndarray] = None, device: int = 0) -> "HipSpladeIndex":
        """From the reference's in-memory form, `postings: HashMap<u32, Vec<(usize, f32)>>` (index.rs:177-187; what
        `SpladeIndex::load` reconstructs from the persisted file): a mapping token -> [(chunk_index, weight), ...]."""
        lib = _lib.load()
        n = len(id_map) if id_map is not None else int(n_chunks)
        keys = list(postings.keys())
        off = np.zeros(len(keys) + 1, dtype=np.uint64)
        for i, t in enumerate(keys):
            off[i + 1] = off[i] + len(postings[t])
        total = int(off[-1])
        ch = np.zeros(total, dtype=np.uint32)
        w = np.zeros(total, dtype=np.float32)
        at = 0
        for t in keys:
            for c, x in postings[t]:
                ch[at] = c
                w[at] = x
                at += 1
        tok = np.asarray(keys, dtype=np.uint32)
        if id_rank is None and id_map is not None:
            id_rank = id_ranks(id_map)
        if id_rank is not None:
            id_rank = np.ascontiguousarray(id_rank, dtype=np.uint32)
        h = C.c_void_p()
        rc = lib.cqs_hip_sparse_index_create_inverted(_ptr(tok), _ptr(off), _ptr(ch), _ptr(w), len(keys), n, _ptr(id_rank), device,
                                                      C.byref(h))
        if rc != _lib.OK:
            raise HipError(rc, "cqs_hip_sparse_index_create_inverted failed")
        return cls(h.value, None if id_map is None else list(id_map))

    # ---- persistence (SpladeIndex::save / load / load_or_build, index.rs:346-1107) -------------
    def save(self, path: str, generation: int) -> int:
        """-> the content checksum.  Raises HipError when the file cannot be written."""
        ck = C.c_uint64()
        rc = self._lib.cqs_hip_sparse_index_save(self._h, str(path).encode(), int(generation), C.byref(ck))
        if rc != _lib.OK:
            raise HipError(rc, "cqs_hip_sparse_index_save failed")
        return int(ck.value)

    @classmethod
    def load(cls, path: str, generation: int, id_map: Optional[List[str]] = None, device: int = 0) -> Optional["HipSpladeIndex"]:
        """`Ok(None)` of the reference = None here: missing file, another generation, anything damaged - the caller rebuilds."""
        lib = _lib.load()
        h = C.c_void_p()
        rc = lib.cqs_hip_sparse_index_load(str(path).encode(), 0 if id_map is None else len(id_map), int(generation), device, C.byref(h))
        if rc != _lib.OK or not h.value:
            return None
        return cls(h.value, None if id_map is None else list(id_map))

    @classmethod
    def load_or_build(cls, path: str, generation: int, rows: Callable[[], Sequence[Tuple[str, SparseVector]]],
                      device: int = 0) -> Tuple["HipSpladeIndex", bool]:
        """`SpladeIndex::load_or_build(path, generation, rows)` (index.rs:1073-1107) -> (index, rebuilt): the persisted
        file when it is current, else built from `rows()` and persisted (a failed save is logged, not fatal).  The chunk ids
        are not in the file: `rows()` is only called for a rebuild, so a loaded index takes its id_map from the caller's
        `ids` attribute when `rows` carries one (`rows.ids`), else stays integer-addressed."""
        ids = getattr(rows, "ids", None)
        got = cls.load(path, generation, ids, device)
        if got is not None:
            return got, False
        idx = cls.build(list(rows()), device)
        try:
            idx.save(path, generation)
        except HipError as e:           # "SPLADE index persist failed, continuing with in-memory index only" (index.rs:1097-1104)
            import logging
            logging.getLogger("cqs.hip").warning("sparse index persist failed: %s", e)
        return idx, True

    # ---- properties -----------------------------------------------------------
    def __len__(self) -> int:
        return int(self._lib.cqs_hip_sparse_index_len(self._h))

    def is_empty(self) -> bool:
        return len(self) == 0

    def unique_tokens(self) -> int:
        return int(self._lib.cqs_hip_sparse_index_unique_tokens(self._h))

    def postings(self) -> int:
        return int(self._lib.cqs_hip_sparse_index_postings(self._h))

    def combine_stats(self) -> Tuple[int, int]:
        """(batches the combining queue ran, queries they carried) since the handle was made"""
        p, q = C.c_uint64(), C.c_uint64()
        self._lib.cqs_hip_sparse_index_combine_stats(self._h, C.byref(p), C.byref(q))
        return int(p.value), int(q.value)

    def last_search(self) -> Tuple[float, int]:
        """(device ms of the last accumulate launch, postings it read)"""
        ms, touched = C.c_float(), C.c_uint64()
        self._lib.cqs_hip_sparse_index_last_search(self._h, C.byref(ms), C.byref(touched))
        return float(ms.value), int(touched.value)

    # ---- search ---------------------------------------------------------------
    def search_raw(self, q_tokens, q_weights, k: int, keep: Optional[np.ndarray] = None):
        """-> (chunk indices u64, scores f32, status).  keep: bool / 0-1 per chunk, or None."""
        qt = np.ascontiguousarray(q_tokens, dtype=np.uint32)
        qw = np.ascontiguousarray(q_weights, dtype=np.float32)
        if qt.size != qw.size:
            raise ValueError("query tokens / weights disagree")
        bits = None
        if keep is not None:
            keep = np.asarray(keep).astype(bool)
            if keep.size != len(self):
                raise ValueError("keep length != chunks")
            packed = np.packbits(keep, bitorder="little")
            bits = np.zeros((keep.size + 31) // 32 * 4, dtype=np.uint8)
            bits[:packed.size] = packed
            bits = bits.view(np.uint32)
        out = np.zeros(max(k, 1), dtype=np.uint64)
        sc = np.zeros(max(k, 1), dtype=np.float32)
        cnt = C.c_uint32()
        rc = self._lib.cqs_hip_sparse_index_search(self._h, _ptr(qt), _ptr(qw), qt.size, k, _ptr(bits), _ptr(out), _ptr(sc),
                                                  C.byref(cnt))
        if rc != _lib.OK:
            buf = C.create_string_buffer(512)
            self._lib.cqs_hip_sparse_index_last_error(self._h, buf, 512)
            self.last_error = buf.value.decode("utf-8", "replace")
            return out[:0], sc[:0], rc
        return out[:cnt.value], sc[:cnt.value], rc

    def search_batch_raw(self, queries, k: int, keep: Optional[np.ndarray] = None):
        """`cqs_hip_sparse_index_search_batch`: queries = [(tokens, weights), ...] (at most 64).
        -> (chunks u64 [b, k], scores f32 [b, k], counts u32 [b], status)."""
        b = len(queries)
        off = np.zeros(b + 1, dtype=np.uint64)
        for i, (t, _w) in enumerate(queries):
            off[i + 1] = off[i] + len(t)
        qt = np.concatenate([np.asarray(t, dtype=np.uint32) for t, _w in queries]) if b else np.zeros(0, np.uint32)
        qw = np.concatenate([np.asarray(w, dtype=np.float32) for _t, w in queries]) if b else np.zeros(0, np.float32)
        bits = None
        if keep is not None:
            keep = np.asarray(keep).astype(bool)
            packed = np.packbits(keep, bitorder="little")
            bits = np.zeros((keep.size + 31) // 32 * 4, dtype=np.uint8)
            bits[:packed.size] = packed
            bits = bits.view(np.uint32)
        out = np.zeros((max(b, 1), max(k, 1)), dtype=np.uint64)
        sc = np.zeros((max(b, 1), max(k, 1)), dtype=np.float32)
        cnt = np.zeros(max(b, 1), dtype=np.uint32)
        rc = self._lib.cqs_hip_sparse_index_search_batch(self._h, _ptr(off), _ptr(qt), _ptr(qw), b, k, _ptr(bits), _ptr(out), _ptr(sc),
                                                        _ptr(cnt))
        if rc != _lib.OK:
            buf = C.create_string_buffer(512)
            self._lib.cqs_hip_sparse_index_last_error(self._h, buf, 512)
            self.last_error = buf.value.decode("utf-8", "replace")
        return out[:b, :k], sc[:b, :k], cnt[:b], rc

    def search(self, query: SparseVector, k: int) -> List[IndexResult]:
        """`search(&self, query, k)` (index.rs:214-216)."""
        return self.search_with_filter(query, k, None)

    def search_with_filter(self, query: SparseVector, k: int, filter: Optional[Callable[[str], bool]]) -> List[IndexResult]:
        """`search_with_filter(&self, query, k, filter: &dyn Fn(&str) -> bool)` (index.rs:223-290)."""
        keep = None
        if filter is not None:
            n = len(self)
            keep = np.fromiter((bool(filter(self._id(i))) for i in range(n)), dtype=bool, count=n)
        ch, sc, _rc = self.search_raw([t for t, _w in query], [w for _t, w in query], k, keep)
        return [IndexResult(self._id(int(c)), float(s)) for c, s in zip(ch, sc)]

    def _id(self, i: int) -> str:
        return self.id_map[i] if self.id_map is not None else str(i)

    # ---- lifetime -------------------------------------------------------------
    def close(self) -> None:
        if self._h:
            self._lib.cqs_hip_sparse_index_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def fuse_hybrid(dense_results: Sequence[IndexResult], sparse_results: Sequence[IndexResult], alpha: float,
                candidate_count: int) -> List[IndexResult]:
    """The fusion of `search_hybrid_inner` (src/search/query.rs:909-1010): sparse scores min-max normalised by the sparse
    pool's maximum (reduce-from-first; a non-positive maximum zeroes the leg), candidates = dense ids then sparse-only ids
    in first-seen order, score = alpha * dense + (1 - alpha) * sparse (alpha <= 0: dense + 0.1 * sparse), sorted by
    (score desc total order, id asc), truncated to candidate_count.  f32 arithmetic throughout."""
    f = np.float32
    max_sparse = f(0.0)
    if len(sparse_results):
        max_sparse = f(sparse_results[0].score)
        for r in sparse_results[1:]:
            s = f(r.score)
            max_sparse = s if (np.isnan(max_sparse) or s > max_sparse) else max_sparse   # f32::max: NaN loses
    dense_scores = {}
    for r in dense_results:
        dense_scores[r.id] = f(r.score)                    # HashMap::insert: the last one wins
    sparse_scores = {}
    for r in sparse_results:
        sparse_scores[r.id] = f(r.score) / max_sparse if max_sparse > 0 else f(0.0)
    all_ids, seen = [], set()
    for r in list(dense_results) + list(sparse_results):
        if r.id not in seen:
            seen.add(r.id)
            all_ids.append(r.id)
    a = f(alpha)
    fused = []
    for cid in all_ids:
        d = dense_scores.get(cid, f(0.0))
        s = sparse_scores.get(cid, f(0.0))
        score = d + s * f(0.1) if a <= 0 else a * d + (f(1.0) - a) * s
        fused.append(IndexResult(cid, float(f(score))))

    def total_key(x: float) -> int:                          # f32::total_cmp as an integer key
        b = int(np.float32(x).view(np.int32))
        return b ^ ((b >> 31) & 0x7FFFFFFF)

    fused.sort(key=lambda r: (-total_key(r.score), r.id.encode("utf-8")))
    return fused[:candidate_count]
