"""ctypes wrapper of oracle/libcqs_oracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module, and only as the checker / reported CPU baseline.  See cqs_oracle.h for the
reference file:line each function restates.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcqs_oracle.so")
_lib = None

DOT_SIMSIMD, DOT_F64, DOT_SEQ, DOT_NATIVE = 0, 1, 2, 3


def build() -> None:
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        vp, sz, f, d, i = C.c_void_p, C.c_size_t, C.c_float, C.c_double, C.c_int
        sig = {
            "cqs_oracle_dot_simsimd": (d, [vp, vp, sz]),
            "cqs_oracle_dot_simsimd_native": (d, [vp, vp, sz]),
            "cqs_oracle_dot_isa": (C.c_char_p, [i]),
            "cqs_oracle_dot_force_scalar": (None, [i]),
            "cqs_oracle_dot_f64": (d, [vp, vp, sz]),
            "cqs_oracle_dot_seq_f32": (f, [vp, vp, sz]),
            "cqs_oracle_cosine_similarity": (i, [vp, sz, vp, sz, vp]),
            "cqs_oracle_full_cosine_similarity": (i, [vp, sz, vp, sz, vp]),
            "cqs_oracle_normalize_l2": (None, [vp, sz]),
            "cqs_oracle_bytes_to_embedding": (i, [vp, sz, sz, vp]),
            "cqs_oracle_apply_scoring_default": (i, [f, f, vp]),
            "cqs_oracle_heap_new": (vp, [sz]),
            "cqs_oracle_heap_free": (None, [vp]),
            "cqs_oracle_heap_would_accept": (i, [vp, f]),
            "cqs_oracle_heap_push_u64": (None, [vp, C.c_uint64, f]),
            "cqs_oracle_heap_push_str": (None, [vp, C.c_char_p, f]),
            "cqs_oracle_heap_len": (sz, [vp]),
            "cqs_oracle_heap_into_sorted": (sz, [vp, vp, vp, sz]),
            "cqs_oracle_brute_force": (sz, [vp, sz, sz, vp, sz, sz, f, i, vp, vp]),
            "cqs_oracle_find_neighbors": (sz, [vp, sz, sz, sz, sz, vp, vp]),
            "cqs_oracle_index_search": (sz, [vp, sz, sz, vp, sz, sz, vp, i, f, i, vp, vp]),
            "cqs_oracle_cagra_cosine_from_l2sq": (f, [f]),
            "cqs_oracle_dist_dot_clamped": (f, [vp, vp, sz]),
            "cqs_oracle_prepare_index_keep": (sz, [vp, sz, sz, vp]),
            "cqs_oracle_dim_scaled_batch": (sz, [sz, sz, sz, sz]),
            "cqs_oracle_candidate_count_for": (sz, [sz, sz]),
            "cqs_oracle_embed_batch_size": (sz, [sz, sz]),
            "cqs_oracle_mean_pool": (None, [vp, vp, sz, sz, sz, vp]),
            "cqs_oracle_cls_pool": (None, [vp, sz, sz, sz, vp]),
            "cqs_oracle_last_token_pool": (None, [vp, vp, sz, sz, sz, vp]),
            "cqs_oracle_brute_force_mt": (sz, [vp, sz, sz, vp, sz, f, i, i, vp, vp]),
            "cqs_oracle_set_worker_cpus": (None, [vp, i]),
            "cqs_oracle_first_touch_copy": (None, [vp, vp, sz, sz, i]),
            "cqs_oracle_splade_build": (vp, [vp, vp, vp, C.c_uint64]),
            "cqs_oracle_splade_free": (None, [vp]),
            "cqs_oracle_splade_len": (sz, [vp]),
            "cqs_oracle_splade_unique_tokens": (sz, [vp]),
            "cqs_oracle_splade_search": (sz, [vp, vp, vp, sz, sz, vp, vp, vp, vp, vp]),
            "cqs_oracle_splade_touched": (C.c_uint64, [vp, vp, sz]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def dot(a, b, kind=DOT_SIMSIMD) -> float:
    a, b = _f32(a), _f32(b)
    L = lib()
    if kind == DOT_F64:
        return L.cqs_oracle_dot_f64(_p(a), _p(b), a.size)
    if kind == DOT_SEQ:
        return L.cqs_oracle_dot_seq_f32(_p(a), _p(b), a.size)
    if kind == DOT_NATIVE:
        return L.cqs_oracle_dot_simsimd_native(_p(a), _p(b), a.size)
    return L.cqs_oracle_dot_simsimd(_p(a), _p(b), a.size)


def cosine_similarity(a, b):
    a, b = _f32(a), _f32(b)
    out = C.c_float()
    ok = lib().cqs_oracle_cosine_similarity(_p(a), a.size, _p(b), b.size, C.byref(out))
    return out.value if ok else None


def full_cosine_similarity(a, b):
    a, b = _f32(a), _f32(b)
    out = C.c_float()
    ok = lib().cqs_oracle_full_cosine_similarity(_p(a), a.size, _p(b), b.size, C.byref(out))
    return out.value if ok else None


def normalize_l2(v) -> np.ndarray:
    v = _f32(v).copy()
    lib().cqs_oracle_normalize_l2(_p(v), v.size)
    return v


def bytes_to_embedding(blob: bytes, dim: int):
    out = np.zeros((dim,), dtype=np.float32)
    buf = np.frombuffer(blob, dtype=np.uint8)
    rc = lib().cqs_oracle_bytes_to_embedding(_p(buf) if len(blob) else None, len(blob), dim, _p(out))
    return out if rc == 0 else None


def apply_scoring_default(score: float, threshold: float):
    out = C.c_float()
    ok = lib().cqs_oracle_apply_scoring_default(score, threshold, C.byref(out))
    return out.value if ok else None


class BoundedScoreHeap:
    """candidate.rs:162-330 via the C restatement (string ids)."""

    def __init__(self, capacity: int):
        self._L = lib()
        self._h = C.c_void_p(self._L.cqs_oracle_heap_new(capacity))
        self._ids = []
        self._cap = capacity

    def would_accept(self, score: float) -> bool:
        return bool(self._L.cqs_oracle_heap_would_accept(self._h, score))

    def push(self, sid: str, score: float) -> None:
        self._ids.append(sid)
        self._L.cqs_oracle_heap_push_str(self._h, sid.encode(), score)

    def into_sorted_vec(self):
        n = self._L.cqs_oracle_heap_len(self._h)
        ids = np.zeros((max(n, 1),), dtype=np.uint64)
        sc = np.zeros((max(n, 1),), dtype=np.float32)
        c = self._L.cqs_oracle_heap_into_sorted(self._h, _p(ids), _p(sc), n)
        return [(self._ids[int(ids[i])], float(sc[i])) for i in range(c)]

    def __del__(self):
        try:
            self._L.cqs_oracle_heap_free(self._h)
        except Exception:
            pass


def brute_force(rows, query, limit, threshold=0.0, kind=DOT_SIMSIMD):
    """search_filtered_with_notes minus SQLite (search/query.rs:348-510).  -> (ids u64, scores f32)."""
    rows, query = _f32(rows), _f32(query)
    n, dim = rows.shape
    ids = np.zeros((max(limit, 1),), dtype=np.uint64)
    sc = np.zeros((max(limit, 1),), dtype=np.float32)
    c = lib().cqs_oracle_brute_force(_p(rows), n, dim, _p(query), query.size, limit, threshold, kind, _p(ids), _p(sc))
    return ids[:c], sc[:c]


def brute_force_mt(rows, query, limit, threshold, threads, kind=DOT_SIMSIMD):
    rows, query = _f32(rows), _f32(query)
    n, dim = rows.shape
    ids = np.zeros((max(limit, 1),), dtype=np.uint64)
    sc = np.zeros((max(limit, 1),), dtype=np.float32)
    c = lib().cqs_oracle_brute_force_mt(_p(rows), n, dim, _p(query), limit, threshold, threads, kind, _p(ids), _p(sc))
    return ids[:c], sc[:c]


def set_worker_cpus(cpus) -> None:
    """Pin worker t of the multi-thread scan / first-touch copy to cpus[t % len(cpus)]; [] = the OS places threads."""
    arr = np.ascontiguousarray(list(cpus), dtype=np.int32)
    lib().cqs_oracle_set_worker_cpus(_p(arr) if len(arr) else None, len(arr))


def first_touch_copy(src, threads):
    """A copy of `src` whose shard t (the row partition of brute_force_mt with `threads` workers) was first touched by
    worker t - with pinned workers each shard then sits on its scanner's NUMA node."""
    src = _f32(src)
    dst = np.empty_like(src)                       # fresh mapping: pages are placed when the workers write them
    lib().cqs_oracle_first_touch_copy(_p(dst), _p(src), src.shape[0], src.shape[1], threads)
    return dst


def dot_isa(native: bool = False) -> str:
    """ISA of the dot body in use on this host ("scalar-fmaf" / "avx2+fma" / "avx512f")."""
    return lib().cqs_oracle_dot_isa(1 if native else 0).decode()


def dot_force_scalar(on: bool) -> None:
    lib().cqs_oracle_dot_force_scalar(1 if on else 0)


def find_neighbors(rows, target_row, limit):
    rows = _f32(rows)
    n, dim = rows.shape
    ids = np.zeros((100,), dtype=np.uint64)
    sc = np.zeros((100,), dtype=np.float32)
    c = lib().cqs_oracle_find_neighbors(_p(rows), n, dim, target_row, limit, _p(ids), _p(sc))
    return ids[:c], sc[:c]


def index_search(rows, query, k, keep_bitset=None, mode=0, threshold=0.0, kind=DOT_SIMSIMD):
    """Exact VectorIndex::search contract (cagra.rs guards, raw dot, (score desc,row asc))."""
    rows, query = _f32(rows), _f32(query)
    n = rows.shape[0]
    dim = rows.shape[1] if rows.ndim == 2 else 0
    ids = np.zeros((max(k, 1),), dtype=np.uint64)
    sc = np.zeros((max(k, 1),), dtype=np.float32)
    kb = None if keep_bitset is None else np.ascontiguousarray(keep_bitset, dtype=np.uint32)
    c = lib().cqs_oracle_index_search(_p(rows), n, dim, _p(query), query.size, k, _p(kb), mode, threshold, kind,
                                      _p(ids), _p(sc))
    return ids[:c], sc[:c]


def prepare_index_keep(rows):
    rows = _f32(rows)
    keep = np.zeros((rows.shape[0],), dtype=np.uint8)
    kept = lib().cqs_oracle_prepare_index_keep(_p(rows), rows.shape[0], rows.shape[1], _p(keep))
    return keep.astype(bool), kept


def mean_pool(hidden, mask):
    hidden = _f32(hidden)
    mask = np.ascontiguousarray(mask, dtype=np.int64)
    b, s, d = hidden.shape
    out = np.zeros((b, d), dtype=np.float32)
    lib().cqs_oracle_mean_pool(_p(hidden), _p(mask), b, s, d, _p(out))
    return out


def cls_pool(hidden):
    hidden = _f32(hidden)
    b, s, d = hidden.shape
    out = np.zeros((b, d), dtype=np.float32)
    lib().cqs_oracle_cls_pool(_p(hidden), b, s, d, _p(out))
    return out


def last_token_pool(hidden, mask):
    hidden = _f32(hidden)
    mask = np.ascontiguousarray(mask, dtype=np.int64)
    b, s, d = hidden.shape
    out = np.zeros((b, d), dtype=np.float32)
    lib().cqs_oracle_last_token_pool(_p(hidden), _p(mask), b, s, d, _p(out))
    return out


def forward_csr(docs):
    """[(token, weight), ...] per chunk -> (doc_off u64 [n + 1], tokens u32, weights f32): the documents' sparse vectors
    (`SparseVector = Vec<(u32, f32)>`, src/splade/mod.rs) as the flat arrays both the oracle and the C ABI take."""
    off = np.zeros(len(docs) + 1, dtype=np.uint64)
    for i, d in enumerate(docs):
        off[i + 1] = off[i] + len(d)
    tok = np.zeros(int(off[-1]), dtype=np.uint32)
    w = np.zeros(int(off[-1]), dtype=np.float32)
    at = 0
    for d in docs:
        for t, x in d:
            tok[at] = t
            w[at] = x
            at += 1
    return off, tok, w


class SpladeIndex:
    """`SpladeIndex` (src/splade/index.rs:177-306): build / search / search_with_filter / len / is_empty /
    unique_tokens, over the C restatement.  `ids`: chunk id strings (None = the chunk index is the id)."""

    def __init__(self, doc_off, tokens, weights, ids=None, id_rank=None):
        self._off = np.ascontiguousarray(doc_off, dtype=np.uint64)
        self._tok = np.ascontiguousarray(tokens, dtype=np.uint32)
        self._w = np.ascontiguousarray(weights, dtype=np.float32)
        self.n = len(self._off) - 1
        self.ids = None if ids is None else list(ids)
        self._rank = None if id_rank is None else np.ascontiguousarray(id_rank, dtype=np.uint32)
        self._cids = None
        if self.ids is not None:
            enc = [s.encode("utf-8") for s in self.ids]
            self._cids = (C.c_char_p * len(enc))(*enc)
        self._h = C.c_void_p(lib().cqs_oracle_splade_build(_p(self._off), _p(self._tok), _p(self._w), self.n))

    @classmethod
    def build(cls, chunks):
        """`SpladeIndex::build(Vec<(String, SparseVector)>)` (index.rs:191-212)."""
        off, tok, w = forward_csr([sv for _id, sv in chunks])
        return cls(off, tok, w, ids=[cid for cid, _sv in chunks])

    def __len__(self):
        return int(lib().cqs_oracle_splade_len(self._h))

    def is_empty(self):
        return len(self) == 0

    def unique_tokens(self):
        return int(lib().cqs_oracle_splade_unique_tokens(self._h))

    def touched(self, q_tokens):
        qt = np.ascontiguousarray(q_tokens, dtype=np.uint32)
        return int(lib().cqs_oracle_splade_touched(self._h, _p(qt), qt.size))

    def search_raw(self, q_tokens, q_weights, k, keep=None):
        """-> (chunk indices u64, scores f32), best first."""
        qt = np.ascontiguousarray(q_tokens, dtype=np.uint32)
        qw = np.ascontiguousarray(q_weights, dtype=np.float32)
        kp = None if keep is None else np.ascontiguousarray(keep, dtype=np.uint8)
        out = np.zeros(max(k, 1), dtype=np.uint64)
        sc = np.zeros(max(k, 1), dtype=np.float32)
        ids = None if self._cids is None else C.cast(self._cids, C.c_void_p)
        c = lib().cqs_oracle_splade_search(self._h, _p(qt), _p(qw), qt.size, k, _p(kp), ids, _p(self._rank), _p(out), _p(sc))
        return out[:c], sc[:c]

    def search(self, query, k):
        """`search(&SparseVector, k)` (index.rs:214-216) -> [(id, score)]."""
        return self.search_with_filter(query, k, None)

    def search_with_filter(self, query, k, pred):
        keep = None
        if pred is not None:
            keep = np.array([1 if pred(self.ids[i] if self.ids is not None else i) else 0 for i in range(self.n)], dtype=np.uint8)
        ch, sc = self.search_raw([t for t, _w in query], [w for _t, w in query], k, keep)
        return [((self.ids[int(c)] if self.ids is not None else int(c)), float(s)) for c, s in zip(ch, sc)]

    def __del__(self):
        try:
            if self._h:
                lib().cqs_oracle_splade_free(self._h)
                self._h = None
        except Exception:
            pass


def hybrid_fuse(dense, sparse, alpha, candidate_count):
    """`search_hybrid_inner`'s fusion (src/search/query.rs:909-1010) restated over plain (id, score) lists - written
    independently of the product's mirror (cqs_amd/splade_index.fuse_hybrid), as its checker.
    :917-921 max_sparse = reduce(f32::max) over the sparse pool (0.0 if empty); :936-950 dense map (last insert wins),
    sparse map = score / max_sparse when max_sparse > 0 else 0.0; :957-970 ids: dense first, then unseen sparse ones;
    :979-997 alpha <= 0 -> d + s * 0.1, else alpha * d + (1 - alpha) * s, absent = 0.0; :1003 sort by (score desc
    total_cmp, id asc); :1004 truncate."""
    f32 = np.float32
    scores = [f32(s) for _i, s in sparse]
    mx = f32(0.0)
    if scores:
        mx = scores[0]
        for s in scores[1:]:
            mx = s if (mx != mx or s > mx) else mx
    dmap, smap = {}, {}
    for i, s in dense:
        dmap[i] = f32(s)
    for i, s in sparse:
        smap[i] = (f32(s) / mx) if mx > 0 else f32(0.0)
    order = []
    for i, _s in list(dense) + list(sparse):
        if i not in order:
            order.append(i)
    a = f32(alpha)
    out = []
    for i in order:
        d, s = dmap.get(i, f32(0.0)), smap.get(i, f32(0.0))
        v = (d + s * f32(0.1)) if a <= 0 else (a * d + (f32(1.0) - a) * s)
        out.append((i, f32(v)))

    def key(t):
        bits = int(np.float32(t[1]).view(np.int32))
        bits ^= (bits >> 31) & 0x7FFFFFFF                 # total_cmp order as an integer
        return (-bits, t[0].encode("utf-8") if isinstance(t[0], str) else t[0])

    out.sort(key=key)
    return [(i, float(v)) for i, v in out[:candidate_count]]
