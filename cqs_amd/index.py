"""Host-side mirror of the reference's vector-index interface over the C ABI.

Reference surface mirrored here (names, argument meaning, error behaviour):
  DistanceMetric          src/index.rs:45-125
  IndexResult             src/index.rs:129-134
  VectorIndex (trait)     src/index.rs:139-239
  IndexBackend / context  src/index.rs:245-291, selector src/cli/store.rs:470-509
  CagraIndex (exemplar)   src/cagra.rs:255-277, search :443-492, filter :727-820,
                          build_from_flat :922-960, CagraBackend::try_open :1676-1802
  prepare_index_data      src/hnsw/mod.rs:688-746

The reference is Rust; with no Rust toolchain in the build image the host side
above the C ABI is written here (Python for the test/bench harness) and as a
Rust shim *source* in rust_shim/ (see INTEGRATION.md).  Every search goes through
libcqs_hip.so; there is no CPU fallback in this module.
"""
from __future__ import annotations

import ctypes as C
import enum
import logging
from dataclasses import dataclass
from typing import Callable, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib

log = logging.getLogger("cqs_amd")


class DistanceMetric(enum.Enum):
    """src/index.rs:45-56."""

    Cosine = 0
    DotProduct = 1

    def as_str(self) -> str:  # src/index.rs:61-66
        return "cosine" if self is DistanceMetric.Cosine else "dot"

    @staticmethod
    def parse(raw: str) -> "DistanceMetric":  # FromStr, src/index.rs:112-125
        s = raw.strip().lower()
        if s == "cosine":
            return DistanceMetric.Cosine
        if s in ("dot", "dotproduct", "dot_product", "dot-product", "innerproduct", "inner_product", "ip"):
            return DistanceMetric.DotProduct
        raise ValueError(f"unknown distance metric {s!r} (supported: cosine, dot)")


@dataclass
class IndexResult:
    """src/index.rs:129-134."""

    id: str
    score: float


class HipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"cqs_hip error {code}: {msg}")
        self.code = code


class VectorIndex:
    """The `VectorIndex` trait (src/index.rs:139-239), defaults included."""

    def search(self, query: np.ndarray, k: int) -> List[IndexResult]:
        raise NotImplementedError

    def __len__(self) -> int:
        raise NotImplementedError

    def is_empty(self) -> bool:
        return len(self) == 0

    def name(self) -> str:
        raise NotImplementedError

    def dim(self) -> int:
        raise NotImplementedError

    def search_with_filter(self, query: np.ndarray, k: int, flt: Callable[[str], bool]) -> List[IndexResult]:
        # default: over-fetch 3x, post-filter, take k (src/index.rs:167-193)
        kk = min(k * 3, 2**63 - 1)
        out = [r for r in self.search(query, kk) if flt(r.id)][:k]
        if len(out) < k and len(self) >= k:
            log.warning("Filter-aware search under-returned: returned=%d requested=%d index_size=%d",
                        len(out), k, len(self))
        return out

    def is_poisoned(self) -> bool:  # src/index.rs:203-205
        return False

    def max_k(self) -> Optional[int]:  # src/index.rs:219-221
        return None

    def index_scores_are_cosine(self) -> bool:  # src/index.rs:236-238
        return False


def prepare_index_data(embeddings: Sequence[Tuple[str, np.ndarray]], expected_dim: int):
    """src/hnsw/mod.rs:688-746: validate dims, skip all-zero and non-finite rows.

    Returns (id_map, flat [kept, dim] f32, kept).  Raises ValueError like
    `HnswError::Build` on empty input, dimension mismatch, or nothing kept.
    """
    if len(embeddings) == 0:
        raise ValueError("No embeddings to index")
    for cid, emb in embeddings:
        if len(emb) != expected_dim:
            raise ValueError(f"Embedding dimension mismatch for {cid}: got {len(emb)}, expected {expected_dim}")
    id_map: List[str] = []
    rows = []
    for cid, emb in embeddings:
        v = np.asarray(emb, dtype=np.float32)
        if not np.any(v != 0.0):
            log.warning("Skipping zero-vector embedding chunk_id=%s", cid)
            continue
        if not np.all(np.isfinite(v)):
            log.warning("Skipping non-finite embedding chunk_id=%s", cid)
            continue
        id_map.append(cid)
        rows.append(v)
    if not id_map:
        raise ValueError("No valid embeddings to index (all were zero-vector or non-finite)")
    return id_map, np.stack(rows).astype(np.float32, copy=False), len(id_map)


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class HipIndex(VectorIndex):
    """Exact GPU index: `[n, dim]` f32 rows resident in HBM, brute-force scan + top-k.

    Same shape as `CagraIndex` (src/cagra.rs:255-277): owns `id_map` (row -> chunk id,
    rowid order), serialises device access inside the handle, never raises for device
    trouble in `search` (logs + empty list, src/cagra.rs:445-470,543-626).
    """

    def __init__(self, handle: int, id_map: Optional[List[str]], metric: DistanceMetric):
        self._lib = _lib.load()
        self._h = C.c_void_p(handle)
        self.id_map = id_map
        self.metric = metric
        self._keepalive = None

    # ---- construction ---------------------------------------------------------
    @classmethod
    def build_from_flat(cls, id_map: Optional[List[str]], flat: np.ndarray,
                        metric: DistanceMetric = DistanceMetric.Cosine, device: int = 0,
                        row_base: int = 0) -> "HipIndex":
        """`CagraIndex::build_from_flat` (src/cagra.rs:922-960).  `id_map=None` = integer ids."""
        lib = _lib.load()
        flat = np.ascontiguousarray(flat, dtype=np.float32)
        if flat.ndim != 2:
            raise ValueError("flat must be [n, dim]")
        n, dim = flat.shape
        if id_map is not None and len(id_map) != n:
            raise ValueError("id_map length != rows")
        h = C.c_void_p()
        rc = lib.cqs_hip_index_create(_ptr(flat), n, dim, metric.value, device, row_base, C.byref(h))
        if rc != _lib.OK:
            raise HipError(rc, "cqs_hip_index_create failed")
        return cls(h.value, id_map, metric)

    @classmethod
    def build_sharded(cls, id_map: Optional[List[str]], flat: np.ndarray, devices: Sequence[int],
                      metric: DistanceMetric = DistanceMetric.Cosine, row_base: int = 0) -> "HipIndex":
        """`cqs_hip_index_create_sharded`: ONE process, the rows cut over `devices` (rowid order); the handle
        behaves like any other (`search`, `search_with_filter`, `find_neighbors`, `extend`, `save`)."""
        lib = _lib.load()
        flat = np.ascontiguousarray(flat, dtype=np.float32)
        if flat.ndim != 2:
            raise ValueError("flat must be [n, dim]")
        n, dim = flat.shape
        if id_map is not None and len(id_map) != n:
            raise ValueError("id_map length != rows")
        devs = np.ascontiguousarray(list(devices), dtype=np.int32)
        h = C.c_void_p()
        rc = lib.cqs_hip_index_create_sharded(_ptr(flat), n, dim, metric.value, _ptr(devs), len(devs), row_base, C.byref(h))
        if rc != _lib.OK:
            raise HipError(rc, "cqs_hip_index_create_sharded failed")
        return cls(h.value, id_map, metric)

    def shards(self):
        """[(device, first_row, rows, gathers_with_rccl)] per shard (one entry for a single-device index)."""
        out = []
        for s in range(int(self._lib.cqs_hip_index_shards(self._h))):
            dev, first, rows, rc = C.c_int32(), C.c_uint64(), C.c_uint64(), C.c_int32()
            if self._lib.cqs_hip_index_shard_info(self._h, s, C.byref(dev), C.byref(first), C.byref(rows), C.byref(rc)) == _lib.OK:
                out.append((dev.value, first.value, rows.value, bool(rc.value)))
        return out

    @classmethod
    def build_from_device(cls, id_map: Optional[List[str]], d_ptr: int, n: int, dim: int,
                          metric: DistanceMetric = DistanceMetric.Cosine, device: int = 0,
                          row_base: int = 0, borrow: bool = True, keepalive=None) -> "HipIndex":
        lib = _lib.load()
        h = C.c_void_p()
        rc = lib.cqs_hip_index_create_device(C.c_void_p(d_ptr), n, dim, metric.value, device, row_base,
                                             1 if borrow else 0, C.byref(h))
        if rc != _lib.OK:
            raise HipError(rc, "cqs_hip_index_create_device failed")
        idx = cls(h.value, id_map, metric)
        idx._keepalive = keepalive if borrow else None
        return idx

    @classmethod
    def build_from_embeddings(cls, embeddings: Sequence[Tuple[str, np.ndarray]], dim: int,
                              metric: DistanceMetric = DistanceMetric.Cosine, device: int = 0) -> "HipIndex":
        id_map, flat, _ = prepare_index_data(embeddings, dim)
        return cls.build_from_flat(id_map, flat, metric, device)

    def extend(self, ids: Optional[List[str]], rows: np.ndarray) -> None:
        """Incremental add (tiered.rs extend contract, src/tiered.rs:1-43)."""
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        if rows.ndim != 2 or rows.shape[1] != self.dim():
            raise ValueError("rows must be [m, dim]")
        if (self.id_map is None) != (ids is None):
            raise ValueError("ids must match the index's id flavour")
        rc = self._lib.cqs_hip_index_extend(self._h, _ptr(rows), rows.shape[0])
        if rc != _lib.OK:
            raise HipError(rc, self.last_error())
        if ids is not None:
            self.id_map.extend(ids)

    # ---- persistence: blob + CagraMeta-style sidecar (src/cagra.rs:973-1157, 1174-1330) -----------
    META_MAGIC = "cqs-hip-flat-meta"
    META_VERSION = 1

    def save(self, path: str) -> None:
        """`CagraIndex::save`: rows blob through the C ABI (atomic tmp -> rename) + `{path}.meta` JSON
        {magic, version, dim, chunk_count, id_map, checksum, metric} (also tmp -> rename)."""
        import json
        import os
        ck = C.c_uint64()
        rc = self._lib.cqs_hip_index_save(self._h, path.encode(), C.byref(ck))
        if rc != _lib.OK:
            raise HipError(rc, self.last_error())
        meta = {"magic": self.META_MAGIC, "version": self.META_VERSION, "dim": self.dim(), "chunk_count": len(self),
                "id_map": self.id_map, "checksum": f"{ck.value:016x}", "metric": self.metric.as_str()}
        tmp = path + ".meta.tmp"
        try:
            with open(tmp, "w") as f:
                json.dump(meta, f)
            os.replace(tmp, path + ".meta")
        except OSError:
            for p in (path, path + ".meta", tmp):   # a blob without its sidecar is useless (cagra.rs:1143-1147)
                try:
                    os.remove(p)
                except OSError:
                    pass
            raise

    @classmethod
    def load(cls, path: str, dim: int, chunk_count: int, device: int = 0,
             devices: Optional[Sequence[int]] = None) -> "HipIndex":
        """`CagraIndex::load`: sidecar magic / version / dim / chunk_count must match the store, the blob's
        checksum must match the sidecar; anything else raises ValueError (caller deletes + rebuilds)."""
        import json
        lib = _lib.load()
        try:
            with open(path + ".meta") as f:
                meta = json.load(f)
        except (OSError, ValueError) as e:
            raise ValueError(f"HIP index sidecar unreadable: {e}")
        if meta.get("magic") != cls.META_MAGIC or meta.get("version") != cls.META_VERSION:
            raise ValueError("HIP index sidecar: bad magic / version")
        if meta.get("dim") != dim or meta.get("chunk_count") != chunk_count:
            raise ValueError("HIP index sidecar: stale (dim / chunk_count mismatch)")
        ids = meta.get("id_map")
        if ids is not None and len(ids) != chunk_count:
            raise ValueError("HIP index sidecar: id_map length mismatch")
        # the sidecar must describe THIS blob: compare its checksum with the blob header's
        import struct
        try:
            with open(path, "rb") as f:
                magic, _ver, _dim, _metric, _pad, _rows, blob_ck = struct.unpack("<8sIIIIQQ", f.read(40))
        except (OSError, struct.error) as e:
            raise ValueError(f"HIP index blob unreadable: {e}")
        if magic != b"CQSHIPF1" or meta.get("checksum") != f"{blob_ck:016x}":
            raise ValueError("HIP index sidecar does not match the blob (checksum)")
        h = C.c_void_p()
        if devices is not None:
            devs = np.ascontiguousarray(list(devices), dtype=np.int32)
            rc = lib.cqs_hip_index_load_sharded(path.encode(), dim, chunk_count, _ptr(devs), len(devs), 0, C.byref(h))
        else:
            rc = lib.cqs_hip_index_load(path.encode(), dim, chunk_count, device, 0, C.byref(h))   # verifies the content
        if rc != _lib.OK:
            raise ValueError(f"HIP index blob rejected (rc={rc})")
        return cls(h.value, ids, DistanceMetric.parse(meta.get("metric", "cosine")))

    @staticmethod
    def delete_persisted(path: str) -> None:
        """`CagraIndex::delete_persisted` (src/cagra.rs:1739-1750)."""
        import os
        for p in (path, path + ".meta"):
            try:
                os.remove(p)
            except OSError:
                pass

    def close(self) -> None:
        if self._h:
            self._lib.cqs_hip_index_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- trait ------------------------------------------------------------------
    def __len__(self) -> int:
        return int(self._lib.cqs_hip_index_len(self._h))

    def name(self) -> str:
        return "HIP"

    def dim(self) -> int:
        return int(self._lib.cqs_hip_index_dim(self._h))

    def is_poisoned(self) -> bool:
        return bool(self._lib.cqs_hip_index_poisoned(self._h))

    def max_k(self) -> Optional[int]:
        return int(self._lib.cqs_hip_index_max_k(self._h))

    def index_scores_are_cosine(self) -> bool:
        # exact dot of unit vectors == the brute-force cosine (src/search/query.rs:1152-1172)
        return self.metric is DistanceMetric.Cosine

    def last_error(self) -> str:
        buf = C.create_string_buffer(512)
        self._lib.cqs_hip_index_last_error(self._h, buf, 512)
        return buf.value.decode("utf-8", "replace")

    def _id(self, row: int) -> str:
        return str(row) if self.id_map is None else self.id_map[row - int(self._lib.cqs_hip_index_row_base(self._h))]

    def search_batch(self, queries: np.ndarray, k: int, keep_bitset: Optional[np.ndarray] = None,
                     mode: int = _lib.MODE_RAW, threshold: float = 0.0):
        """Block of queries through `cqs_hip_index_search`.  Returns (rows u64 [b,k], scores f32 [b,k], counts u32 [b])."""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim == 1:
            q = q[None, :]
        b, qd = q.shape
        rows = np.zeros((b, max(k, 1)), dtype=np.uint64)
        scores = np.zeros((b, max(k, 1)), dtype=np.float32)
        counts = np.zeros((b,), dtype=np.uint32)
        kb = None
        if keep_bitset is not None:
            kb = np.ascontiguousarray(keep_bitset, dtype=np.uint32)
            if kb.shape[0] < (len(self) + 31) // 32:
                raise ValueError("keep_bitset too short")
        rc = self._lib.cqs_hip_index_search(self._h, _ptr(q), b, qd, k, _ptr(kb), mode, threshold,
                                            _ptr(rows), _ptr(scores), _ptr(counts))
        if rc != _lib.OK:
            raise HipError(rc, self.last_error())
        return rows[:, :k], scores[:, :k], counts

    def search(self, query: np.ndarray, k: int) -> List[IndexResult]:
        """`VectorIndex::search` (src/index.rs:146): sorted by score desc; never raises for device trouble."""
        if self.is_empty() or k == 0:
            return []
        query = np.asarray(query, dtype=np.float32).reshape(-1)
        if query.shape[0] != self.dim():
            log.warning("Query dimension mismatch expected_dim=%d actual_dim=%d", self.dim(), query.shape[0])
            return []
        if not np.all(np.isfinite(query)):
            log.warning("HIP query embedding contains non-finite values (NaN/Inf), returning empty results")
            return []
        k = min(k, self.max_k())
        try:
            rows, scores, counts = self.search_batch(query, k)
        except HipError as e:
            log.error("HIP search failed: %s", e)
            return []
        c = int(counts[0])
        return [IndexResult(self._id(int(rows[0, i])), float(scores[0, i])) for i in range(c)]

    def search_with_filter(self, query: np.ndarray, k: int, flt: Callable[[str], bool]) -> List[IndexResult]:
        """GPU-native filtered search: host builds the keep-bitset by evaluating the predicate per
        id (src/cagra.rs:747-757); all-pass -> unfiltered, none -> [], k capped at `included`
        (src/cagra.rs:760-775, done inside the C ABI)."""
        if self.is_empty() or k == 0:
            return []
        query = np.asarray(query, dtype=np.float32).reshape(-1)
        if query.shape[0] != self.dim():
            log.warning("Query dimension mismatch expected_dim=%d actual_dim=%d", self.dim(), query.shape[0])
            return []
        if not np.all(np.isfinite(query)):
            return []
        n = len(self)
        base = int(self._lib.cqs_hip_index_row_base(self._h))
        keep = np.fromiter((bool(flt(self._id(base + i))) for i in range(n)), dtype=bool, count=n)
        bits = np.packbits(keep, bitorder="little")
        bits = np.concatenate([bits, np.zeros((-len(bits)) % 4, dtype=np.uint8)]).view(np.uint32)
        k = min(k, self.max_k())
        try:
            rows, scores, counts = self.search_batch(query, k, keep_bitset=bits)
        except HipError as e:
            log.error("HIP filtered search failed: %s", e)
            return []
        c = int(counts[0])
        return [IndexResult(self._id(int(rows[0, i])), float(scores[0, i])) for i in range(c)]

    def neighbors_rows(self, target_row: int, limit: int):
        """`cqs_hip_index_neighbors`: (rows u64, scores f32) of the stored row's nearest neighbours, itself excluded."""
        rows = np.zeros((_lib.NEIGHBORS_MAX,), dtype=np.uint64)
        scores = np.zeros((_lib.NEIGHBORS_MAX,), dtype=np.float32)
        c = C.c_uint32()
        rc = self._lib.cqs_hip_index_neighbors(self._h, target_row, max(0, min(int(limit), 2**32 - 1)), _ptr(rows), _ptr(scores), C.byref(c))
        if rc != _lib.OK:
            raise HipError(rc, self.last_error())
        return rows[:c.value], scores[:c.value]

    def find_neighbors(self, target_id: str, limit: int) -> List[IndexResult]:
        """`find_neighbors(store, target, limit)` (src/cli/commands/search/neighbors.rs:86-132) on the resident
        corpus: ids instead of `ChunkSummary`s (hydration stays with the store, :139-146).  Raises `KeyError`
        when the target is not indexed (the reference: "Could not load embedding for ..."), :98-106."""
        base = int(self._lib.cqs_hip_index_row_base(self._h))
        if self.id_map is None:
            row = int(target_id)
            if not (base <= row < base + len(self)):
                raise KeyError(target_id)
        else:
            if not hasattr(self, "_row_of") or len(self._row_of) != len(self.id_map):
                self._row_of = {cid: i for i, cid in enumerate(self.id_map)}
            if target_id not in self._row_of:
                raise KeyError(target_id)
            row = base + self._row_of[target_id]
        rows, scores = self.neighbors_rows(row, limit)
        return [IndexResult(self._id(int(r)), float(s)) for r, s in zip(rows, scores)]

    # ---- device-resident path (bench, sharded search) ---------------------------
    def search_device(self, d_queries: int, b: int, k: int, d_out_keys: int, d_out_counts: int,
                      d_keep: int = 0, mode: int = _lib.MODE_RAW, threshold: float = 0.0, stream: int = 0) -> None:
        """Enqueue `cqs_hip_index_search_device` on `stream` (raw device pointers / hipStream_t as ints)."""
        rc = self._lib.cqs_hip_index_search_device(
            self._h, C.c_void_p(d_queries), b, k, C.c_void_p(d_keep) if d_keep else None, mode, threshold,
            C.c_void_p(d_out_keys), C.c_void_p(d_out_counts), C.c_void_p(stream) if stream else None)  # 0 -> NULL = null stream
        if rc != _lib.OK:
            raise HipError(rc, self.last_error())

    def combine_stats(self) -> Tuple[int, int]:
        """(passes, queries) the handle's combining queue has run since it was made (`cqs_hip_index_combine_stats`)."""
        p, q = C.c_uint64(), C.c_uint64()
        self._lib.cqs_hip_index_combine_stats(self._h, C.byref(p), C.byref(q))
        return int(p.value), int(q.value)

    def set_timing(self, on: bool) -> None:
        self._lib.cqs_hip_index_set_timing(self._h, 1 if on else 0)

    def scan_time(self) -> Tuple[int, float]:
        """(searches bracketed, summed scan-kernel milliseconds) since timing was enabled / last read."""
        n, ms = C.c_uint32(), C.c_double()
        rc = self._lib.cqs_hip_index_scan_time(self._h, C.byref(n), C.byref(ms))
        if rc != _lib.OK:
            raise HipError(rc, self.last_error())
        return int(n.value), float(ms.value)


def unpack_keys(keys: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Decode packed candidate keys -> (rows u64, scores f32) via the C ABI helper."""
    lib = _lib.load()
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    rows = np.zeros(keys.shape, dtype=np.uint64)
    scores = np.zeros(keys.shape, dtype=np.float32)
    lib.cqs_hip_unpack_keys(_ptr(keys), keys.size, _ptr(rows), _ptr(scores))
    return rows, scores


def merge_keys(lists: np.ndarray, counts: np.ndarray, k: int) -> np.ndarray:
    """Host k-way merge of per-shard descending key lists ([n_lists, stride] u64) -> top-k keys."""
    lib = _lib.load()
    lists = np.ascontiguousarray(lists, dtype=np.uint64)
    counts = np.ascontiguousarray(counts, dtype=np.uint32)
    out = np.zeros((k,), dtype=np.uint64)
    c = lib.cqs_hip_merge_keys(_ptr(lists), _ptr(counts), lists.shape[0], lists.shape[1], k, _ptr(out))
    return out[:c]


# ---- backend registration (src/index.rs:245-345) ------------------------------
@dataclass
class BackendContext:
    """src/index.rs:245-260: what a backend may look at when deciding to open."""

    cqs_dir: str
    store: object            # needs .dim, .chunk_count(), .embedding_batches(batch) -> iter[list[(id, vec)]]
    ef_search: Optional[int] = None
    hip_threshold: int = 5000  # same gate as CQS_CAGRA_THRESHOLD (src/cagra.rs:1683-1690)
    device: int = 0
    persist: bool = True       # CQS_CAGRA_PERSIST analogue (src/cagra.rs:1013)
    devices: Optional[Sequence[int]] = None   # CQS_HIP_DEVICES: shard the corpus over these GPUs (one process)


def dim_scaled_batch(baseline: int, dim: int, lo: int, hi: int) -> int:
    """src/limits.rs:292-300."""
    if dim == 0:
        return max(lo, min(baseline, hi))
    return max(lo, min(baseline * 1024 // dim, hi))


class HipBackend:
    """`IndexBackend` (src/index.rs:271-291) for the exact GPU index; modelled on
    `CagraBackend::try_open` (src/cagra.rs:1676-1802): `None` = not applicable
    (falls through to the next backend, then brute force, src/cli/store.rs:503-508)."""

    def name(self) -> str:
        return "hip"

    def priority(self) -> int:
        return 200  # above cagra (100) and tiered (150), src/index.rs:338-345

    def try_open(self, ctx: BackendContext) -> Optional[VectorIndex]:
        lib = _lib.load()
        n = ctx.store.chunk_count()
        if n < ctx.hip_threshold:
            return None
        if lib.cqs_hip_device_count() <= 0:
            return None
        dim = ctx.store.dim
        free, total = C.c_uint64(), C.c_uint64()
        devs = list(ctx.devices) if ctx.devices else [ctx.device]
        # gpu_available_for (src/cagra.rs:336-376) for EVERY device that will hold a shard: a device counted twice
        # (two shards on one GPU) must have room for both
        for d in sorted(set(devs)):
            if lib.cqs_hip_device_mem(d, C.byref(free), C.byref(total)) != _lib.OK:
                return None
            if n * dim * 4 * 1.25 * devs.count(d) / len(devs) > free.value:
                log.warning("HIP backend: corpus does not fit the memory of device %d, falling through", d)
                return None
        import os
        path = os.path.join(ctx.cqs_dir, "index.hipflat")
        if ctx.persist and os.path.exists(path):           # persisted first (src/cagra.rs:1726-1752)
            try:
                idx = HipIndex.load(path, dim, n, ctx.device, devices=ctx.devices)
                log.info("Vector index backend selected backend=hip source=persisted vectors=%d", len(idx))
                return idx
            except ValueError as e:
                log.warning("HIP persisted load failed, rebuilding from store: %s", e)
                HipIndex.delete_persisted(path)
        embeddings = []
        for batch in ctx.store.embedding_batches(dim_scaled_batch(10_000, dim, 500, 50_000)):
            embeddings.extend(batch)
        try:
            if ctx.devices:
                id_map, flat, _ = prepare_index_data(embeddings, dim)
                idx = HipIndex.build_sharded(id_map, flat, ctx.devices, DistanceMetric.Cosine)
            else:
                idx = HipIndex.build_from_embeddings(embeddings, dim, DistanceMetric.Cosine, ctx.device)
        except (ValueError, HipError) as e:
            log.warning("HIP backend build failed, falling through: %s", e)
            return None
        log.info("Vector index backend selected backend=hip source=rebuilt vectors=%d", len(idx))
        if ctx.persist and len(idx) == n:                  # best-effort save (src/cagra.rs:1786-1795)
            try:
                idx.save(path)
            except (HipError, OSError) as e:
                log.warning("Failed to persist HIP index (will rebuild next restart): %s", e)
        return idx
