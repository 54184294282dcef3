"""Host mirror of cqs's two BERT-family auxiliary models over the C ABI (include/cqs_hip.h, BERT section):

  * `SpladeEncoder` — `SpladeEncoder::{encode, encode_batch}` (src/splade/mod.rs:595-760, :774-1075) below the
    tokenizer: token-id sequences in, `SparseVector`s (`Vec<(u32, f32)>`, ascending id, weight > threshold) out;
  * `Reranker` — `Reranker::compute_scores_opt` (src/reranker.rs:343-533) below the tokenizer: encoded (query, passage)
    pairs in, `sigmoid(logit)` scores out.

The tokenizer stays on the host in the reference too (tokenizers crate); tests and the bench feed token ids.
There is no CPU fallback here: without libcqs_hip.so / a GPU the constructors raise."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib

SparseVector = List[Tuple[int, float]]
DEFAULT_SPLADE_THRESHOLD = 0.01        # src/splade/mod.rs:400-410 (CQS_SPLADE_THRESHOLD overrides it in the reference)


class BertError(RuntimeError):
    pass


def bert_config(head: int, **overrides) -> _lib.BertConfig:
    cfg = _lib.BertConfig()
    rc = _lib.load().cqs_hip_bert_config_default(head, C.byref(cfg))
    if rc != _lib.OK:
        raise BertError(f"cqs_hip_bert_config_default failed ({rc})")
    for k, v in overrides.items():
        setattr(cfg, k, v)
    return cfg


class HipBertEngine:
    """One `cqs_hip_bert` handle (an encoder + one head)."""

    def __init__(self, cfg: _lib.BertConfig, device: int = 0):
        self._lib = _lib.load()
        self.cfg = cfg
        h = C.c_void_p()
        rc = self._lib.cqs_hip_bert_create(C.byref(cfg), device, C.byref(h))
        if rc != _lib.OK:
            raise BertError(f"cqs_hip_bert_create failed ({rc})")
        self._h = h

    @classmethod
    def load_dir(cls, model_dir: str, cfg: _lib.BertConfig, device: int = 0) -> "HipBertEngine":
        """`cqs_hip_bert_load_dir`: onnx/model.onnx, model.onnx or model.safetensors under `model_dir`."""
        self = cls.__new__(cls)
        self._lib = _lib.load()
        self.cfg = cfg
        h = C.c_void_p()
        rc = self._lib.cqs_hip_bert_load_dir(str(model_dir).encode(), C.byref(cfg), device, C.byref(h))
        self._h = h if rc == _lib.OK else None
        if rc != _lib.OK:
            raise BertError(f"cqs_hip_bert_load_dir failed ({rc})")
        return self

    def close(self):
        if self._h:
            self._lib.cqs_hip_bert_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def last_error(self) -> str:
        buf = C.create_string_buffer(512)
        self._lib.cqs_hip_bert_last_error(self._h, buf, 512)
        return buf.value.decode("utf-8", "replace")

    def _check(self, rc: int, what: str):
        if rc != _lib.OK:
            raise BertError(f"{what}: {self.last_error()} ({rc})")

    def set_weights(self, weights: Dict[str, np.ndarray]):
        for name, t in weights.items():
            a = np.ascontiguousarray(t, dtype=np.float32)
            self._check(self._lib.cqs_hip_bert_set_tensor(self._h, name.encode(), a.ctypes.data_as(C.c_void_p), a.size),
                        "set_tensor " + name)
        self._check(self._lib.cqs_hip_bert_finalize(self._h), "finalize")

    @staticmethod
    def _pack(seqs: Sequence[np.ndarray]):
        lens = np.array([len(s) for s in seqs], np.uint32)
        toks = np.concatenate([np.asarray(s, np.int32) for s in seqs]) if len(seqs) and lens.sum() else np.zeros(0, np.int32)
        return np.ascontiguousarray(toks, np.int32), lens

    def splade_dense(self, seqs: Sequence[np.ndarray]) -> np.ndarray:
        """[B, vocab] f32 activations ln(1 + max(0, max_s logits))."""
        toks, lens = self._pack(seqs)
        out = np.empty((len(seqs), int(self.cfg.vocab_size)), np.float32)
        self._check(self._lib.cqs_hip_splade_encode(self._h, toks.ctypes.data_as(C.c_void_p), lens.ctypes.data_as(C.c_void_p),
                                                    len(seqs), out.ctypes.data_as(C.c_void_p)), "splade_encode")
        return out

    def splade_sparse(self, seqs: Sequence[np.ndarray], threshold: float, cap: int = 2048):
        """`cqs_hip_splade_encode_sparse`: (ids u32 [B, cap], weights f32 [B, cap], counts u32 [B]); counts may exceed cap."""
        toks, lens = self._pack(seqs)
        B = len(seqs)
        ids = np.empty((B, cap), np.uint32); wts = np.empty((B, cap), np.float32); cnt = np.empty(B, np.uint32)
        self._check(self._lib.cqs_hip_splade_encode_sparse(self._h, toks.ctypes.data_as(C.c_void_p), lens.ctypes.data_as(C.c_void_p),
                                                           B, C.c_float(threshold), cap, ids.ctypes.data_as(C.c_void_p),
                                                           wts.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p)),
                    "splade_encode_sparse")
        return ids, wts, cnt

    def _pack_types(self, type_ids, lens):
        """token_type_ids packed like the ids; the C side reads one per token, so the lengths must line up."""
        if type_ids is None:
            return None
        tt, tl = self._pack(type_ids)
        if not np.array_equal(tl, lens):
            raise BertError("token_type_ids do not line up with input_ids")
        return tt

    # ---- tickets: submit returns at once, collect waits (up to 3 in flight; consecutive tickets alternate between the
    # engine's two execution contexts) - the overlap the index pipeline wants from its SPLADE / BERT embed stages
    def submit_sparse(self, seqs: Sequence[np.ndarray], threshold: float, cap: int = 2048) -> Tuple[int, int, int]:
        """`cqs_hip_splade_submit_sparse` -> ticket handle (ticket, batch, cap) for `collect_sparse`."""
        toks, lens = self._pack(seqs)
        t = C.c_uint64()
        self._check(self._lib.cqs_hip_splade_submit_sparse(self._h, toks.ctypes.data_as(C.c_void_p), lens.ctypes.data_as(C.c_void_p),
                                                           len(seqs), C.c_float(threshold), cap, C.byref(t)), "splade_submit_sparse")
        return int(t.value), len(seqs), int(cap)

    def collect_sparse(self, handle: Tuple[int, int, int]):
        ticket, B, cap = handle
        ids = np.empty((B, cap), np.uint32); wts = np.empty((B, cap), np.float32); cnt = np.empty(B, np.uint32)
        self._check(self._lib.cqs_hip_splade_collect_sparse(self._h, ticket, ids.ctypes.data_as(C.c_void_p),
                                                            wts.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p)),
                    "splade_collect_sparse")
        return ids, wts, cnt

    def abandon_sparse(self, handle: Tuple[int, int, int]) -> None:
        self._lib.cqs_hip_splade_collect_sparse(self._h, handle[0], None, None, None)

    def embed_submit(self, seqs: Sequence[np.ndarray], type_ids: Optional[Sequence[np.ndarray]] = None, pooling: str = "mean") -> Tuple[int, int]:
        toks, lens = self._pack(seqs)
        tt = self._pack_types(type_ids, lens)
        t = C.c_uint64()
        self._check(self._lib.cqs_hip_bert_embed_submit(self._h, toks.ctypes.data_as(C.c_void_p),
                                                        tt.ctypes.data_as(C.c_void_p) if tt is not None else None,
                                                        lens.ctypes.data_as(C.c_void_p), len(seqs), {"mean": 0, "cls": 1}[pooling],
                                                        C.byref(t)), "bert_embed_submit")
        return int(t.value), len(seqs)

    def embed_collect(self, handle: Tuple[int, int]) -> np.ndarray:
        out = np.empty((handle[1], int(self.cfg.hidden)), np.float32)
        self._check(self._lib.cqs_hip_bert_embed_collect(self._h, handle[0], out.ctypes.data_as(C.c_void_p)), "bert_embed_collect")
        return out

    def embed_abandon(self, handle: Tuple[int, int]) -> None:
        self._lib.cqs_hip_bert_embed_collect(self._h, handle[0], None)

    def rerank_logits(self, seqs: Sequence[np.ndarray], type_ids: Optional[Sequence[np.ndarray]]) -> np.ndarray:
        toks, lens = self._pack(seqs)
        tt = self._pack_types(type_ids, lens)
        out = np.empty((len(seqs), int(self.cfg.num_labels)), np.float32)
        self._check(self._lib.cqs_hip_rerank_logits(self._h, toks.ctypes.data_as(C.c_void_p),
                                                    tt.ctypes.data_as(C.c_void_p) if tt is not None else None,
                                                    lens.ctypes.data_as(C.c_void_p), len(seqs), out.ctypes.data_as(C.c_void_p)),
                    "rerank_logits")
        return out

    def embed(self, seqs: Sequence[np.ndarray], type_ids: Optional[Sequence[np.ndarray]] = None, pooling: str = "mean") -> np.ndarray:
        """`cqs_hip_bert_embed`: pooled sentence vectors f32 [B, hidden], not normalised (head = NONE engines)."""
        toks, lens = self._pack(seqs)
        tt = self._pack_types(type_ids, lens)
        out = np.empty((len(seqs), int(self.cfg.hidden)), np.float32)
        self._check(self._lib.cqs_hip_bert_embed(self._h, toks.ctypes.data_as(C.c_void_p),
                                                 tt.ctypes.data_as(C.c_void_p) if tt is not None else None,
                                                 lens.ctypes.data_as(C.c_void_p), len(seqs), {"mean": 0, "cls": 1}[pooling],
                                                 out.ctypes.data_as(C.c_void_p)), "bert_embed")
        return out

    def hidden(self, seqs: Sequence[np.ndarray], type_ids: Optional[Sequence[np.ndarray]] = None) -> np.ndarray:
        toks, lens = self._pack(seqs)
        tt = self._pack_types(type_ids, lens)
        out = np.empty((int(lens.sum()), int(self.cfg.hidden)), np.float32)
        self._check(self._lib.cqs_hip_bert_hidden(self._h, toks.ctypes.data_as(C.c_void_p),
                                                  tt.ctypes.data_as(C.c_void_p) if tt is not None else None,
                                                  lens.ctypes.data_as(C.c_void_p), len(seqs), out.ctypes.data_as(C.c_void_p)),
                    "bert_hidden")
        return out


class SpladeEncoder:
    """`SpladeEncoder` (src/splade/mod.rs:95-118) over a HIP engine: `encode_batch` on token-id sequences."""

    def __init__(self, engine: HipBertEngine, threshold: float = DEFAULT_SPLADE_THRESHOLD, max_seq_len: int = 512,
                 sparse_cap: int = 2048):
        self.engine, self.threshold, self.max_seq_len = engine, float(threshold), int(max_seq_len)
        self.sparse_cap = int(sparse_cap)

    def encode_batch_arrays(self, seqs: Sequence[Sequence[int]]) -> List[Tuple[np.ndarray, np.ndarray]]:
        """`encode_batch` with each sparse vector as (ids u32 ascending, weights f32) arrays - what a caller that
        feeds an inverted index wants; one vectorised threshold pass over the [batch, vocab] activations."""
        if not len(seqs):
            return []
        cut = [np.asarray(s, np.int32)[: self.max_seq_len] for s in seqs]     # truncation (src/splade/mod.rs:860-880)
        if hasattr(self.engine, "splade_sparse"):
            # threshold filter on the device; a row with more survivors than the cap goes through the dense form
            ids, wts, cnt = self.engine.splade_sparse(cut, self.threshold, self.sparse_cap)
            out = [(ids[b, :cnt[b]].copy(), wts[b, :cnt[b]].copy()) if cnt[b] <= self.sparse_cap else None for b in range(len(cut))]
            over = [b for b, o in enumerate(out) if o is None]
            if over:
                for b, o in zip(over, self._from_dense([cut[b] for b in over])):
                    out[b] = o
            return out
        return self._from_dense(cut)

    def _from_dense(self, cut) -> List[Tuple[np.ndarray, np.ndarray]]:
        dense = self.engine.splade_dense(cut)
        rows, cols = np.nonzero(dense > np.float32(self.threshold))           # row-major: ascending id per row; NaN > t is False
        bounds = np.searchsorted(rows, np.arange(len(cut) + 1))
        vals = dense[rows, cols]
        return [(cols[bounds[b]:bounds[b + 1]].astype(np.uint32), vals[bounds[b]:bounds[b + 1]]) for b in range(len(cut))]

    DEPTH = 3   # tickets in flight (= the library's submission slots)

    def encode_batches_arrays(self, batches: Sequence[Sequence[Sequence[int]]]) -> List[List[Tuple[np.ndarray, np.ndarray]]]:
        """The index pipeline's form: a stream of batches with up to three in flight (host packing and PCIe of batch
        i + 1, i + 2 under batch i's kernels; two kernel chains interleaved on the device).  Same results as
        `encode_batch_arrays` per batch, in order.  A failing batch abandons the tickets in flight (their slots come
        back) before the error propagates."""
        out: List = [None] * len(batches)
        inflight = []                                         # (batch index, handle, cut)
        nxt = 0
        try:
            while nxt < len(batches) or inflight:
                while nxt < len(batches) and len(inflight) < self.DEPTH:
                    cut = [np.asarray(s, np.int32)[: self.max_seq_len] for s in batches[nxt]]
                    if len(cut):
                        inflight.append((nxt, self.engine.submit_sparse(cut, self.threshold, self.sparse_cap), cut))
                    else:
                        out[nxt] = []
                    nxt += 1
                if not inflight:
                    continue
                bi, handle, cut = inflight.pop(0)
                ids, wts, cnt = self.engine.collect_sparse(handle)
                res = [(ids[b, :cnt[b]].copy(), wts[b, :cnt[b]].copy()) if cnt[b] <= self.sparse_cap else None for b in range(len(cut))]
                over = [b for b, o in enumerate(res) if o is None]
                if over:                                      # more survivors than the cap: that row through the dense form
                    for b, o in zip(over, self._from_dense([cut[b] for b in over])):
                        res[b] = o
                out[bi] = res
        except BaseException:
            for _, handle, _ in inflight:
                self.engine.abandon_sparse(handle)
            raise
        return out

    def encode_batch(self, seqs: Sequence[Sequence[int]]) -> List[SparseVector]:
        """`SpladeEncoder::encode_batch`: `Vec<SparseVector>`, a sparse vector = [(token id, weight)] ascending id."""
        return [list(zip(ids.tolist(), wts.tolist())) for ids, wts in self.encode_batch_arrays(seqs)]

    def encode(self, seq: Sequence[int]) -> SparseVector:
        return self.encode_batch([seq])[0]


class Reranker:
    """`Reranker::compute_scores_opt` (src/reranker.rs:343-533) on encoded pairs: sigmoid of the first logit."""

    def __init__(self, engine: HipBertEngine, max_length: int = 512):
        self.engine, self.max_length = engine, int(max_length)

    def scores(self, ids: Sequence[Sequence[int]], type_ids: Optional[Sequence[Sequence[int]]] = None) -> np.ndarray:
        if not len(ids):
            return np.zeros(0, np.float32)
        a = [np.asarray(s, np.int32)[: self.max_length] for s in ids]
        t = [np.asarray(s, np.int32)[: self.max_length] for s in type_ids] if type_ids is not None else None
        logits = self.engine.rerank_logits(a, t)
        return (1.0 / (1.0 + np.exp(-logits[:, 0].astype(np.float64)))).astype(np.float32)
