"""Index-pipeline glue around the HIP embedding engine.

Reference surface mirrored (src/cli/pipeline/embedding.rs unless noted):
  gpu_embed_stage :226-421   one ParsedBatch at a time: prepare -> `embed_documents` -> EmbeddedBatch to the writer;
                             a GPU failure counts the chunks into `gpu_failures` and requeues them to the CPU stage
  flush_to_cpu    :160-223   on failure: cached embeddings go on to the writer, `to_embed` goes to `fail_tx`
  embed_documents core.rs:718-751, embed_batch_size() models.rs:789-817 (= 32 for EmbeddingGemma)

What is MI355X-first here: the reference hands ORT batches of 32 texts in input order, padded to the longest text
of each batch (core.rs:1020-1035).  The HIP forward runs PACKED tokens, so the unit that matters is tokens per
launch, not sequences: `EmbedPipeline` sorts the chunks by length, cuts batches by a token budget
(`embed_batch_size() x 512` tokens by default, at most 8 x embed_batch_size() sequences), keeps up to three batches
in flight through `cqs_hip_embed_submit_ragged` / `_collect` (pinned staging inside the library: the host packs
batch i+1 while the device runs batch i) and returns the rows in INPUT order, L2-normalised exactly like
`normalize_l2` (pooling.rs:60-67).  Tokenisation stays with the caller (host, as in the reference).
"""
from __future__ import annotations

import ctypes as C
import time
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence

import numpy as np

from . import _lib
from .embedder import EmbedderError, HipEmbedEngine, embed_batch_size


def plan_batches(lens: Sequence[int], token_budget: int, max_seqs: int) -> List[np.ndarray]:
    """Length-sorted batches (longest first, so the device scratch is sized once): each batch holds at most
    `max_seqs` sequences and at most `token_budget` tokens (one sequence alone may exceed the budget).
    Returns index arrays into `lens`; every index appears exactly once."""
    lens = np.asarray(lens, dtype=np.int64)
    order = np.argsort(-lens, kind="stable")
    out, cur, tok = [], [], 0
    for i in order:
        L = int(lens[i])
        if cur and (len(cur) >= max_seqs or tok + L > token_budget):
            out.append(np.asarray(cur, dtype=np.int64))
            cur, tok = [], 0
        cur.append(int(i))
        tok += L
    if cur:
        out.append(np.asarray(cur, dtype=np.int64))
    return out


def normalize_l2_rows(rows: np.ndarray) -> np.ndarray:
    """`normalize_l2` (pooling.rs:60-67) on every row, in place (C helper: exact f32 left-to-right semantics)."""
    assert rows.dtype == np.float32 and rows.flags["C_CONTIGUOUS"] and rows.ndim == 2
    _lib.load().cqs_hip_normalize_l2_rows(rows.ctypes.data_as(C.c_void_p), rows.shape[0], rows.shape[1])
    return rows


class EmbedPipeline:
    """Length-sorted, token-budgeted, 3-deep pipelined embedding of tokenised chunks."""

    DEPTH = 3   # tickets in flight (= the library's submission slots)

    def __init__(self, engine: HipEmbedEngine, token_budget: Optional[int] = None, max_seqs: Optional[int] = None):
        self.engine = engine
        base = embed_batch_size(engine.dim(), engine.max_seq())           # 32 for EmbeddingGemma
        self.token_budget = int(token_budget or base * 512)               # 16 384 tokens = 256 attention workgroups
        self.max_seqs = int(max_seqs or base * 8)
        self.max_seq = engine.max_seq()
        self._stats = {"batches": 0, "chunks": 0, "tokens": 0, "submit_s": 0.0, "collect_wait_s": 0.0}

    def abandon(self, tickets) -> None:
        """Give the library's submission slots of `tickets` back (results discarded, errors ignored)."""
        for t in tickets:
            try:
                self.engine.abandon(t)
            except EmbedderError:
                pass                                       # a failed collect releases its slot as well

    def stats(self) -> dict:
        s = dict(self._stats)
        s.update(token_budget=self.token_budget, max_seqs=self.max_seqs, depth=self.DEPTH)
        for k in ("submit_s", "collect_wait_s"):
            s[k] = round(s[k], 4)
        return s

    def embed_token_lists(self, chunks: Sequence[np.ndarray], normalize: bool = True) -> np.ndarray:
        """chunks: token-id arrays (special tokens already added, truncated here to max_seq like
        `max_len = min(longest, max_seq_length)`, core.rs:1020-1025).  -> f32 [len(chunks), dim], input order."""
        n = len(chunks)
        dim = self.engine.dim()
        out = np.zeros((n, dim), dtype=np.float32)
        if n == 0:
            return out
        toks = [np.asarray(c, dtype=np.int32)[: self.max_seq] for c in chunks]
        lens = np.fromiter((t.size for t in toks), dtype=np.int64, count=n)
        batches = plan_batches(lens, self.token_budget, self.max_seqs)
        inflight = []                                     # (ticket, index array)
        nxt = 0
        try:
            while nxt < len(batches) or inflight:
                while nxt < len(batches) and len(inflight) < self.DEPTH:
                    sel = batches[nxt]
                    t0 = time.perf_counter()
                    cat = np.concatenate([toks[i] for i in sel]) if len(sel) else np.zeros(0, np.int32)
                    ticket = self.engine.submit_ragged(cat, lens[sel].astype(np.uint32))
                    self._stats["submit_s"] += time.perf_counter() - t0
                    inflight.append((ticket, sel))
                    nxt += 1
                ticket, sel = inflight.pop(0)
                t0 = time.perf_counter()
                rows = self.engine.collect(ticket, len(sel))
                self._stats["collect_wait_s"] += time.perf_counter() - t0
                out[sel] = rows
                self._stats["batches"] += 1
                self._stats["chunks"] += len(sel)
                self._stats["tokens"] += int(lens[sel].sum())
        except BaseException:
            # A submission slot is freed only when its ticket is collected (embedder.hip: submit_locked / collect), and a
            # non-device failure (a token id out of range in ONE batch) does not poison the engine: without this drain
            # the tickets still in flight would strand their slots and every later call would fail with "every
            # submission slot is in flight" on an engine that still looks healthy.
            self.abandon([t for t, _ in inflight])
            raise
        return normalize_l2_rows(out) if normalize else out


# ---- the stage contract of the reference's index pipeline ------------------------------------------------
@dataclass
class PreparedEmbedding:
    """`PreparedEmbedding` (src/cli/pipeline/types.rs): what `prepare_for_embedding` hands the embed stage."""
    cached: list = field(default_factory=list)        # [(chunk, embedding)] reused from the caches
    to_embed: list = field(default_factory=list)      # chunks that need a forward
    tokens: list = field(default_factory=list)        # tokenised NL text of each `to_embed` chunk


@dataclass
class EmbeddedBatch:
    """`EmbeddedBatch` (types.rs): chunk/embedding pairs for the store stage."""
    chunk_embeddings: list
    cached_count: int


class GpuEmbedStage:
    """`gpu_embed_stage` (embedding.rs:226-421) over the HIP engine: every prepared batch becomes one
    `EmbeddedBatch` on `embed_tx`; if the GPU path raises (`EmbedderError`, i.e. `InferenceFailed`), the chunks are
    counted into `gpu_failures`, the cached pairs still go to the writer and the uncached chunks are requeued on
    `fail_tx` for the CPU stage (`flush_to_cpu`, :160-223) - the stage itself never raises for device trouble."""

    def __init__(self, pipeline: EmbedPipeline, embed_tx: Callable[[EmbeddedBatch], None],
                 fail_tx: Callable[[list], None]):
        self.pipeline = pipeline
        self.embed_tx = embed_tx
        self.fail_tx = fail_tx
        self.gpu_failures = 0
        self.embedded_count = 0

    def run(self, prepared_batches) -> None:
        for prep in prepared_batches:
            if not prep.to_embed:                                   # all cached: straight through (:262-281)
                self.embedded_count += len(prep.cached)
                self.embed_tx(EmbeddedBatch(list(prep.cached), len(prep.cached)))
                continue
            try:
                embs = self.pipeline.embed_token_lists(prep.tokens)
            except EmbedderError:
                self.gpu_failures += len(prep.to_embed)             # :404-421
                if prep.cached:                                     # flush_to_cpu: cached first ...
                    self.embedded_count += len(prep.cached)
                    self.embed_tx(EmbeddedBatch(list(prep.cached), len(prep.cached)))
                self.fail_tx(list(prep.to_embed))                   # ... then requeue the rest to the CPU stage
                continue
            pairs = list(prep.cached) + list(zip(prep.to_embed, embs))
            self.embedded_count += len(pairs)
            self.embed_tx(EmbeddedBatch(pairs, len(prep.cached)))
