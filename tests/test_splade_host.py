"""Host logic of the SPLADE / reranker mirror (cqs_amd/splade.py) with a stand-in engine - no GPU: the threshold filter
of src/splade/mod.rs:1049-1062 (ascending ids, strict `>`, NaN dropped, +Inf kept), truncation to max_seq_len, and the
reranker's sigmoid (src/reranker.rs:516-518)."""
import math

import numpy as np

from cqs_amd.splade import Reranker, SpladeEncoder


class FakeEngine:
    def __init__(self, dense=None, logits=None):
        self.dense, self.logits, self.seen = dense, logits, None

    def splade_dense(self, seqs):
        self.seen = [np.asarray(s) for s in seqs]
        return self.dense[: len(seqs)]

    def rerank_logits(self, seqs, type_ids):
        self.seen = ([np.asarray(s) for s in seqs], None if type_ids is None else [np.asarray(t) for t in type_ids])
        return self.logits[: len(seqs)]


def test_sparse_vectors_from_dense_activations():
    dense = np.array([[0.0, 0.5, 0.01, np.nan, np.inf, 0.011],
                      [0.0, 0.0, 0.0, 0.0, 0.0, 0.0],
                      [2.0, 0.0099, 0.3, 0.0, 1e-9, 0.01]], np.float32)
    enc = SpladeEncoder(FakeEngine(dense=dense), threshold=0.01, max_seq_len=4)
    sv = enc.encode_batch([[1, 2, 3, 4, 5, 6], [], [7]])
    assert [i for i, _ in sv[0]] == [1, 4, 5]                      # 0.01 is not > 0.01; NaN > t is false; +Inf survives
    assert sv[0][0][1] == 0.5 and math.isinf(sv[0][1][1]) and abs(sv[0][2][1] - 0.011) < 1e-7
    assert sv[1] == []
    assert [i for i, _ in sv[2]] == [0, 2]
    arrays = enc.encode_batch_arrays([[1, 2, 3, 4, 5, 6], [], [7]])
    assert arrays[0][0].dtype == np.uint32 and arrays[0][0].tolist() == [1, 4, 5] and arrays[1][0].size == 0
    assert [len(s) for s in enc.engine.seen] == [4, 0, 1]          # truncated to max_seq_len before the forward
    assert enc.encode([9, 9]) == sv[0] and enc.encode_batch([]) == []


def test_reranker_sigmoid_and_truncation():
    logits = np.array([[0.0, 9.0], [2.0, 9.0], [-3.0, 9.0]], np.float32)      # stride 2: only column 0 counts
    rr = Reranker(FakeEngine(logits=logits), max_length=3)
    s = rr.scores([[1, 2, 3, 4], [5], [6, 7]], [[0, 0, 1, 1], [0], [0, 1]])
    assert np.allclose(s, [0.5, 1 / (1 + math.exp(-2.0)), 1 / (1 + math.exp(3.0))], atol=1e-7) and s.dtype == np.float32
    ids, types = rr.engine.seen
    assert [len(x) for x in ids] == [3, 1, 2] and [len(x) for x in types] == [3, 1, 2]
    assert rr.scores([]).size == 0
    assert rr.scores([[1]], None).shape == (1,) and rr.engine.seen[1] is None
