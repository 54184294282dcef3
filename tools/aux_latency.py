#!/usr/bin/env python3
"""Search-time latency of the BERT-family engines: one short SPLADE query; a rerank of 20 / 100 passages."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import bert_ref as R
from cqs_amd import _lib
from cqs_amd.splade import HipBertEngine, bert_config

rng = np.random.default_rng(0)
e = HipBertEngine(bert_config(_lib.BERT_HEAD_CLASSIFIER)); e.set_weights(R.seeded_weights(R.minilm_l6(), "classifier", seed=2))
for npass in (20, 100):
    seqs = [rng.integers(1, 30522, size=int(l)).astype(np.int32) for l in rng.integers(80, 400, size=npass)]
    tt = [np.r_[np.zeros(12, np.int32), np.ones(len(s) - 12, np.int32)] for s in seqs]
    for _ in range(5): e.rerank_logits(seqs, tt)
    ts = []
    for _ in range(30):
        t0 = time.perf_counter(); e.rerank_logits(seqs, tt); ts.append(time.perf_counter() - t0)
    print("rerank of %3d passages (%d tokens): median %.3f ms" % (npass, sum(len(s) for s in seqs), float(np.median(ts)) * 1e3), flush=True)
e.close()

e = HipBertEngine(bert_config(_lib.BERT_HEAD_MLM)); e.set_weights(R.seeded_weights(R.splade_base(), "mlm", seed=1))
for n in (8, 32):
    q = [rng.integers(1, 30522, size=n).astype(np.int32)]
    for _ in range(5): e.splade_sparse(q, 1.5)
    ts = []
    for _ in range(30):
        t0 = time.perf_counter(); e.splade_sparse(q, 1.5); ts.append(time.perf_counter() - t0)
    print("splade query of %2d tokens: median %.3f ms" % (n, float(np.median(ts)) * 1e3), flush=True)
e.close()
