#!/usr/bin/env python3
"""Rerank of 20 passages x10 + SPLADE query of 16 tokens x10 (for a kernel trace)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import bert_ref as R
from cqs_amd import _lib
from cqs_amd.splade import HipBertEngine, bert_config
rng = np.random.default_rng(0)
e = HipBertEngine(bert_config(_lib.BERT_HEAD_CLASSIFIER)); e.set_weights(R.seeded_weights(R.minilm_l6(), "classifier", seed=2))
seqs = [rng.integers(1, 30522, size=int(l)).astype(np.int32) for l in rng.integers(80, 400, size=20)]
tt = [np.r_[np.zeros(12, np.int32), np.ones(len(s) - 12, np.int32)] for s in seqs]
for _ in range(10): e.rerank_logits(seqs, tt)
e.close()
e = HipBertEngine(bert_config(_lib.BERT_HEAD_MLM)); e.set_weights(R.seeded_weights(R.splade_base(), "mlm", seed=1))
q = [rng.integers(1, 30522, size=16).astype(np.int32)]
for _ in range(10): e.splade_sparse(q, 1.5)
e.close()
