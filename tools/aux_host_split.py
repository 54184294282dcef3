#!/usr/bin/env python3
"""Where the wall time of a 20-passage rerank goes: Python packing vs the C call."""
import os, sys, time, ctypes as C
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
    toks, lens = e._pack(seqs); t2, _ = e._pack(tt)
    out = np.empty((npass, 1), np.float32)
    f = e._lib.cqs_hip_rerank_logits
    for _ in range(5): f(e._h, toks.ctypes.data_as(C.c_void_p), t2.ctypes.data_as(C.c_void_p), lens.ctypes.data_as(C.c_void_p), npass, out.ctypes.data_as(C.c_void_p))
    t0 = time.perf_counter()
    for _ in range(30): f(e._h, toks.ctypes.data_as(C.c_void_p), t2.ctypes.data_as(C.c_void_p), lens.ctypes.data_as(C.c_void_p), npass, out.ctypes.data_as(C.c_void_p))
    c_ms = (time.perf_counter() - t0) / 30 * 1e3
    t0 = time.perf_counter()
    for _ in range(30): e.rerank_logits(seqs, tt)
    w_ms = (time.perf_counter() - t0) / 30 * 1e3
    print("rerank %3d passages: C call %.3f ms, python wrapper total %.3f ms" % (npass, c_ms, w_ms), flush=True)
