#!/usr/bin/env python3
"""e5-base / bge-large / MiniLM mid-size batches under the current CQS_HIP_GEMM_BIAS_FEWROWS_MH (tokens x hidden) threshold."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import bert_ref as R
from cqs_amd import _lib
from cqs_amd.splade import HipBertEngine, bert_config
rng = np.random.default_rng(0)
for name, cfg in (("e5-base", R.e5_base()), ("bge-large", R.bge_large()), ("minilm", R.minilm_l6())):
    e = HipBertEngine(bert_config(_lib.BERT_HEAD_NONE, hidden=cfg.hidden, layers=cfg.layers, heads=cfg.heads, intermediate=cfg.intermediate))
    e.set_weights(R.seeded_weights(cfg, "none", seed=1))
    for B, n in ((1, 256), (2, 256), (4, 256), (8, 256), (16, 256)):
        seqs = [rng.integers(1, 30522, size=n).astype(np.int32) for _ in range(B)]
        for _ in range(4): e.embed(seqs)
        ts = []
        for _ in range(12):
            t0 = time.perf_counter(); e.embed(seqs); ts.append(time.perf_counter() - t0)
        print("FEWROWS_MH<=%s %-9s %2d x %d tokens: %.3f ms" % (os.environ.get("CQS_HIP_GEMM_BIAS_FEWROWS_MH", "655360"), name, B, n, float(np.median(ts)) * 1e3), flush=True)
    e.close()
