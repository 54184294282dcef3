#!/usr/bin/env python3
"""Throughput of the two BERT-family engines at their real geometry (seeded weights, synthetic token ids):
SPLADE encode (BERT-base masked-LM + pooling) and reranker scoring (MiniLM-L6 cross-encoder)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import bert_ref as R
from cqs_amd import _lib
from cqs_amd.splade import HipBertEngine, bert_config

def engine(cfg, head, seed):
    kind = _lib.BERT_HEAD_MLM if head == "mlm" else _lib.BERT_HEAD_CLASSIFIER
    c = bert_config(kind)
    e = HipBertEngine(c)
    e.set_weights(R.seeded_weights(cfg, head, seed=seed))
    return e

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--len", type=int, default=256)
    a = ap.parse_args()
    rng = np.random.default_rng(0)
    cfg = R.splade_base()
    e = engine(cfg, "mlm", 1)
    seqs = [rng.integers(1, cfg.vocab_size, size=a.len).astype(np.int32) for _ in range(a.batch)]
    e.splade_dense(seqs)
    t0 = time.perf_counter()
    for _ in range(a.iters):
        e.splade_dense(seqs)
    dt = (time.perf_counter() - t0) / a.iters
    print(f"splade: batch={a.batch} len={a.len} ms={dt*1e3:.2f} docs/s={a.batch/dt:.0f} tokens/s={a.batch*a.len/dt:.0f}", flush=True)
    e.close()
    cfg = R.minilm_l6()
    e = engine(cfg, "classifier", 2)
    seqs = [rng.integers(1, cfg.vocab_size, size=512).astype(np.int32) for _ in range(32)]
    tt = [np.r_[np.zeros(16, np.int32), np.ones(496, np.int32)] for _ in range(32)]
    e.rerank_logits(seqs, tt)
    t0 = time.perf_counter()
    for _ in range(a.iters):
        e.rerank_logits(seqs, tt)
    dt = (time.perf_counter() - t0) / a.iters
    print(f"rerank: batch=32 len=512 ms={dt*1e3:.2f} pairs/s={32/dt:.0f}", flush=True)
    e.close()

if __name__ == "__main__":
    main()
