#!/usr/bin/env python3
"""The sparse index's single-query and batched entry points on a 1M-chunk corpus (for kernel traces): 40 single queries at
k = 500, then 8 batches of 32."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cqs_amd import synth
from cqs_amd.splade_index import HipSpladeIndex

off, tok, w = synth.sparse_corpus(1_000_000, 30522)
h = HipSpladeIndex.build_from_csr(None, off, tok, w)
qs = synth.sparse_queries(64, 64, 30522, seed=0x5BA2E3)
for qt, qw in qs[:4]:
    h.search_raw(qt, qw, 500)
t0 = time.perf_counter()
for qt, qw in qs[:40]:
    h.search_raw(qt, qw, 500)
print("single: %.4f ms per query" % ((time.perf_counter() - t0) / 40 * 1e3))
h.search_batch_raw(qs[:32], 500)
t0 = time.perf_counter()
for i in range(8):
    h.search_batch_raw(qs[(i % 2) * 32:(i % 2) * 32 + 32], 500)
print("batch of 32: %.4f ms per call" % ((time.perf_counter() - t0) / 8 * 1e3))
h.close()
