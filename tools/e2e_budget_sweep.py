#!/usr/bin/env python3
"""Index-pipeline embedding throughput (log-normal chunk lengths, length-sorted token-budget batches, tickets in flight)
as a function of the token budget per batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tools.embed_two_streams_lib import make_engine
from cqs_amd.pipeline import EmbedPipeline

e, cfg = make_engine(0)
rng = np.random.default_rng(7)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
lens = np.clip(np.exp(rng.normal(np.log(300.0), 0.6, size=n)).astype(int), 8, cfg.max_seq)
chunks = [rng.integers(1, cfg.vocab_size, size=int(L)).astype(np.int64) for L in lens]
for budget in (8192, 16384, 24576, 32768, 49152, 16384):
    pipe = EmbedPipeline(e, token_budget=budget, max_seqs=4096)
    pipe.embed_token_lists(chunks[:512])
    t0 = time.perf_counter()
    out = pipe.embed_token_lists(chunks)
    dt = time.perf_counter() - t0
    print("token budget %6d: %.0f chunks/s  %.2f M tokens/s  batches %d" % (budget, n / dt, lens.sum() / dt / 1e6, pipe.stats().get("batches", -1)), flush=True)
