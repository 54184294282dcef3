#!/usr/bin/env python3
"""Search-time latency of the embedding forward: ONE short query through the blocking call (what `embed_query` costs
before `VectorIndex::search`), full EmbeddingGemma-300m geometry."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tools.embed_two_streams_lib import make_engine

e, cfg = make_engine(0)
rng = np.random.default_rng(3)
for n in (8, 16, 32, 64, 128, 512):
    ids = rng.integers(1, 262144, size=(1, n)).astype(np.int64); mask = np.ones((1, n), np.int64)
    for _ in range(5): e.run(ids, mask)
    t0 = time.perf_counter()
    for _ in range(30): e.run(ids, mask)
    dt = (time.perf_counter() - t0) / 30
    print("query of %3d tokens: %.3f ms (device %.3f ms)" % (n, dt * 1e3, e.last_ms()), flush=True)
