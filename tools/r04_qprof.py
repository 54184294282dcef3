#!/usr/bin/env python3
"""One query length through the search-time chain, many times (for rocprofv3 --kernel-trace): python r04_qprof.py T [eager]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from cqs_amd.embedder import HipEmbedEngine, default_config
T = int(sys.argv[1])
cfg = default_config(); cfg.vocab_size = 32768
eng = HipEmbedEngine(cfg)
for k, v in bench.seeded_embed_weights(np, cfg).items():
    eng.set_tensor(k, v)
eng.set_weights({})
ids = np.random.default_rng(1).integers(1, cfg.vocab_size, size=(1, T)).astype(np.int64)
mask = np.ones((1, T), np.int64)
ms = []
for _ in range(40):
    eng.run(ids, mask); ms.append(eng.last_ms())
print("T", T, "device_ms median", float(np.median(ms[5:])), eng.query_graph_stats())
