#!/usr/bin/env python3
"""Per-call wall times of the blocking query call at a few lengths (finds one-off stalls a mean hides)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tools.embed_two_streams_lib import make_engine

e, cfg = make_engine(0)
rng = np.random.default_rng(3)
for n in (8, 16, 32, 16, 8):
    ids = rng.integers(1, 262144, size=(1, n)).astype(np.int64); mask = np.ones((1, n), np.int64)
    ts = []
    for _ in range(40):
        t0 = time.perf_counter(); e.run(ids, mask); ts.append((time.perf_counter() - t0) * 1e3)
    print("T=%2d: " % n + " ".join("%.2f" % t for t in ts), flush=True)
