#!/usr/bin/env python3
"""One query length through the blocking call, many times (for `rocprofv3 --kernel-trace --stats`): per-kernel time of the
search-time chain.  usage: query_profile.py <tokens> [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tools.embed_two_streams_lib import make_engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
e, cfg = make_engine(0)
rng = np.random.default_rng(3)
ids = rng.integers(1, 262144, size=(1, n)).astype(np.int64); mask = np.ones((1, n), np.int64)
for _ in range(6): e.run(ids, mask)
t0 = time.perf_counter(); dms = 0.0
for _ in range(reps):
    e.run(ids, mask); dms += e.last_ms()
dt = (time.perf_counter() - t0) / reps
print("query of %3d tokens: %.3f ms (device %.3f ms)" % (n, dt * 1e3, dms / reps), flush=True)
