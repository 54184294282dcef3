#!/usr/bin/env python3
"""Where does wall - device go for one 512-token sequence?  Times submit and collect separately."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tools.embed_two_streams_lib import make_engine
e, cfg = make_engine(0)
rng = np.random.default_rng(3)
for n in (128, 256, 384, 512, 1024):
    ids = rng.integers(1, 262144, size=(1, n)).astype(np.int64); mask = np.ones((1, n), np.int64)
    for _ in range(5): e.run(ids, mask)
    ts, tc = 0.0, 0.0
    for _ in range(20):
        t0 = time.perf_counter(); t = e.submit(ids, mask); t1 = time.perf_counter(); e.collect(t, 1); t2 = time.perf_counter()
        ts += t1 - t0; tc += t2 - t1
    print("tokens %4d: submit %.3f ms, collect %.3f ms, device %.3f ms" % (n, ts / 20 * 1e3, tc / 20 * 1e3, e.last_ms()), flush=True)
