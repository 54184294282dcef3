#!/usr/bin/env python3
"""Forward throughput of ONE engine with tickets in flight (what the index pipeline does): 32 x 512 batches, depth 1..3."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tools.embed_two_streams_lib import make_engine

def run(e, B, L, iters, depth):
    rng = np.random.default_rng(1)
    ids = rng.integers(1, 262144, size=(B, L)).astype(np.int64); mask = np.ones((B, L), np.int64)
    e.run(ids, mask); e.run(ids, mask)
    pend = []
    t0 = time.perf_counter()
    for _ in range(iters):
        pend.append(e.submit(ids, mask))
        if len(pend) == depth: e.collect(pend.pop(0), B)
    for t in pend: e.collect(t, B)
    return iters * B / (time.perf_counter() - t0)

if __name__ == "__main__":
    e, cfg = make_engine(0)
    for depth in (1, 2, 3, 1, 3):
        print("depth %d: %.0f chunks/s" % (depth, run(e, 32, 512, 24, depth)), flush=True)
