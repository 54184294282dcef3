#!/usr/bin/env python3
"""Experiment: two engines x 16 x 512 with the GEMM planner told it has half the chip (CQS_HIP_GEMM_CUS=128)."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tools.embed_two_streams_lib import make_engine

def run(engines, B, L, iters):
    rng = np.random.default_rng(1)
    ids = rng.integers(1, 262144, size=(B, L)).astype(np.int64); mask = np.ones((B, L), np.int64)
    for e in engines: e.run(ids, mask); e.run(ids, mask)
    def work(e):
        pend = []
        for _ in range(iters):
            pend.append(e.submit(ids, mask))
            if len(pend) == 3: e.collect(pend.pop(0), B)
        for t in pend: e.collect(t, B)
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(e,)) for e in engines]
    [t.start() for t in th]; [t.join() for t in th]
    return len(engines) * iters * B / (time.perf_counter() - t0)

if __name__ == '__main__':
    e1, cfg = make_engine(0)
    print("cus=%s one engine 32x512: %.0f" % (os.environ.get("CQS_HIP_GEMM_CUS"), run([e1], 32, 512, 24)), flush=True)
    print("cus=%s one engine 16x512: %.0f" % (os.environ.get("CQS_HIP_GEMM_CUS"), run([e1], 16, 512, 48)), flush=True)
    e2, _ = make_engine(1)
    print("cus=%s two engines 16x512: %.0f" % (os.environ.get("CQS_HIP_GEMM_CUS"), run([e1, e2], 16, 512, 48)), flush=True)
