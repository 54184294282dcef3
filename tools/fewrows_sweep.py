#!/usr/bin/env python3
"""Device time of mid-size batches (1 x N tokens and 8 x N/8) under the current CQS_HIP_GEMM_FEWROWS threshold."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tools.embed_two_streams_lib import make_engine
e, cfg = make_engine(0)
rng = np.random.default_rng(3)
for B, n in ((1, 768), (1, 1024), (1, 2048), (4, 512), (8, 256), (8, 512)):
    ids = rng.integers(1, 262144, size=(B, n)).astype(np.int64); mask = np.ones((B, n), np.int64)
    for _ in range(4): e.run(ids, mask)
    ms = []
    for _ in range(10):
        e.run(ids, mask); ms.append(e.last_ms())
    print("FEWROWS<=%s  %d x %4d tokens: device %.3f ms" % (os.environ.get("CQS_HIP_GEMM_FEWROWS", "512"), B, n, float(np.median(ms))), flush=True)
