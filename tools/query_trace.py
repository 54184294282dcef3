#!/usr/bin/env python3
"""One 32-token query, 10 times (for a kernel trace of the search-time embedding path)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tools.embed_two_streams_lib import make_engine
e, cfg = make_engine(0)
ids = np.random.default_rng(3).integers(1, 262144, size=(1, 32)).astype(np.int64); mask = np.ones((1, 32), np.int64)
for _ in range(10): e.run(ids, mask)
