#!/usr/bin/env python3
"""Sharded handle on ONE device ([0,0,0,0]): a few blocking searches, for a rocprofv3 kernel trace (tools/r04_shard_trace.sh)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

def main():
    from cqs_amd import HipIndex
    n, dim, k = 1_000_000, 768, 20
    rng = np.random.default_rng(7)
    rows = rng.standard_normal((n, dim), dtype=np.float32)
    rows /= np.linalg.norm(rows, axis=1, keepdims=True)
    q = rows[rng.integers(0, n, size=64)] + 0.05 * rng.standard_normal((64, dim), dtype=np.float32)
    devs = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0,0,0,0").split(",")]
    sh = HipIndex.build_sharded(None, rows, devs)
    single = HipIndex.build_from_flat(None, rows)
    for name, ix in (("sharded", sh), ("single", single)):
        if ix is None: continue
        for i in range(10): ix.search_batch(q[i], k)
        t0 = time.perf_counter()
        for i in range(40): ix.search_batch(q[i % 64], k)
        print(name, "ms/query %.4f" % ((time.perf_counter() - t0) / 40 * 1e3))
    sh.close()

if __name__ == "__main__":
    main()
