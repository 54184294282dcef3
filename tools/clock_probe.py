#!/usr/bin/env python3
"""Workload for the effective-clock / MFMA-utilisation PMC pass (tools/clock_probe.sh): the 256-query batched scan
over 1M x 768 rows, the 8192 x 4096 x 4096 bf16 GEMM, and the 32 x 512 embedding forward, a few launches each."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cqs_amd import HipIndex, _lib

def main():
    n, dim, B = 1_000_000, 768, 256
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    rows = torch.randn(n, dim, device="cuda", generator=g)
    rows /= rows.norm(dim=1, keepdim=True)
    idx = HipIndex.build_from_device(None, rows.data_ptr(), n, dim, keepalive=rows)
    q = np.random.default_rng(2).standard_normal((B, dim)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    for _ in range(6):
        idx.search_batch(q, 20)
    f = _lib.load().cqs_hip_debug_gemm_ms
    f.restype = C.c_float; f.argtypes = [C.c_uint32] * 4 + [C.c_int32]
    print("gemm ms", f(8192, 4096, 4096, 20, 0))
    from tools.embed_two_streams_lib import make_engine
    e, cfg = make_engine(0)
    rng = np.random.default_rng(1)
    ids = rng.integers(1, 262144, size=(32, 512)).astype(np.int64); mask = np.ones((32, 512), np.int64)
    for _ in range(4):
        e.run(ids, mask)
    print("ok")

if __name__ == "__main__":
    main()
