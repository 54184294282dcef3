#!/usr/bin/env python3
"""GEMM tuning aid: times cqs_hip_debug_gemm_ms over the shapes of the embedding forward, per kernel variant
(CQS_HIP_GEMM_TILE is read at every launch).  usage: python tools/gemm_bench.py [variant ...]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cqs_amd import _lib
lib = _lib.load()
f = lib.cqs_hip_debug_gemm_ms
f.restype = C.c_float
f.argtypes = [C.c_uint32] * 4 + [C.c_int32]
shapes = [(16384, 1280, 768, 0), (16384, 768, 768, 0), (16384, 2304, 768, 2), (16384, 768, 1152, 0),
          (10240, 1280, 768, 0), (10240, 768, 768, 0), (10240, 2304, 768, 2), (10240, 768, 1152, 0),
          (65536, 1280, 768, 0), (8192, 4096, 4096, 0), (4096, 4096, 4096, 0)]
variants = sys.argv[1:] or ["auto", "small", "pp:3", "pp:4", "pp:5"]
for rep in range(2):
    for M, N, K, kind in shapes:
        row = []
        for v in variants:
            if v == "auto":
                os.environ.pop("CQS_HIP_GEMM_TILE", None)
            else:
                os.environ["CQS_HIP_GEMM_TILE"] = v
            ms = f(M, N, K, 20, kind)
            row.append("%s %6.1fus %5.0fTF" % (v, ms * 1e3, 2.0 * M * N * K / ms / 1e9))
        print(f"M={M:6d} N={N:5d} K={K:5d} o{kind}: " + " | ".join(row), flush=True)
