#!/usr/bin/env python3
"""GEMM tuning aid: times cqs_hip_debug_gemm_ms over the shapes of the embedding forward."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cqs_amd import _lib
lib = _lib.load()
f = lib.cqs_hip_debug_gemm_ms
f.restype = C.c_float
f.argtypes = [C.c_uint32] * 4 + [C.c_int32]
shapes = [(16384, 1280, 768, 0), (16384, 768, 768, 0), (16384, 2304, 768, 2), (16384, 768, 1152, 0),
          (65536, 1280, 768, 0), (16384, 1280, 4096, 0), (8192, 4096, 4096, 0)]
for M, N, K, kind in shapes:
    ms = f(M, N, K, 20, kind)
    print(f"M={M:6d} N={N:5d} K={K:5d} out={kind}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:8.1f} TF/s")
