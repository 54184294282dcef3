"""cqs_amd — MI355X-native (gfx950) hot path of cqs: exact scan + top-k, embedding forward.

Product code lives here and in cqs_amd/csrc (HIP kernels + C ABI, include/cqs_hip.h).
It never imports anything from oracle/ (test infrastructure).
"""
from .index import (BackendContext, DistanceMetric, HipBackend, HipError, HipIndex, IndexResult,
                    VectorIndex, merge_keys, prepare_index_data, unpack_keys)

__all__ = ["BackendContext", "DistanceMetric", "HipBackend", "HipError", "HipIndex", "IndexResult",
           "VectorIndex", "merge_keys", "prepare_index_data", "unpack_keys"]
