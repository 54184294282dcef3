"""ctypes binding of libcqs_hip.so — the C ABI declared in include/cqs_hip.h.

The product path has no CPU fallback: if the HIP library is missing this module
raises (callers on a GPU box must fail loudly, never fall back to the oracle).
"""
from __future__ import annotations

import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
# CQS_HIP_LIB overrides the library file (kernel-variant A/B runs); default = the in-tree build
LIB_PATH = os.environ.get("CQS_HIP_LIB") or os.path.join(_HERE, "libcqs_hip.so")

OK = 0
ERR_INVALID = -1
ERR_DEVICE = -2
ERR_NOMEM = -3
ERR_POISONED = -4
ERR_NO_DEVICE = -5

METRIC_COSINE = 0
METRIC_DOT = 1
MODE_RAW = 0
MODE_PIPELINE = 1
MAX_K = 1024
NEIGHBORS_MAX = 100

_c_idx = C.c_void_p
_pp = C.POINTER

# (name, restype, argtypes) — one row per symbol include/cqs_hip.h declares.
SIGNATURES = [
    ("cqs_hip_version", C.c_char_p, []),
    ("cqs_hip_device_count", C.c_int32, []),
    ("cqs_hip_device_mem", C.c_int32, [C.c_int32, _pp(C.c_uint64), _pp(C.c_uint64)]),
    ("cqs_hip_index_create", C.c_int32,
     [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int32, C.c_uint64, _pp(_c_idx)]),
    ("cqs_hip_index_create_device", C.c_int32,
     [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int32, C.c_uint64, C.c_int32, _pp(_c_idx)]),
    ("cqs_hip_index_create_sharded", C.c_int32,
     [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint64, _pp(_c_idx)]),
    ("cqs_hip_index_load_sharded", C.c_int32,
     [C.c_char_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint32, C.c_uint64, _pp(_c_idx)]),
    ("cqs_hip_index_shards", C.c_uint32, [_c_idx]),
    ("cqs_hip_index_shard_info", C.c_int32,
     [_c_idx, C.c_uint32, _pp(C.c_int32), _pp(C.c_uint64), _pp(C.c_uint64), _pp(C.c_int32)]),
    ("cqs_hip_index_extend", C.c_int32, [_c_idx, C.c_void_p, C.c_uint64]),
    ("cqs_hip_index_destroy", None, [_c_idx]),
    ("cqs_hip_index_save", C.c_int32, [_c_idx, C.c_char_p, _pp(C.c_uint64)]),
    ("cqs_hip_index_load", C.c_int32, [C.c_char_p, C.c_uint32, C.c_uint64, C.c_int32, C.c_uint64, _pp(_c_idx)]),
    ("cqs_hip_index_len", C.c_uint64, [_c_idx]),
    ("cqs_hip_index_dim", C.c_uint32, [_c_idx]),
    ("cqs_hip_index_metric", C.c_uint32, [_c_idx]),
    ("cqs_hip_index_max_k", C.c_uint32, [_c_idx]),
    ("cqs_hip_index_poisoned", C.c_int32, [_c_idx]),
    ("cqs_hip_index_device", C.c_int32, [_c_idx]),
    ("cqs_hip_index_row_base", C.c_uint64, [_c_idx]),
    ("cqs_hip_index_last_error", C.c_size_t, [_c_idx, C.c_char_p, C.c_size_t]),
    ("cqs_hip_index_search", C.c_int32,
     [_c_idx, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_float,
      C.c_void_p, C.c_void_p, C.c_void_p]),
    ("cqs_hip_index_search_device", C.c_int32,
     [_c_idx, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_float,
      C.c_void_p, C.c_void_p, C.c_void_p]),
    ("cqs_hip_index_combine_stats", None, [_c_idx, _pp(C.c_uint64), _pp(C.c_uint64)]),
    ("cqs_hip_index_neighbors", C.c_int32, [_c_idx, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, _pp(C.c_uint32)]),
    ("cqs_hip_unpack_keys", None, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    ("cqs_hip_merge_keys", C.c_size_t,
     [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p]),
    ("cqs_hip_embed_config_default", None, [C.c_void_p]),
    ("cqs_hip_embedder_create", C.c_int32, [C.c_void_p, C.c_int32, _pp(_c_idx)]),
    ("cqs_hip_embedder_set_tensor", C.c_int32, [_c_idx, C.c_char_p, C.c_void_p, C.c_uint64]),
    ("cqs_hip_embedder_finalize", C.c_int32, [_c_idx]),
    ("cqs_hip_embedder_load_dir", C.c_int32, [C.c_char_p, C.c_void_p, C.c_int32, _pp(_c_idx)]),
    ("cqs_hip_embedder_destroy", None, [_c_idx]),
    ("cqs_hip_embedder_dim", C.c_uint32, [_c_idx]),
    ("cqs_hip_embedder_max_seq", C.c_uint32, [_c_idx]),
    ("cqs_hip_embedder_poisoned", C.c_int32, [_c_idx]),
    ("cqs_hip_embedder_last_error", C.c_size_t, [_c_idx, C.c_char_p, C.c_size_t]),
    ("cqs_hip_embed", C.c_int32, [_c_idx, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    ("cqs_hip_embed_submit", C.c_int32, [_c_idx, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, _pp(C.c_uint64)]),
    ("cqs_hip_embed_submit_ragged", C.c_int32, [_c_idx, C.c_void_p, C.c_void_p, C.c_uint32, _pp(C.c_uint64)]),
    ("cqs_hip_embed_collect", C.c_int32, [_c_idx, C.c_uint64, C.c_void_p]),
    ("cqs_hip_embedder_warm", C.c_int32, [_c_idx, C.c_uint32]),
    ("cqs_hip_embedder_query_graph_stats", None, [_c_idx, _pp(C.c_uint64), _pp(C.c_uint64), _pp(C.c_uint64), _pp(C.c_uint64)]),
    ("cqs_hip_normalize_l2_rows", None, [C.c_void_p, C.c_uint64, C.c_uint32]),
    ("cqs_hip_embed_hidden", C.c_int32, [_c_idx, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    ("cqs_hip_embedder_last_ms", C.c_float, [_c_idx]),
    ("cqs_hip_index_set_timing", None, [_c_idx, C.c_int32]),
    ("cqs_hip_index_scan_time", C.c_int32, [_c_idx, _pp(C.c_uint32), _pp(C.c_double)]),
    # BERT-family auxiliary models (SPLADE encoder / cross-encoder reranker)
    ("cqs_hip_bert_config_default", C.c_int32, [C.c_uint32, C.c_void_p]),
    ("cqs_hip_bert_create", C.c_int32, [C.c_void_p, C.c_int32, _pp(_c_idx)]),
    ("cqs_hip_bert_set_tensor", C.c_int32, [_c_idx, C.c_char_p, C.c_void_p, C.c_uint64]),
    ("cqs_hip_bert_finalize", C.c_int32, [_c_idx]),
    ("cqs_hip_bert_load_dir", C.c_int32, [C.c_char_p, C.c_void_p, C.c_int32, _pp(_c_idx)]),
    ("cqs_hip_bert_destroy", None, [_c_idx]),
    ("cqs_hip_splade_encode", C.c_int32, [_c_idx, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    ("cqs_hip_splade_encode_sparse", C.c_int32,
     [_c_idx, C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("cqs_hip_splade_submit_sparse", C.c_int32, [_c_idx, C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, C.c_uint32, C.c_void_p]),
    ("cqs_hip_splade_collect_sparse", C.c_int32, [_c_idx, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("cqs_hip_bert_embed_submit", C.c_int32, [_c_idx, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    ("cqs_hip_bert_embed_collect", C.c_int32, [_c_idx, C.c_uint64, C.c_void_p]),
    ("cqs_hip_rerank_logits", C.c_int32, [_c_idx, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    ("cqs_hip_bert_embed", C.c_int32, [_c_idx, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    ("cqs_hip_bert_hidden", C.c_int32, [_c_idx, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    ("cqs_hip_bert_vocab", C.c_uint32, [_c_idx]),
    ("cqs_hip_bert_poisoned", C.c_int32, [_c_idx]),
    ("cqs_hip_bert_last_error", C.c_size_t, [_c_idx, C.c_char_p, C.c_size_t]),
    ("cqs_hip_sparse_index_create", C.c_int32,
     [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int32, _pp(_c_idx)]),
    ("cqs_hip_sparse_index_create_inverted", C.c_int32,
     [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_int32, _pp(_c_idx)]),
    ("cqs_hip_sparse_index_save", C.c_int32, [_c_idx, C.c_char_p, C.c_uint64, _pp(C.c_uint64)]),
    ("cqs_hip_sparse_index_load", C.c_int32, [C.c_char_p, C.c_uint64, C.c_uint64, C.c_int32, _pp(_c_idx)]),
    ("cqs_hip_sparse_index_destroy", None, [_c_idx]),
    ("cqs_hip_sparse_index_len", C.c_uint64, [_c_idx]),
    ("cqs_hip_sparse_index_unique_tokens", C.c_uint64, [_c_idx]),
    ("cqs_hip_sparse_index_postings", C.c_uint64, [_c_idx]),
    ("cqs_hip_sparse_index_search", C.c_int32,
     [_c_idx, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, _pp(C.c_uint32)]),
    ("cqs_hip_sparse_index_search_batch", C.c_int32,
     [_c_idx, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("cqs_hip_sparse_index_combine_stats", None, [_c_idx, _pp(C.c_uint64), _pp(C.c_uint64)]),
    ("cqs_hip_sparse_index_last_search", C.c_int32, [_c_idx, _pp(C.c_float), _pp(C.c_uint64)]),
    ("cqs_hip_sparse_index_poisoned", C.c_int32, [_c_idx]),
    ("cqs_hip_sparse_index_last_error", C.c_size_t, [_c_idx, C.c_char_p, C.c_size_t]),
]

BERT_HEAD_MLM = 0
BERT_HEAD_CLASSIFIER = 1
BERT_HEAD_NONE = 2


class BertConfig(C.Structure):
    """`cqs_hip_bert_config` (include/cqs_hip.h)."""
    _fields_ = [("vocab_size", C.c_uint32), ("hidden", C.c_uint32), ("layers", C.c_uint32), ("heads", C.c_uint32),
                ("intermediate", C.c_uint32), ("max_pos", C.c_uint32), ("type_vocab", C.c_uint32),
                ("num_labels", C.c_uint32), ("head", C.c_uint32), ("ln_eps", C.c_float)]



class EmbedConfig(C.Structure):
    """`cqs_hip_embed_config` (include/cqs_hip.h)."""
    _fields_ = [("vocab_size", C.c_uint32), ("hidden", C.c_uint32), ("layers", C.c_uint32), ("heads", C.c_uint32),
                ("kv_heads", C.c_uint32), ("head_dim", C.c_uint32), ("intermediate", C.c_uint32),
                ("dense_hidden", C.c_uint32), ("sliding_window", C.c_uint32), ("sliding_pattern", C.c_uint32),
                ("max_seq", C.c_uint32), ("rms_eps", C.c_float), ("rope_theta_global", C.c_float),
                ("rope_theta_local", C.c_float), ("query_pre_attn_scalar", C.c_float)]


_lib = None


class HipLibraryMissing(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load libcqs_hip.so (built in-tree by __graft_entry__.build / `make -C cqs_amd/csrc`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C cqs_amd/csrc`. There is no CPU fallback for the cqs_amd product path.")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 (same soname
    # as /opt/rocm's).  If the system copy is loaded first and torch's afterwards, torch finds
    # no GPU; loading torch first makes both share torch's runtime.  Consumers without torch
    # (the Rust shim) simply bind the system ROCm runtime.
    if "torch" not in sys.modules and not os.environ.get("CQS_HIP_NO_TORCH_PRELOAD"):
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    lib = C.CDLL(LIB_PATH)
    for name, res, args in SIGNATURES:
        fn = getattr(lib, name)  # AttributeError if the header and the library disagree
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
