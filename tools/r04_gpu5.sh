#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_query_path_gpu.py tests/test_embed_gpu.py tests/test_bert_gpu.py -m gpu -x -q > gpurun_out/r04h_tests.log 2>&1
rc=$?
tail -25 gpurun_out/r04h_tests.log
[ $rc -eq 0 ] || exit $rc
python - <<'PY'
import sys, time, ctypes as C
sys.path.insert(0, '.')
import numpy as np
import bench
from cqs_amd.embedder import HipEmbedEngine, default_config
cfg = default_config(); cfg.vocab_size = 32768
eng = HipEmbedEngine(cfg)
w = bench.seeded_embed_weights(np, cfg)
for k, v in w.items(): eng.set_tensor(k, v)
eng.set_weights({})
eng.warm(128)
print(eng.query_graph_stats())
rng = np.random.default_rng(1)
for n in (8, 16, 32, 48, 64, 65, 80, 96, 97, 112, 128):
    ids = rng.integers(1, cfg.vocab_size, size=(1, n)).astype(np.int64); mask = np.ones((1, n), np.int64)
    out = np.zeros((1, 768), np.float32)
    args = (eng._h, ids.ctypes.data_as(C.c_void_p), mask.ctypes.data_as(C.c_void_p), 1, n, out.ctypes.data_as(C.c_void_p))
    ts = []
    for _ in range(60):
        t0 = time.perf_counter(); eng._lib.cqs_hip_embed(*args); ts.append(time.perf_counter() - t0)
    print(n, "abi_ms %.4f" % (np.median(ts) * 1e3), "dev %.4f" % eng.last_ms())
PY
