#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_embed_gpu.py tests/test_random_sweep_gpu.py tests/test_pipeline.py tests/test_e2e_gpu.py -m gpu -x -q > gpurun_out/r04g_tests.log 2>&1
rc=$?
tail -15 gpurun_out/r04g_tests.log
[ $rc -eq 0 ] || exit $rc
bash tools/r04_embed_ab.sh CQS_HIP_GEMM_FUSE_NORM_MIN_ROWS=1024 CQS_HIP_GEMM_FUSE_NORM_MIN_ROWS=2048 CQS_HIP_GEMM_FUSE_NORM_MIN_ROWS=4096 CQS_HIP_GEMM_FUSE_NORM_MIN_ROWS=8192 CQS_HIP_GEMM_FUSE_NORM_MIN_ROWS=4096,CQS_HIP_QKV_FUSE=0
