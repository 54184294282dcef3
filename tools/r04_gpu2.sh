#!/bin/bash
# round 4: pair-split fused projection kernel - parity, A/B, kernel trace
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_embed_gpu.py tests/test_random_sweep_gpu.py tests/test_pipeline.py -m gpu -x -q > gpurun_out/r04b_tests.log 2>&1
rc=$?
tail -5 gpurun_out/r04b_tests.log
[ $rc -eq 0 ] || exit $rc
bash tools/r04_embed_ab.sh CQS_HIP_GEMM_FUSE_NORM=2 CQS_HIP_GEMM_FUSE_NORM=1 CQS_HIP_GEMM_FUSE_NORM=0 CQS_HIP_GEMM_FUSE_NORM=2,CQS_HIP_GEMM_FUSE_NORM_MIN_ROWS=4096 || exit 1
cd /tmp && export TMPDIR=/tmp
for f in 2 1; do
  CQS_HIP_GEMM_FUSE_NORM=$f CQS_HIP_EMBED_CONTEXTS=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r04b_prof_f$f" -- python3 "$GRAFT_REPO_ROOT/bench.py" --rows 20000 --steps 5 --warmup 2 --extras 0 --e2e-chunks 0 --cpu-seconds 0 --embed-steps 8 --abi-devices "" > /dev/null 2>&1 || exit 1
  python3 "$GRAFT_REPO_ROOT/tools/summarize_prof.py" "$GRAFT_REPO_ROOT/gpurun_out/r04b_prof_f$f" > "$GRAFT_REPO_ROOT/gpurun_out/r04b_prof_f$f.txt"
  grep -E "rowfuse|gemm_pp|attention|kv_prep|add_norm" "$GRAFT_REPO_ROOT/gpurun_out/r04b_prof_f$f.txt" | head -12
  rm -rf "$GRAFT_REPO_ROOT/gpurun_out/r04b_prof_f$f"
done
