#!/bin/bash
# Forward time of ragged batches under forced GEMM tile kinds (planner check; run on the GPU box)
for b in 16 32 48; do
  for T in default small pp:3 pp:4 pp:5; do
    if [ $T = default ]; then unset CQS_HIP_GEMM_TILE; else export CQS_HIP_GEMM_TILE=$T; fi
    python tools/embed_bench.py --iters 5 --vocab 8192 --lognormal --batch $b 2>/dev/null | sed "s/^/tile=$T /" | cut -c1-100
  done
done
