#!/bin/bash
# Forward time of ragged (log-normal) batches of several sizes under the two attention layouts (run on the GPU box)
for b in 4 8 12 16 24 32 48; do
  for L in per-head shared; do
    CQS_HIP_ATT_LAYOUT=$L python tools/embed_bench.py --iters 5 --vocab 8192 --lognormal --batch $b 2>/dev/null | sed "s/^/layout=$L /" | cut -c1-110
  done
done
