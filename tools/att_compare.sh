#!/bin/bash
# Kernel-trace timing of the two head-sharing attention kernels on the 32 x 512 forward (same box, back to back).
OUT=$PWD/gpurun_out
REPO=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for k in ${@:-dma reg}; do
  export CQS_HIP_ATT_KERNEL=$k
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/att_$k -o kt --output-format csv -- python3 $REPO/tools/embed_bench.py --iters 4 > $OUT/att_$k.txt 2> $OUT/att_$k.err
  echo "== $k: $(cat $OUT/att_$k.txt)"
  python3 $REPO/tools/summarize_prof.py $OUT/att_$k | head -16 
done
