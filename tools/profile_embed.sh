#!/bin/bash
# rocprofv3 evidence for the embedding forward (32 x 512 tokens, full geometry): kernel trace + PMC passes.
# Usage: tools/profile_embed.sh <tag>     (outputs under gpurun_out/<tag>_*; summary in gpurun_out/<tag>_embed_summary.txt)
TAG=${1:-r02}
OUT=$PWD/gpurun_out
REPO=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $REPO/tools/embed_bench.py --iters 4"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_ekt -o kt --output-format csv -- $CMD > $OUT/${TAG}_embed_under_profiler.txt 2> $OUT/${TAG}_ekt.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/${TAG}_efetch -o pmc --output-format csv -- $CMD > /dev/null 2> $OUT/${TAG}_efetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/${TAG}_ewrite -o pmc --output-format csv -- $CMD > /dev/null 2> $OUT/${TAG}_ewrite.err
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 -d $OUT/${TAG}_esq -o pmc --output-format csv -- $CMD > /dev/null 2> $OUT/${TAG}_esq.err
cd $REPO
python3 tools/summarize_prof.py $OUT/${TAG}_ekt $OUT/${TAG}_efetch $OUT/${TAG}_ewrite $OUT/${TAG}_esq > $OUT/${TAG}_embed_summary.txt
cat $OUT/${TAG}_embed_under_profiler.txt
head -30 $OUT/${TAG}_embed_summary.txt
