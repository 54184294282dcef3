#!/bin/bash
# LDS counters of the attention kernel on the 32 x 512 forward.
OUT=$PWD/gpurun_out; REPO=$PWD; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/att_pmc -o pmc --output-format csv -- python3 $REPO/tools/embed_bench.py --iters 3 > /dev/null 2> $OUT/att_pmc.err
cd $REPO
python3 tools/summarize_prof.py $OUT/att_pmc | grep -E "attention|^#"
tail -3 $OUT/att_pmc.err
