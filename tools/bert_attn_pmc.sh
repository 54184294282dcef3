#!/bin/bash
# LDS / VALU / MFMA counters of the BERT attention kernels (SPLADE 64 x 256 + rerank 32 x 512), two passes.
OUT=$PWD/gpurun_out; REPO=$PWD; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/bert_attn_pmc1 -o pmc --output-format csv -- python3 $REPO/tools/bert_bench.py --iters 2 > /dev/null 2> $OUT/bert_attn_pmc1.err &&
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM -d $OUT/bert_attn_pmc2 -o pmc --output-format csv -- python3 $REPO/tools/bert_bench.py --iters 2 > /dev/null 2> $OUT/bert_attn_pmc2.err
cd $REPO
python3 tools/summarize_prof.py $OUT/bert_attn_pmc1 $OUT/bert_attn_pmc2 | grep -E "attention|^#"
tail -3 $OUT/bert_attn_pmc1.err $OUT/bert_attn_pmc2.err
