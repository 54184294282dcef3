#!/bin/bash
# usage: tools/pmc_scan_small.sh <queries> [env assignments...]   SQ / LDS counters of the small-block scan kernel (one pass each set)
B=${1:-32}; shift
OUT=$PWD/gpurun_out; REPO=$PWD; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM"; do
  rm -rf $OUT/pmc_small
  env "$@" timeout -k 10 200 rocprofv3 --pmc $set -d $OUT/pmc_small -o pmc --output-format csv -- python3 $REPO/tools/time_scan.py 1000000 $B > /dev/null 2> $OUT/pmc_small.err
  ( cd $REPO; python3 tools/summarize_prof.py $OUT/pmc_small | grep "scan_mfma" )
done
rm -rf $OUT/pmc_small
