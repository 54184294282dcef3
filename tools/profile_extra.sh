#!/bin/bash
# usage: tools/profile_extra.sh <tag>
# Kernel traces (and, for the batched scans, pipe-busy / FETCH_SIZE counter passes - separate runs) of what
# tools/profile_round.sh does not cover: the batched scans at 256 / 64 / 32 queries, the scan + select at k = 500, the sparse
# index's single-query and batched paths.  Outputs under gpurun_out/<tag>_*; summaries in gpurun_out/<tag>_extra_summary.txt.
TAG=${1:-r05}
OUT=$PWD/gpurun_out
REPO=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/${TAG}_extra_summary.txt
run_kt() {   # name, command...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_x_$name -o kt --output-format csv -- "$@" > $OUT/${TAG}_x_$name.out 2> $OUT/${TAG}_x_$name.err
  ( cd $REPO; echo "## $name: $*"; cat $OUT/${TAG}_x_$name.out; python3 tools/summarize_prof.py $OUT/${TAG}_x_$name | grep -v "^$" | head -14; echo ) >> $OUT/${TAG}_extra_summary.txt
  rm -rf $OUT/${TAG}_x_$name
}
run_pmc() {  # name, counters, command...
  local name=$1 ctr=$2; shift; shift
  timeout -k 10 300 rocprofv3 --pmc $ctr -d $OUT/${TAG}_p_$name -o pmc --output-format csv -- "$@" > /dev/null 2> $OUT/${TAG}_p_$name.err
  ( cd $REPO; echo "## $name counters ($ctr): $*"; python3 tools/summarize_prof.py $OUT/${TAG}_p_$name | grep "scan_mfma" ; echo ) >> $OUT/${TAG}_extra_summary.txt
  rm -rf $OUT/${TAG}_p_$name
}
for b in 256 64 32; do
  run_kt scan_b$b python3 $REPO/tools/time_scan.py 1000000 $b
  run_pmc scan_b${b}_sq "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE" python3 $REPO/tools/time_scan.py 1000000 $b
  run_pmc scan_b${b}_fetch FETCH_SIZE python3 $REPO/tools/time_scan.py 1000000 $b
done
run_kt scan_k500 python3 $REPO/tools/time_scan.py 1000000 1 500
run_kt sparse python3 $REPO/tools/sparse_batch_bench.py
cd $REPO
cat $OUT/${TAG}_extra_summary.txt
