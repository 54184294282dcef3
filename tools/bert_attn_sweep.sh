#!/bin/bash
# Kernel trace of the BERT engines' attention kernels under the query-split settings (run on the GPU box):
#   tools/bert_attn_sweep.sh <out_dir>
set -e
out=${1:-gpurun_out/bert_attn}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
for cfg in "256 64" "512 32"; do
  set -- $cfg
  for q in auto 1 2 4 old; do
    d=$out/len$1_q$q
    if [ $q = auto ]; then unset CQS_HIP_BERT_ATTN_QSPLIT; unset CQS_HIP_BERT_ATTN_RESIDENT;
    elif [ $q = old ]; then unset CQS_HIP_BERT_ATTN_QSPLIT; export CQS_HIP_BERT_ATTN_RESIDENT=0;
    else export CQS_HIP_BERT_ATTN_QSPLIT=$q; unset CQS_HIP_BERT_ATTN_RESIDENT; fi
    rocprofv3 --kernel-trace --stats -d $d -o kt --output-format csv -- python3 tools/bert_bench.py --iters 5 --len $1 --batch $2 > $d.log 2>&1
    echo "== len $1 batch $2 qsplit $q"; grep -h "attention" $d/kt_kernel_stats.csv | cut -d, -f1-4 | sed 's/"//g'
    grep -h "splade:\|rerank:" $d.log
  done
done
