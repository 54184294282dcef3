#!/bin/bash
# Ablation variants of the resident-key BERT attention kernel (build/variants/lib_batt_<name>.so).
#   tools/bert_attn_ablate.sh build NAME "-DFLAG ..."   (here, no GPU needed)      tools/bert_attn_ablate.sh run   (GPU box)
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  mkdir -p build/variants
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $3 -c cqs_amd/csrc/bert_kernels.hip -o build/variants/batt_$2.o
  objs=$(ls cqs_amd/csrc/build/*.o | grep -v -e bert_kernels.o -e amdgcn)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o build/variants/lib_batt_$2.so build/variants/batt_$2.o $objs -ldl
  rm build/variants/batt_$2.o
else
  OUT=$PWD/gpurun_out; REPO=$PWD; mkdir -p $OUT
  cd /tmp && export TMPDIR=/tmp
  shopt -s nullglob   # no variant built: the loop runs the default library only (an unmatched glob used to become a file name)
  for f in $REPO/cqs_amd/libcqs_hip.so $REPO/build/variants/lib_batt_*.so; do
    n=$(basename $f .so)
    export CQS_HIP_LIB=$f
    timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/babl_$n -o kt --output-format csv -- python3 $REPO/tools/bert_bench.py --iters 3 > $OUT/babl_$n.txt 2> $OUT/babl_$n.err
    echo "== $n: $(python3 $REPO/tools/summarize_prof.py $OUT/babl_$n | grep -E 'attention' | head -2 | tr -s ' ' | tr '\n' '|')"
  done
fi
