#!/bin/bash
# Build ablation variants of the DMA attention kernel (build/variants/lib_att_<name>.so) — run here (no GPU needed).
#   tools/att_ablate.sh build NAME "-DFLAG ..."      tools/att_ablate.sh run   (on the GPU box)
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  mkdir -p build/variants
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $3 -c cqs_amd/csrc/embed_kernels.hip -o build/variants/att_$2.o
  objs=$(ls cqs_amd/csrc/build/*.o | grep -v -e embed_kernels.o -e amdgcn)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o build/variants/lib_att_$2.so build/variants/att_$2.o $objs -ldl
  rm build/variants/att_$2.o
else
  OUT=$PWD/gpurun_out; REPO=$PWD; mkdir -p $OUT
  cd /tmp && export TMPDIR=/tmp
  for f in $REPO/cqs_amd/libcqs_hip.so $REPO/build/variants/lib_att_*.so; do
    n=$(basename $f .so)
    export CQS_HIP_LIB=$f
    timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/abl_$n -o kt --output-format csv -- python3 $REPO/tools/embed_bench.py --iters 3 > $OUT/abl_$n.txt 2> $OUT/abl_$n.err
    echo "== $n: $(python3 $REPO/tools/summarize_prof.py $OUT/abl_$n | grep -E 'attention' | head -2)"
  done
fi
