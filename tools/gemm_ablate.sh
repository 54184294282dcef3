#!/bin/bash
# Same-box timing of kernel-ablation builds (build/variants/lib_p8_*.so): shares of the GEMM's time.
for f in cqs_amd/libcqs_hip.so build/variants/lib_p8_*.so; do
  echo "== $f"
  CQS_HIP_LIB=$PWD/$f timeout -k 10 120 python tools/gemm_bench.py "$@" 2>/dev/null | grep -E "M= 16384|M=  4096 N= 4096" | tail -5
done
