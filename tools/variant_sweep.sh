#!/bin/bash
# A/B of libcqs_hip.so kernel variants (build/variants/lib_*.so) with bench.py on one GPU.
# usage: tools/variant_sweep.sh [bench args...]
for f in cqs_amd/libcqs_hip.so build/variants/lib_*.so; do
  CQS_HIP_LIB=$PWD/$f timeout -k 10 120 python bench.py --steps 300 --warmup 30 --cpu-seconds 0 "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-36s q/s=%8.1f step_ms=%.4f scan_ms=%.4f GB/s=%7.1f' % ('$f'.split('/')[-1], d['value'], d['ms_per_step'], r['avg_launch_ms'], r['achieved']))"
done
