#!/bin/bash
# batched scan timing (1M x 768) for query blocks 16..128 across library variants
cd "${GRAFT_REPO_ROOT:?}" || exit 1
for lib in "" "$@"; do
  for b in 16 32 48 64 128; do
    if [ -n "$lib" ]; then export CQS_HIP_LIB="/root/repo/build/variants/$lib"; else unset CQS_HIP_LIB; fi
    timeout -k 10 120 python tools/time_scan.py 1000000 $b || exit 1
  done
done
