#!/bin/bash
# embed leg only (tiny scan corpus), A/B over environment settings given as arguments: "NAME=VAL,NAME=VAL" per arm ("-" = defaults)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out
for arm in "$@"; do
  envs=()
  if [ "$arm" != "-" ]; then IFS=',' read -ra envs <<< "$arm"; fi
  for rep in 1 2; do
    env "${envs[@]}" timeout -k 10 300 python bench.py --rows 20000 --steps 5 --warmup 2 --extras 0 --e2e-chunks 0 --cpu-seconds 0 --embed-steps 24 --abi-devices "" 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=l['embed']
f=e['fixed_len_512']; r=e['lognormal_len']; b=e['fixed_len_512_batch128']
print('$arm rep$rep fixed tickets %.0f sync %.0f dev_ms %.3f | lognormal tickets %.0f sync %.0f | b128 tickets %.0f sync %.0f' % (f['chunks_per_sec'], f['sync_api']['chunks_per_sec'], f['sync_api']['device_ms_per_batch'], r['chunks_per_sec'], r['sync_api']['chunks_per_sec'], b['chunks_per_sec'], b['sync_api']['chunks_per_sec']))
" || exit 1
  done
done
