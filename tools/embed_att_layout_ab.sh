#!/bin/bash
# (A/B helper from round 3; run on the GPU box from the repo root)
for i in 1 2; do
for L in default shared; do
if [ $L = shared ]; then export CQS_HIP_ATT_LAYOUT=shared; else unset CQS_HIP_ATT_LAYOUT; fi
python bench.py --steps 5 --warmup 2 --extras 0 --e2e-chunks 0 --cpu-seconds 0 --embed-steps 16 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());e=d['embed']
print('layout=$L', 'lognormal tickets', e['lognormal_len']['chunks_per_sec'], 'sync', e['lognormal_len']['sync_api']['chunks_per_sec'], 'dev ms', e['lognormal_len']['sync_api']['device_ms_per_batch'], 'tokens', e['lognormal_len']['tokens_per_batch'])"
done; done
