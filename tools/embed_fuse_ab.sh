#!/bin/bash
# (A/B helper from round 3; run on the GPU box from the repo root)
for i in 1 2; do
for f in 1 0; do
CQS_HIP_GEMM_FUSE_NORM=$f python bench.py --steps 5 --warmup 2 --extras 0 --e2e-chunks 0 --cpu-seconds 0 --embed-steps 16 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());e=d['embed'];k=[x for x in e if x.startswith('fixed_len_512') and 'batch' not in x][0]
print('fuse=$f', e[k]['chunks_per_sec'], e[k]['sync_api']['chunks_per_sec'], e['lognormal_len']['chunks_per_sec'], [v for kk,v in e.items() if 'batch128' in kk][0]['chunks_per_sec'])"
done; done
