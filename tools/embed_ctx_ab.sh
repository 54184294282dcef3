#!/bin/bash
# A/B on one box: embedding tickets over two execution contexts (default) vs one (CQS_HIP_EMBED_CONTEXTS=1)
for i in 1 2 3; do
for c in 2 1; do
CQS_HIP_EMBED_CONTEXTS=$c python bench.py --steps 5 --warmup 2 --extras 0 --e2e-chunks 0 --cpu-seconds 0 --embed-steps 16 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());e=d['embed'];k=[x for x in e if x.startswith('fixed_len_512') and 'batch' not in x][0]
print('contexts=$c', 'tickets', e[k]['chunks_per_sec'], 'sync', e[k]['sync_api']['chunks_per_sec'], 'lognormal', e['lognormal_len']['chunks_per_sec'], 'batch128', [v for kk,v in e.items() if 'batch128' in kk][0]['chunks_per_sec'])"
done; done
