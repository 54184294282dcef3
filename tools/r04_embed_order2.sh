#!/bin/bash
# does the sparse leg in front of the embed leg cost the ticketed measurement anything? (same box, back to back)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
run() {
  timeout -k 10 500 python bench.py --e2e-chunks 0 --abi-devices "" --cpu-seconds 0 "$@" 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=l['embed']; f=e['fixed_len_512']; b=e['fixed_len_512_batch128']
print('$*', '| fixed tickets %.0f sync %.0f dev_ms %.3f | b128 tickets %.0f sync %.0f' % (f['chunks_per_sec'], f['sync_api']['chunks_per_sec'], f['sync_api']['device_ms_per_batch'], b['chunks_per_sec'], b['sync_api']['chunks_per_sec']))"
}
run --sparse-chunks 0 || exit 1
run || exit 1
run --sparse-chunks 0 || exit 1
run || exit 1
