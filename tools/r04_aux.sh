#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
true
python - <<'PY'
import sys, argparse
sys.path.insert(0, '.')
import numpy as np, bench
a = argparse.Namespace(cpu_seconds=0.0, embed_steps=8)
import json
out = bench.aux_models_leg(a, np)
for k, v in out.items():
    if isinstance(v, dict):
        print(k, {kk: vv for kk, vv in v.items() if kk in ('docs_per_sec','pairs_per_sec','chunks_per_sec','tflops','roofline','ms_per_batch','query_ms')})
PY
