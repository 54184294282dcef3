#!/usr/bin/env python3
"""bench.py's aux-models leg alone (SPLADE, bge-large / e5-base embedders, reranker): throughput lines only."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--embed-steps", type=int, default=8)
a = ap.parse_args()
from bench_legs.aux_models import aux_models_leg
out = aux_models_leg(a, np)
for k, v in out.items():
    if isinstance(v, dict):
        print(k, json.dumps({kk: vv for kk, vv in v.items() if "per_sec" in kk or kk in ("ms_per_batch", "sync_api", "tflops")}))
