#!/usr/bin/env python3
"""bench.py's concurrent_clients leg alone (1M x 768 corpus, native threads on one handle, every answer compared bit for
bit with the lone call's) - for A/B runs of the combining queue / the gemv passes (e.g. CQS_HIP_SCAN_GEMV16=1)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cqs_amd import HipIndex
from bench_legs.common import make_unit_rows
from bench_legs.clients import concurrent_clients_leg

dev = torch.device("cuda", 0)
n, dim, k = int(os.environ.get("ROWS", 1_000_000)), 768, 20
rows = make_unit_rows(torch, n, dim, 0xC950001, dev)
idx = HipIndex.build_from_device(None, rows.data_ptr(), n, dim, borrow=True, keepalive=rows)
q = make_unit_rows(torch, 96, dim, 0xC950031, dev).cpu().numpy()
out = concurrent_clients_leg(np, idx, q, k, dim, rows_ptr=rows.data_ptr())
print(json.dumps({t: (v["queries_per_sec"], v["ms_per_call"]) for t, v in out["native_threads"].items()}))
print("relaxed", json.dumps({t: (v["queries_per_sec"], v["ms_per_call"]) for t, v in out.get("native_threads_relaxed_bits", {}).items() if t != "what"}))
