#!/usr/bin/env python3
"""Experiment (round 5): can TWO 8-query gemv passes over the SAME corpus, launched together from two
handles (own scratch + stream each, rows borrowed from one allocation), finish sooner than two passes back to back?
The second kernel's task t reads rows the first one's task t has just read (L2 / Infinity Cache), so 16 callers might ride
one HBM sweep while every caller still gets its lone-call bits (gemv passes only).  Prints ms per 16 queries both ways."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cqs_amd import HipIndex
from bench_legs.common import make_unit_rows

dev = torch.device("cuda", 0)
n, dim, k = int(os.environ.get("ROWS", 1_000_000)), 768, 20
rows = make_unit_rows(torch, n, dim, 1, dev)
q = make_unit_rows(torch, 16, dim, 2, dev)
a = HipIndex.build_from_device(None, rows.data_ptr(), n, dim, borrow=True, keepalive=rows)
b = HipIndex.build_from_device(None, rows.data_ptr(), n, dim, borrow=True, keepalive=rows)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
ka = torch.zeros((8, k), dtype=torch.int64, device=dev); ca = torch.zeros(8, dtype=torch.int32, device=dev)
kb = torch.zeros((8, k), dtype=torch.int64, device=dev); cb = torch.zeros(8, dtype=torch.int32, device=dev)
qa, qb = q[:8].contiguous(), q[8:].contiguous()


def both(concurrent, reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        a.search_device(qa.data_ptr(), 8, k, ka.data_ptr(), ca.data_ptr(), stream=s1.cuda_stream)
        if not concurrent:
            s1.synchronize()
        b.search_device(qb.data_ptr(), 8, k, kb.data_ptr(), cb.data_ptr(), stream=s2.cuda_stream)
        s1.synchronize(); s2.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for mode in (False, True, False, True):
    both(mode, 5)
    print("concurrent" if mode else "back to back", "%.4f ms per 16 queries" % both(mode, 40), flush=True)
ref = HipIndex.build_from_device(None, rows.data_ptr(), n, dim, borrow=True, keepalive=rows)
kr = torch.zeros((8, k), dtype=torch.int64, device=dev); cr = torch.zeros(8, dtype=torch.int32, device=dev)
ref.search_device(qb.data_ptr(), 8, k, kr.data_ptr(), cr.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print("bit-identical to a lone pass:", bool(torch.equal(kr, kb)))
