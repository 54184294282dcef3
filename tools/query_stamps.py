#!/usr/bin/env python3
"""Where the search-time chain spends its time: per kernel slot (5 per layer + 2), from the in-kernel stamps
(CQS_HIP_QUERY_STAMPS=1): span of the kernel over its workgroups, phase times of the median workgroup, the gap to the
previous kernel's last stamp, and the shader clock.  usage: query_stamps.py <tokens> [eager|graph]"""
import ctypes as C, os, sys
os.environ["CQS_HIP_QUERY_STAMPS"] = "1"
if len(sys.argv) > 2 and sys.argv[2] in ("eager", "repeat"):
    os.environ["CQS_HIP_QUERY_GRAPH"] = "0"
REP = 2 if len(sys.argv) > 2 and sys.argv[2] == "repeat" else 1
if REP == 2:
    os.environ["CQS_HIP_QUERY_DEBUG_REPEAT"] = "2"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tools.embed_two_streams_lib import make_engine
from cqs_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
e, cfg = make_engine(0)
rng = np.random.default_rng(3)
ids = rng.integers(1, 262144, size=(1, n)).astype(np.int64); mask = np.ones((1, n), np.int64)
for _ in range(9): e.run(ids, mask)          # odd count: the last query ran on context 1 ... read both, keep the later
lib = _lib.load()
f = lib.cqs_hip_debug_query_stamps
f.restype = C.c_uint64
f.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64]
slots = (cfg.layers * 5 + 2) * REP
best = None
for ctx in (0, 1):
    buf = np.zeros(slots * 256 * 8, np.uint64)
    got = f(e._h, ctx, buf.ctypes.data_as(C.c_void_p), buf.size)
    if got and (best is None or buf.max() > best.max()):
        best = buf
d = best.reshape(slots, 256, 8).astype(np.int64)
names = ["qkv", "attn", "oproj", "geglu", "down"]
prev_end = None
tot = {}
print("slot kernel  wgs  span_us | median wg: start->operands  ->reduced  ->end | gap_from_prev_us  clock_GHz")
for s in range(slots):
    live = d[s, :, 0] > 0
    if not live.any():
        continue
    rt = d[s, live, :4].astype(np.float64) * 0.01          # us (100 MHz)
    ex = d[s, live, 4:8].astype(np.float64) * 0.01
    ends = np.where(rt[:, 3] > 0, rt[:, 3], rt[:, 2])
    start, end = rt[:, 0].min(), ends.max()
    med = np.median(rt, axis=0)
    ks = s // REP
    name = names[ks % 5] if ks < cfg.layers * 5 else ("dense1", "dense2")[ks - cfg.layers * 5]
    ghz = 0.0
    if (ex > 0).all() and s < 12 * REP:
        m4 = np.median(ex - rt[:, :1], axis=0)
        print("      %s extra stamps (us after start): %.2f  %.2f  %.2f  %.2f" % ((name,) + tuple(m4)))
    ks = s // REP
    name = names[ks % 5] if ks < cfg.layers * 5 else ("dense1", "dense2")[ks - cfg.layers * 5]
    if REP == 2:
        name += ".%d" % (s % 2)
    gap = start - prev_end if prev_end is not None else 0.0
    prev_end = end
    key = name
    tot.setdefault(key, []).append((end - start, gap, med[1] - med[0], med[2] - med[1], (med[3] if med[3] > 0 else med[2]) - med[2]))
    if s < 10 * REP or s >= slots - 2 * REP:
        print(f"{s:4d} {name:6s} {int(live.sum()):4d} {end - start:8.2f} | {med[1] - med[0]:10.2f} {med[2] - med[1]:10.2f} {(med[3] if med[3] > 0 else med[2]) - med[2]:8.2f} | {gap:8.2f} {ghz:8.2f}")
print("\nper kernel kind, mean over layers: span, gap before, start->operands, ->reduced, ->end (us)")
allspan = allgap = 0.0
for k, v in tot.items():
    a = np.array(v)
    print(f"{k:7s} x{len(v):3d}  span {a[:,0].mean():6.2f}  gap {a[:,1].mean():6.2f}  ph1 {a[:,2].mean():6.2f}  ph2 {a[:,3].mean():6.2f}  ph3 {a[:,4].mean():6.2f}")
    allspan += a[:, 0].sum(); allgap += a[:, 1].sum()
print(f"chain: kernels {allspan:.1f} us + gaps {allgap:.1f} us = {allspan + allgap:.1f} us for {n} tokens")
