#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel_stats / counter_collection) into a short text summary
with kernel names truncated, for committing under profiles/.

  summarize_prof.py DIR [DIR ...]                       text summary on stdout
  summarize_prof.py --scan-traffic FETCH_DIR WRITE_DIR ALG_BYTES OUT.json [COMMIT]
        per-launch HBM bytes of scan_gemv_kernel from the two PMC passes, corrected as MI355X_MICROARCH.md prescribes
        (FETCH_SIZE is in KiB and counts wide reads at half size on gfx950: x 1024 x 2; WRITE_SIZE: x 1024)"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    m = re.match(r"_ZN3cqs(?:12_GLOBAL__N_1)?(\d+)([A-Za-z_0-9]+)", name)
    if m:                                  # mangled: cqs::<len><name>I...E
        n = int(m.group(1))
        base = m.group(2)[:n]
        targs = re.search(r"I((?:L[ijb]\d+E)+)E", name)
        t = ""
        if targs:
            t = "<" + ",".join(re.findall(r"L[ijb](\d+)E", targs.group(1))) + ">"
        return ("cqs::" + base + t)[:90]
    cut = name.find("(")
    if cut > 0:
        name = name[:cut]
    return name[:90]


def kernel_stats(d):
    for f in glob.glob(d + "/**/*_kernel_stats.csv", recursive=True):
        print(f"# kernel stats ({f.split('/')[-1]})")
        print(f"{'kernel':72s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
        for r in csv.DictReader(open(f)):
            print(f"{short(r['Name']):72s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:10.2f} "
                  f"{float(r['MinNs'])/1e3:10.2f} {float(r['MaxNs'])/1e3:10.2f} {float(r['Percentage']):6.2f}")


def counter_means(d):
    acc = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            key = (short(r["Kernel_Name"]), r["Counter_Name"])
            acc[key][0] += 1
            acc[key][1] += float(r["Counter_Value"])
    return acc


def counters(d):
    acc = counter_means(d)
    if acc:
        print(f"# counters ({d.rstrip('/').split('/')[-1]}): per-dispatch mean")
    for (k, c), (cnt, tot) in sorted(acc.items()):
        if k.startswith("cqs::") or "rocclr" in k:
            print(f"{k:72s} {c:28s} dispatches={cnt:5d} mean={tot/cnt:16.3f}")


def scan_traffic(fetch_dir, write_dir, alg_bytes, out, commit=None):
    import hashlib
    import os
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        sha = hashlib.sha256(open(os.path.join(here, "cqs_amd", "csrc", "scan_kernels.hip"), "rb").read()).hexdigest()[:16]
    except OSError:
        sha = None
    fm, wm = counter_means(fetch_dir), counter_means(write_dir)
    f = [(k, v) for k, v in fm.items() if k[1] == "FETCH_SIZE" and "scan_gemv_kernel" in k[0]]
    w = [(k, v) for k, v in wm.items() if k[1] == "WRITE_SIZE" and "scan_gemv_kernel" in k[0]]
    if not f or not w:
        raise SystemExit("no scan_gemv_kernel counters found")
    fk, (fc, ft) = max(f, key=lambda kv: kv[1][0])
    wk, (wc, wt) = max(w, key=lambda kv: kv[1][0])
    rd = ft / fc * 1024.0 * 2.0
    wr = wt / wc * 1024.0
    json.dump({"kernel": fk[0], "alg_bytes_per_launch": int(alg_bytes), "hbm_bytes_per_launch": round(rd + wr),
               "read_bytes": round(rd), "write_bytes": round(wr), "dispatches": fc,
               "commit": commit, "scan_kernels_sha256": sha,     # what the counters were measured on (bench.py quotes both)
               "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `bench.py --steps 50`, "
                         "FETCH_SIZE KiB x 1024 x 2 (gfx950 wide-read correction), WRITE_SIZE KiB x 1024; "
                         "see profiles/ for the summary of the same run"}, open(out, "w"), indent=1)
    print(f"{fk[0]}: read {rd:.4e} B + write {wr:.4e} B per launch vs algorithmic {float(alg_bytes):.4e} B "
          f"(x{(rd + wr) / float(alg_bytes):.4f})")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--scan-traffic":
        scan_traffic(*sys.argv[2:7])
    else:
        for d in sys.argv[1:]:
            kernel_stats(d)
            counters(d)
