#!/bin/bash
# kernel trace (timestamps) of the sharded handle's blocking search in its one-GPU form; prints the last queries' timelines
REPO="${GRAFT_REPO_ROOT:?}"
OUT="$REPO/gpurun_out"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 "$REPO/tools/r04_shard_trace.py" ${1:-0,0,0,0} > "$OUT/shtr_plain.log" 2>&1 || { tail -5 "$OUT/shtr_plain.log"; exit 1; }
cat "$OUT/shtr_plain.log"
timeout -k 10 300 rocprofv3 --kernel-trace -d "$OUT/shtr" -o kt --output-format csv -- python3 "$REPO/tools/r04_shard_trace.py" ${1:-0,0,0,0} > "$OUT/shtr.log" 2> "$OUT/shtr.err" || { tail -5 "$OUT/shtr.err"; exit 1; }
cat "$OUT/shtr.log"
python3 - "$OUT/shtr" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/kt_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows))
# the sharded queries: 50 x (4 scans); print the timeline of queries 45..47 (groups split at gaps > 100 us after a select)
scans = [i for i, e in enumerate(ev) if "scan_gemv" in e[2]]
# sharded phase = first 200 scan kernels (50 queries x 4 shards)
first = scans[4 * 45]; last = scans[4 * 48 - 1]
t0 = ev[first][0]
for e in ev[first - 6:last + 6]:
    print("%9.1f us  +%7.1f  %-60s q=%s" % ((e[0] - t0) / 1e3, (e[1] - e[0]) / 1e3, e[2], e[3]))
PY
rm -rf "$OUT/shtr"
