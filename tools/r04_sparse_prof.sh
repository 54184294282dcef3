#!/bin/bash
# kernel trace of the sparse index search (tools/sparse_bench.py): per-kernel durations
REPO="${GRAFT_REPO_ROOT:?}"
OUT="$REPO/gpurun_out"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 "$REPO/tools/sparse_bench.py" "$@" > "$OUT/sparse_plain.log" 2>&1 || { tail -5 "$OUT/sparse_plain.log"; exit 1; }
grep -v amdgpu.ids "$OUT/sparse_plain.log"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/sparse_prof" -o kt --output-format csv -- python3 "$REPO/tools/sparse_bench.py" "$@" > "$OUT/sparse_prof.log" 2> "$OUT/sparse_prof.err" || { tail -5 "$OUT/sparse_prof.err"; exit 1; }
python3 "$REPO/tools/summarize_prof.py" "$OUT/sparse_prof" > "$OUT/sparse_prof.txt"
head -12 "$OUT/sparse_prof.txt"
rm -rf "$OUT/sparse_prof"
