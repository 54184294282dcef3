#!/bin/bash
# sparse index search (tools/sparse_bench.py) under rocprofv3: kernel trace, then FETCH_SIZE and WRITE_SIZE in passes of
# their own (no tracing domain beside them); one summary -> gpurun_out/sparse_prof.txt.  Arguments go to sparse_bench.py.
REPO="${GRAFT_REPO_ROOT:?}"
OUT="$REPO/gpurun_out"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 "$REPO/tools/sparse_bench.py" "$@" > "$OUT/sparse_plain.log" 2>&1 || { tail -5 "$OUT/sparse_plain.log"; exit 1; }
grep -v amdgpu.ids "$OUT/sparse_plain.log"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/sparse_prof" -o kt --output-format csv -- python3 "$REPO/tools/sparse_bench.py" "$@" > "$OUT/sparse_prof.log" 2> "$OUT/sparse_prof.err" || { tail -5 "$OUT/sparse_prof.err"; exit 1; }
if [ -n "$SPARSE_PMC" ]; then
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d "$OUT/sparse_fetch" -o pmc --output-format csv -- python3 "$REPO/tools/sparse_bench.py" "$@" > /dev/null 2> "$OUT/sparse_fetch.err" || { tail -5 "$OUT/sparse_fetch.err"; exit 1; }
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d "$OUT/sparse_write" -o pmc --output-format csv -- python3 "$REPO/tools/sparse_bench.py" "$@" > /dev/null 2> "$OUT/sparse_write.err" || { tail -5 "$OUT/sparse_write.err"; exit 1; }
  { echo "# tools/sparse_bench.py $*"; cat "$OUT/sparse_plain.log" | grep -v amdgpu.ids; python3 "$REPO/tools/summarize_prof.py" "$OUT/sparse_prof" "$OUT/sparse_fetch" "$OUT/sparse_write"; } > "$OUT/sparse_prof.txt"
  rm -rf "$OUT/sparse_fetch" "$OUT/sparse_write"
else
  { echo "# tools/sparse_bench.py $*"; cat "$OUT/sparse_plain.log" | grep -v amdgpu.ids; python3 "$REPO/tools/summarize_prof.py" "$OUT/sparse_prof"; } > "$OUT/sparse_prof.txt"
fi
head -30 "$OUT/sparse_prof.txt"
rm -rf "$OUT/sparse_prof"
