#!/bin/bash
# kernel trace of the search-time chain at the given lengths (eager launches: graph replays hide kernel names from the stats)
REPO="${GRAFT_REPO_ROOT:?}"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for T in "$@"; do
  CQS_HIP_QUERY_GRAPH=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/qprof_$T" -o kt --output-format csv -- python3 "$REPO/tools/r04_qprof.py" $T > "$OUT/qprof_$T.log" 2> "$OUT/qprof_$T.err" || { tail -5 "$OUT/qprof_$T.err"; exit 1; }
  { cat "$OUT/qprof_$T.log"; python3 "$REPO/tools/summarize_prof.py" "$OUT/qprof_$T"; } > "$OUT/qprof_$T.txt"
  rm -rf "$OUT/qprof_$T"
  echo "== T=$T"; cat "$OUT/qprof_$T.log"; grep -E "cqs::qf" "$OUT/qprof_$T.txt" | head -12
done
