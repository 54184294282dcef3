#!/bin/bash
# kernel trace of the embedding forward (tools/embed_bench.py, one blocking call at a time) per environment arm.
# usage: r04_prof.sh TAG "ENV=VAL,ENV=VAL" ["ENV..." ...]   ("-" = defaults); summaries -> gpurun_out/TAG_<n>.txt
REPO="${GRAFT_REPO_ROOT:?}"
OUT="$REPO/gpurun_out"
mkdir -p "$OUT"
TAG="$1"; shift
cd /tmp && export TMPDIR=/tmp
n=0
for arm in "$@"; do
  n=$((n + 1))
  ( if [ "$arm" != "-" ]; then IFS=',' read -ra kv <<< "$arm"; for e in "${kv[@]}"; do export "$e"; done; fi
    timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/${TAG}_$n" -o kt --output-format csv -- python3 "$REPO/tools/embed_bench.py" --iters 6 --vocab 32768 ${EMBED_BENCH_ARGS} > "$OUT/${TAG}_$n.log" 2> "$OUT/${TAG}_$n.err" ) || { tail -5 "$OUT/${TAG}_$n.err"; exit 1; }
  { echo "# arm: $arm"; cat "$OUT/${TAG}_$n.log"; python3 "$REPO/tools/summarize_prof.py" "$OUT/${TAG}_$n"; } > "$OUT/${TAG}_$n.txt"
  rm -rf "$OUT/${TAG}_$n"
  echo "== $arm"; cat "$OUT/${TAG}_$n.log"; grep -E "cqs::" "$OUT/${TAG}_$n.txt" | sort -k5 -n -r | head -9
done
