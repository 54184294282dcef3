#!/bin/bash
# A/B of libcqs_hip.so builds on the sparse bench under the kernel trace: usage r04_sparse_ab.sh "ARGS" LIB...
REPO="${GRAFT_REPO_ROOT:?}"
ARGS="$1"; shift
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  ( [ "$lib" != "-" ] && export CQS_HIP_LIB="$lib"
    rm -rf /tmp/spab; timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/spab -o kt --output-format csv -- python3 "$REPO/tools/sparse_bench.py" $ARGS > /tmp/spab.log 2> /tmp/spab.err || { tail -3 /tmp/spab.err; exit 1; }
    echo "== $lib $ARGS"; python3 "$REPO/tools/summarize_prof.py" /tmp/spab | grep "sparse_acc" ) || exit 1
done
