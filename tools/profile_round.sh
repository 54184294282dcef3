#!/bin/bash
# usage: tools/profile_round.sh <tag> [commit]   (commit = what the counters are measured on; recorded in <tag>_scan_traffic.json)
# One gpurun call: rocprofv3 kernel trace of the default bench command + the two PMC passes of the scan (separate
# passes, no tracing domains beside --kernel-trace: MI355X_MICROARCH.md / gpurun rules), the embed forward's kernel
# trace + PMC, and the summaries.  Usage: tools/profile_round.sh <tag>   (outputs under gpurun_out/<tag>_*)
TAG=${1:-r02}
COMMIT=${2:-unknown}
OUT=$PWD/gpurun_out
REPO=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 900 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_kt -o kt --output-format csv -- python3 $REPO/bench.py --cpu-seconds 0 --embed-steps 3 --extras 0 --e2e-chunks 0 --abi-devices '' > $OUT/${TAG}_bench_under_profiler.json 2> $OUT/${TAG}_kt.err
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE -d $OUT/${TAG}_fetch -o pmc --output-format csv -- python3 $REPO/bench.py --cpu-seconds 0 --embed-steps 0 --extras 0 --e2e-chunks 0 --abi-devices '' --steps 50 > /dev/null 2> $OUT/${TAG}_fetch.err
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE -d $OUT/${TAG}_write -o pmc --output-format csv -- python3 $REPO/bench.py --cpu-seconds 0 --embed-steps 0 --extras 0 --e2e-chunks 0 --abi-devices '' --steps 50 > /dev/null 2> $OUT/${TAG}_write.err
cd $REPO
# (--abi-devices '': the sharded leg launches the same scan instantiation on 250k-row shards and would dilute the per-launch means)
python3 tools/summarize_prof.py $OUT/${TAG}_kt $OUT/${TAG}_fetch $OUT/${TAG}_write > $OUT/${TAG}_scan_summary.txt
python3 tools/summarize_prof.py --scan-traffic $OUT/${TAG}_fetch $OUT/${TAG}_write 3072000000 $OUT/${TAG}_scan_traffic.json $COMMIT
bash tools/profile_embed.sh ${TAG} > /dev/null 2>&1
python3 tools/summarize_prof.py $OUT/${TAG}_ekt $OUT/${TAG}_efetch $OUT/${TAG}_ewrite $OUT/${TAG}_esq > $OUT/${TAG}_embed_summary.txt
head -12 $OUT/${TAG}_scan_summary.txt; cat $OUT/${TAG}_scan_traffic.json; head -16 $OUT/${TAG}_embed_summary.txt
rm -rf $OUT/${TAG}_kt $OUT/${TAG}_fetch $OUT/${TAG}_write $OUT/${TAG}_ekt $OUT/${TAG}_efetch $OUT/${TAG}_ewrite $OUT/${TAG}_esq
echo done
