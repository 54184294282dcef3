#!/bin/bash
# One gpurun call: GPU tests, the default bench line, then rocprofv3 kernel-trace and the two PMC passes
# of the same bench command.  Usage: tools/profile_round.sh <tag>   (outputs under gpurun_out/<tag>_*)
set -e
TAG=${1:-r01d}
OUT=$PWD/gpurun_out
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_pytest.log 2>&1
tail -3 $OUT/${TAG}_pytest.log
timeout -k 10 600 python bench.py > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench.err
cat $OUT/${TAG}_bench_line.json
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --cpu-seconds 0 --embed-steps 3 --extras 0"
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_kt -o kt --output-format csv -- $BENCH > $OUT/${TAG}_bench_under_profiler.json 2> $OUT/${TAG}_kt.err
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE -d $OUT/${TAG}_fetch -o pmc --output-format csv -- python3 $REPO/bench.py --cpu-seconds 0 --embed-steps 0 --extras 0 --steps 50 > /dev/null 2> $OUT/${TAG}_fetch.err
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE -d $OUT/${TAG}_write -o pmc --output-format csv -- python3 $REPO/bench.py --cpu-seconds 0 --embed-steps 0 --extras 0 --steps 50 > /dev/null 2> $OUT/${TAG}_write.err
cd $REPO
python3 tools/summarize_prof.py $OUT/${TAG}_kt $OUT/${TAG}_fetch $OUT/${TAG}_write > $OUT/${TAG}_summary.txt
echo done
