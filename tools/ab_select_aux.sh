#!/bin/bash
# A/B of the select's (argmax, runner-up) index on ONE box: scan + select per step at k = 20 / 500, with and without it,
# interleaved three times (tools/time_scan.py: device API, HIP events over 200 steps).
for rep in 1 2 3; do
  for k in 20 500; do
    echo -n "aux=1 k=$k "; python3 tools/time_scan.py 1000000 1 $k | awk '{print $NF}'
    echo -n "aux=0 k=$k "; CQS_HIP_SELECT_AUX=0 python3 tools/time_scan.py 1000000 1 $k | awk '{print $NF}'
  done
done
