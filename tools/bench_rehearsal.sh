#!/bin/bash
# Logic rehearsal of bench.py's N>1 paths on ONE GPU (never a performance number): 2 gloo ranks on cuda:0 in the
# strong (configs[4]) and weak modes, a 1-rank RCCL group, and the single-process sharded ABI over devices 0,0.
set -e
OUT=${1:-gpurun_out/rehearsal}
mkdir -p $OUT
COMMON="--steps 6 --warmup 2 --embed-steps 0 --cpu-seconds 0 --extras 0 --e2e-chunks 0"
CQS_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 \
  bench.py --gpus 2 --total-rows 400000 $COMMON > $OUT/strong2.json 2> $OUT/strong2.err
CQS_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 \
  bench.py --gpus 2 --mode weak --rows 200000 $COMMON > $OUT/weak2.json 2> $OUT/weak2.err
CQS_BENCH_FORCE_DIST=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29613 \
  bench.py --gpus 1 --total-rows 400000 $COMMON > $OUT/strong1_rccl.json 2> $OUT/strong1_rccl.err
timeout -k 10 300 python bench.py --rows 400000 --abi-devices 0,0,0,0 $COMMON > $OUT/abi.json 2> $OUT/abi.err
for f in strong2 weak2 strong1_rccl abi; do python - "$OUT/$f.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], d["config"]["mode"], "n_gpus", d["n_gpus"], "value", d["value"], d["scaling"], d.get("abi_sharded"))
PY
done
