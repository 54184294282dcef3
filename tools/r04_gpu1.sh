#!/bin/bash
# round 4, first GPU pass: the whole -m gpu suite, then the default bench line
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04a_tests.log 2>&1
rc=$?
tail -5 gpurun_out/r04a_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py > gpurun_out/r04a_bench.json 2> gpurun_out/r04a_bench.err
rc=$?
tail -c 600 gpurun_out/r04a_bench.err
python - <<'PY'
import json
l = json.loads(open("gpurun_out/r04a_bench.json").read().strip().splitlines()[-1])
print("value", l["value"], "ms", l["ms_per_step"], "roofline", l["roofline"]["frac"])
print("latency_host_api", l["latency_host_api"])
print("clients", json.dumps(l["concurrent_clients"]))
print("abi_sharded", json.dumps(l["abi_sharded"]))
print("cpu", json.dumps(l["cpu_baseline"]))
e = l["embed"]
print("embed fixed", e["fixed_len_512"]["chunks_per_sec"], e["fixed_len_512"]["roofline"]["frac"], "sync", e["fixed_len_512"]["sync_api"])
print("lognormal", e["lognormal_len"]["chunks_per_sec"], e["lognormal_len"]["roofline"]["frac"])
print("b128", e["fixed_len_512_batch128"]["chunks_per_sec"], e["fixed_len_512_batch128"]["sync_api"])
print("qlat", json.dumps(e["query_latency"]["by_tokens"]))
print("first", json.dumps(e["query_latency"].get("first_call_ms")))
print("rand", json.dumps(e["query_latency"].get("random_lengths")))
PY
exit $rc
