"""bench.py's N>1 code paths rehearsed on ONE GPU (logic only, never a performance number): 2 gloo ranks sharing
cuda:0 in the strong mode (BASELINE configs[4]: one corpus cut over the ranks, one all-gather of per-shard top-k,
host merge - the bench asserts the merged answer against a direct fp64 dot and an exhaustive cross-shard count),
the weak mode, and a 1-rank RCCL group that runs the real collective calls."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--steps", "6", "--warmup", "2", "--embed-steps", "0", "--cpu-seconds", "0", "--extras", "0", "--e2e-chunks", "0"]


def _run(nproc, port, extra, env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc)] + extra + COMMON
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_strong_mode_two_ranks(hip):
    d = _run(2, 29621, ["--total-rows", "300000"], {"CQS_BENCH_REHEARSAL": "1"})
    assert d["scaling"] == "strong" and d["n_gpus"] == 2 and d["config"]["mode"] == "strong"
    assert d["config"]["total_rows"] == 300000 and d["config"]["rows_per_gpu"] == 150000
    assert "configs[4]" in d["config"]["workload"] and d["value"] > 0
    # a rehearsal line can not be mistaken for a multi-GPU measurement
    assert d["rehearsal"] is True and d["metric"].startswith("REHEARSAL") and "gloo" in d["config"]["collective"]
    assert d["config"]["ranks_seen"] == 2 and [r["device"] for r in d["config"]["rank_devices"]] == [0, 0]
    assert d["prewarm_steps"] == 20


def test_weak_mode_two_ranks(hip):
    d = _run(2, 29622, ["--mode", "weak", "--rows", "150000"], {"CQS_BENCH_REHEARSAL": "1"})
    assert d["scaling"] == "weak" and d["config"]["total_rows"] == 300000 and d["config"]["queries_per_step"] == 2


def test_strong_mode_one_rank_rccl(hip):
    d = _run(1, 29623, ["--total-rows", "300000"], {"CQS_BENCH_FORCE_DIST": "1"})
    assert d["scaling"] == "strong" and d["n_gpus"] == 1 and d["roofline"]["bound"] == "hbm"
    # the self-proving keys of the first real N > 1 run, rehearsed with the 1-rank RCCL group: group size as an
    # all-reduce over the backend saw it, the device behind every rank, and the C ABI's sharded handle run by rank 0
    # after the group is gone (devices 0..N-1 = [0] here; RCCL calls inside the library with one rank)
    assert d["config"]["ranks_seen"] == 1 and d["config"]["collective"].startswith("rccl") and d["rehearsal"] is True
    assert d["config"]["rank_devices"][0]["device"] == 0 and ":" in d["config"]["rank_devices"][0]["pci"]
    ab = d["abi_sharded"]
    assert ab and "error" not in ab, ab
    assert ab["devices"] == [0] and ab["checked_vs_single_device"] and ab["rows"] == 250000 and ab["queries_per_sec_host_api"] > 0
    # ... and the N = 1 point of the line's own strong-scaling curve: the same corpus size on one GPU, measured by rank 0
    # in the same run (the driver's N = 1 run is the headline 1M-row workload, not comparable with a 10M-row line)
    s1 = d["strong_scaling_n1"]
    assert s1 and "error" not in s1, s1
    assert s1["n_gpus"] == 1 and s1["rows"] == 300000 and s1["value"] > 0 and s1["unit"] == "queries/s"


def test_single_gpu_line_carries_the_same_keys(hip):
    env = dict(os.environ)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rows", "200000", "--steps", "5", "--warmup", "2"] + COMMON[4:],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert d["config"]["ranks_seen"] == 1 and d["prewarm_steps"] == 200 and "rehearsal" not in d and d["n_gpus"] == 1


DRIVER_CMD = ["--gpus", "1", "--steps", "20", "--warmup", "5"]       # what the driver runs at round end, verbatim
TOP_LEVEL = ["roofline", "cpu_baseline", "latency_host_api", "concurrent_clients", "other_configs", "abi_sharded",
             "strong_scaling_n1", "embed", "e2e", "aux_models", "sparse_index"]


def _errors_in(x, path=""):
    """Every {"error": ...} / {"skipped": ...} anywhere under a leg's key."""
    out = []
    if isinstance(x, dict):
        for key in ("error", "skipped"):
            if key in x:
                out.append((path, key, x[key]))
        for k, v in x.items():
            out += _errors_in(v, path + "/" + str(k))
    return out


def test_the_drivers_command_with_every_leg_on(hip):
    """VERDICT r04 #1c: `bench.py --gpus 1 --steps 20 --warmup 5` with extras ON (round 4's line died in a leg that no test
    ran at the driver's --steps).  Sizes are scaled by flags only; every top-level key must be there, non-null, error-free."""
    scale = ["--rows", "200000", "--sparse-chunks", "200000", "--e2e-chunks", "2000", "--embed-steps", "2", "--cpu-seconds", "0.5",
             "--strict", "1"]
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + DRIVER_CMD + scale, cwd=ROOT, env=dict(os.environ),
                       capture_output=True, text=True, timeout=900)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (p.returncode, p.stdout[-1500:], p.stderr[-3000:])
    d = json.loads(lines[0])
    assert d["leg_errors"] is None and p.returncode == 0, (d["leg_errors"], p.stderr[-3000:])
    assert d["steps"] == 20 and d["warmup"] == 5 and d["n_gpus"] == 1 and d["value"] > 0
    for key in TOP_LEVEL:
        assert d.get(key), key
        assert not _errors_in(d[key]), (key, _errors_in(d[key]))
    assert d["sparse_index"]["hybrid"]["two_threads_ms_per_query"] > 0            # the sub-leg that crashed round 4
    assert d["roofline"]["frac"] > 0 and d["cpu_baseline"]["value"] > 0
    assert d["embed"]["cpu_baseline"]["chunks_per_sec"] > 0 and d["e2e"]["recall_vs_cpu_oracle"]["R@5"] > 0.9


def test_a_failing_leg_does_not_cost_the_line(hip):
    """VERDICT r04 #1b: an exception inside a leg becomes {"error": ...} under that leg's key; headline, roofline and
    cpu_baseline still print; `--strict 1` turns it into exit status 3 AFTER the line."""
    env = dict(os.environ)
    env["CQS_BENCH_FAIL_LEG"] = "concurrent_clients"          # test hook of Legs.run: that leg raises before it starts
    args = ["--rows", "100000", "--embed-steps", "0", "--e2e-chunks", "0", "--sparse-chunks", "0", "--cpu-seconds", "0.3",
            "--abi-devices", ""]
    for strict, rc in (("0", 0), ("1", 3)):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + DRIVER_CMD + args + ["--strict", strict], cwd=ROOT, env=env,
                           capture_output=True, text=True, timeout=600)
        lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
        assert p.returncode == rc and len(lines) == 1, (p.returncode, p.stderr[-2000:])
        d = json.loads(lines[0])
        assert "concurrent_clients" in d["leg_errors"] and "error" in d["concurrent_clients"]
        assert d["value"] > 0 and d["roofline"]["frac"] > 0 and d["cpu_baseline"]["value"] > 0 and d["other_configs"]["k500_1M"]["checked"]


def test_a_hanging_leg_does_not_cost_the_line_either(hip):
    """A leg that never returns (a hung kernel, a deadlocked collective) cannot be caught as an exception: past --hard-limit-s
    (540 s by default, the driver's limit being 600) a watchdog thread prints the line with everything measured so far,
    names the leg in `incomplete` / `leg_errors`, and ends the process."""
    env = dict(os.environ)
    subprocess.run([sys.executable, "-c", "import torch"], env=env, timeout=600)      # (the limit counts from process start: page torch in first)
    env["CQS_BENCH_HANG_LEG"] = "other_configs"              # test hook of Legs.run: that leg sleeps forever
    args = ["--rows", "100000", "--embed-steps", "0", "--e2e-chunks", "0", "--sparse-chunks", "0", "--cpu-seconds", "0.3",
            "--abi-devices", "", "--hard-limit-s", "45"]
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + DRIVER_CMD + args, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, (p.returncode, p.stderr[-2000:])
    d = json.loads(lines[0])
    assert "other_configs" in d["incomplete"] and "other_configs" in d["leg_errors"]
    assert d["value"] > 0 and d["roofline"]["frac"] > 0 and d["latency_host_api"]["ms_per_query"] > 0
    assert d["concurrent_clients"]["native_threads"]["8"]["checked"] and d["other_configs"] is None
