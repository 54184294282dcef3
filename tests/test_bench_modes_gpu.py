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


def test_weak_mode_two_ranks(hip):
    d = _run(2, 29622, ["--mode", "weak", "--rows", "150000"], {"CQS_BENCH_REHEARSAL": "1"})
    assert d["scaling"] == "weak" and d["config"]["total_rows"] == 300000 and d["config"]["queries_per_step"] == 2


def test_strong_mode_one_rank_rccl(hip):
    d = _run(1, 29623, ["--total-rows", "300000"], {"CQS_BENCH_FORCE_DIST": "1"})
    assert d["scaling"] == "strong" and d["n_gpus"] == 1 and d["roofline"]["bound"] == "hbm"
