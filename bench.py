#!/usr/bin/env python3
"""bench.py — headline benchmark of the cqs hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W        (N=1 default; N>1 under torch.distributed.run)

N = 1 (BASELINE.json configs[1], what the metric is quoted on): synthetic 1 000 000 x 768 fp32 unit vectors
resident in HBM, single-query brute-force cosine top-20.  A "step" = one query scanned against the corpus
(3.072 GB of HBM reads) + the exact top-k.  value = queries/s.

N > 1 (BASELINE.json configs[4], `--mode strong`, the default): ONE fixed corpus of 10 000 000 x 768 rows cut
row-wise over the N ranks (10M / N rows per GPU), one query per step scanned by every shard, ONE RCCL all-gather
of the per-shard k x u64 candidates, host merge (north_star).  "scaling": "strong".  `--mode weak` keeps round 1's
variant (1M rows per GPU, N queries per step) for comparison.

Inputs (corpus, queries) are resident in HBM before the timed region.  Rank 0 prints ONE JSON line with the
driver's contract fields plus `roofline`, `cpu_baseline`, `latency_host_api`, `other_configs` (the remaining
BASELINE / SURVEY §8d configurations, each CHECKED outside its timed region), `embed` (index-build leg, with its
own cpu_baseline) and `e2e` (configs[3]: chunks -> embed -> index -> k-NN with R@K against the CPU oracle).

Every leg beside the headline lives in its own module under `bench_legs/` and runs under `Legs.run`: an exception in a leg
becomes `{"error": ...}` under that leg's key (and a `leg_errors` entry), a leg that would start past `--budget-s` becomes
`{"skipped": ...}` - the headline, `roofline` and `cpu_baseline` print either way.  No leg borrows the headline's
`--steps` / `--warmup` arrays (round 4's crash); each makes its own inputs.
"""
import argparse
import json
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
_T_PROCESS = time.perf_counter()       # (before `import torch`: the driver's clock runs from process start)



def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--mode", choices=["auto", "single", "strong", "weak"], default="auto",
                    help="auto: single at N=1, strong (configs[4]) at N>1")
    ap.add_argument("--rows", type=int, default=1_000_000, help="corpus rows per GPU (single / weak modes)")
    ap.add_argument("--total-rows", type=int, default=10_000_000, help="whole-corpus rows (strong mode)")
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--batch", type=int, default=1, help="queries per step (per rank in weak mode)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of each cpu_baseline leg (0 = skip)")
    ap.add_argument("--extras", type=int, default=1, help="N=1 only: also time + check the other BASELINE configs "
                    "(k=500, 17.5k rows, 256-query blocks, 10M rows) into `other_configs` (0 = skip)")
    ap.add_argument("--embed-steps", type=int, default=24, help="timed embedding batches per rank (0 = skip the embed leg); "
                    "the 4 x batch leg runs a quarter of them (round 3 timed 8 / 2: mostly pipeline fill and drain)")
    ap.add_argument("--embed-batch", type=int, default=32, help="sequences per embedding batch (reference: embed_batch_size() = 32)")
    ap.add_argument("--embed-len", type=int, default=512, help="tokens per sequence of the fixed-length embed leg")
    ap.add_argument("--e2e-chunks", type=int, default=100_000, help="configs[3]: chunks embedded + indexed end to end (0 = skip)")
    ap.add_argument("--sparse-chunks", type=int, default=1_000_000, help="N=1 only: chunks of the sparse (SPLADE) index leg (0 = skip)")
    ap.add_argument("--abi-after", type=int, default=1, help="N>1 (RCCL) only: after the timed region rank 0 also runs the "
                    "single-process sharded handle over devices 0..N-1 into `abi_sharded` (0 = skip)")
    ap.add_argument("--abi-devices", type=str, default="0,0,0,0", help="N=1 only: also run the single-process sharded index "
                    "(cqs_hip_index_create_sharded) over this comma-separated device list, e.g. 0,1,2,3; the default names "
                    "device 0 four times (the one-GPU form: four 250k-row shards, per-shard scan, gather, host merge); '' = skip")
    ap.add_argument("--budget-s", type=float, default=400.0, help="N=1: an auxiliary leg that would START later than this many seconds "
                    "into the run is skipped (recorded as such), so that the line prints inside the driver's limit; the two "
                    "cpu_baseline legs are not subject to it (they are bounded by --cpu-seconds)")
    ap.add_argument("--child", type=str, default="", help="internal: `after_group` = run the N > 1 line's after-the-group legs (the "
                    "single-process sharded handle over devices 0..N-1, the N = 1 point of the strong-scaling curve) in THIS process and "
                    "print their results as one JSON object - rank 0 starts it as a child so that a crash in there cannot cost the line")
    ap.add_argument("--hard-limit-s", type=float, default=540.0, help="rank 0: if the process is still running this many seconds after it "
                    "started (a leg that hangs rather than raises), a watchdog thread prints the line with what has been measured so far "
                    "(`incomplete` names the leg that was running) and ends the process - the driver's limit is 600 s; 0 = no watchdog")
    ap.add_argument("--strict", type=int, default=0, help="1: exit with status 3 AFTER printing the line when any leg errored "
                    "(what tests/test_bench_modes_gpu.py runs); default 0 keeps status 0 so that a measured headline is never "
                    "discarded over an auxiliary leg - the errors are in the line (`leg_errors`) either way")
    return ap.parse_args()

from bench_legs.common import HBM_PEAK_GBS, MFMA_F32_PEAK_TF, check_topk, file_sha256, make_unit_rows  # noqa: E402


class Legs:
    """Failure isolation of the auxiliary legs: the one JSON line must survive any of them."""

    def __init__(self, budget_s):
        self.t0 = time.perf_counter()
        self.budget_s = budget_s
        self.errors = {}
        self.seconds = {}
        self.running = None

    def run(self, name, fn, *args, budgeted=True, **kw):
        now = time.perf_counter() - self.t0
        if budgeted and now > self.budget_s:
            self.seconds[name] = 0.0
            return {"skipped": "leg would start %.0f s into the run, past --budget-s %.0f" % (now, self.budget_s)}
        t = time.perf_counter()
        self.running = name
        try:
            if os.environ.get("CQS_BENCH_FAIL_LEG") == name:       # test hook (tests/test_bench_modes_gpu.py)
                raise RuntimeError("CQS_BENCH_FAIL_LEG")
            if os.environ.get("CQS_BENCH_HANG_LEG") == name:       # test hook: a leg that never returns
                while True:
                    time.sleep(1.0)
            return fn(*args, **kw)
        except KeyboardInterrupt:
            raise
        except BaseException as e:      # noqa: BLE001 - the line must survive
            traceback.print_exc(file=sys.stderr)
            msg = "%s: %s" % (type(e).__name__, str(e)[:300])
            self.errors[name] = msg
            return {"error": msg}
        finally:
            self.seconds[name] = round(time.perf_counter() - t, 1)
            self.running = None


def host_api_latency(idx, torch, k, dim, dev):
    """What `VectorIndex::search` sees: the synchronous host-buffer entry point (query H2D + results D2H + sync), 200 queries
    of the leg's own."""
    qh = make_unit_rows(torch, 200, dim, 0xC950032, dev).cpu().numpy()
    for j in range(10):
        idx.search_batch(qh[j], k)
    t1 = time.perf_counter()
    for j in range(len(qh)):
        idx.search_batch(qh[j], k)
    el = time.perf_counter() - t1
    return {"queries_per_sec": round(len(qh) / el, 1), "ms_per_query": round(el / len(qh) * 1e3, 4), "queries": len(qh),
            "what": "cqs_hip_index_search, host query in / host results out, one call at a time"}


def after_group_child(a):
    """`--child after_group`: the two legs rank 0 of an N > 1 run wants after its process group is gone, in a process of their
    own (an RCCL clique over N distinct devices inside the library has never run on hardware: an abort in there must not
    take the N > 1 line with it).  Prints {"abi": ..., "strong_n1": ...}."""
    import numpy as np
    import torch
    from bench_legs.sharded import abi_after_group_leg, strong_n1_leg
    out = {"abi": None, "strong_n1": None}
    out["abi"] = abi_after_group_leg(a, torch, np, a.gpus, a.k, a.dim)
    if a.mode == "strong" and not (out["abi"] or {}).get("_hung"):
        out["strong_n1"] = strong_n1_leg(a, torch, a.k, a.dim, a.total_rows)
    print(json.dumps(out), flush=True)
    os._exit(0)            # (a leg left running in its watchdog thread must not keep the child alive)


def run_after_group_child(a, world, k, dim, total_rows, mode, budget_s=330.0):
    """Rank 0: start `--child after_group` and read its one JSON object; whatever happens to the child becomes an `error`."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--child", "after_group", "--gpus", str(world), "--k", str(k), "--dim", str(dim),
           "--total-rows", str(total_rows), "--mode", mode, "--extras", str(a.extras)]
    try:
        p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=budget_s)
    except subprocess.TimeoutExpired:
        return {"error": "after-group child exceeded its %.0f s budget and was killed" % budget_s, "_hung": True}, None
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if p.returncode != 0 or not lines:
        return {"error": "after-group child ended with status %d: %s" % (p.returncode, (p.stderr or "")[-400:].replace("\n", " | "))}, None
    try:
        d = json.loads(lines[-1])
    except ValueError as e:
        return {"error": "after-group child printed no JSON: %s" % e}, None
    return d.get("abi"), d.get("strong_n1")


def main():
    a = parse()
    if a.child == "after_group":
        return after_group_child(a)
    import numpy as np
    import torch
    from cqs_amd import HipIndex, unpack_keys
    legs = Legs(a.budget_s)
    from cqs_amd.sharded import ShardedSearch, shard_bounds

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run --nproc-per-node N")
    # CQS_BENCH_REHEARSAL=1: rehearse the N>1 code path on ONE GPU (all ranks on cuda:0, gloo collectives
    # staged through host memory).  Logic check only - never a performance number.
    rehearsal = os.environ.get("CQS_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # CQS_BENCH_FORCE_DIST=1 (with the torch.distributed.run env of a 1-rank launch): run the N>1 code with a
    # real RCCL process group of size 1 - exercises the collective API calls on the one GPU a dev box has.
    force_dist = os.environ.get("CQS_BENCH_FORCE_DIST") == "1" and "MASTER_ADDR" in os.environ
    if world > 1 or force_dist:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    # Self-proving N > 1 record: the size of the group as the COLLECTIVE saw it (an all-reduce of ones over the backend that
    # carries the data path) and the physical device behind every rank (ordinal + PCI bus id, gathered the same way).
    ranks_seen, rank_devices, collective = None, None, None
    if dist is not None:
        cdev = "cpu" if rehearsal else dev
        ones = torch.ones(1, dtype=torch.int64, device=cdev)
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())
        props = torch.cuda.get_device_properties(local_rank)
        me = torch.tensor([local_rank, int(getattr(props, "pci_domain_id", -1)), int(getattr(props, "pci_bus_id", -1)),
                           int(getattr(props, "pci_device_id", -1))], dtype=torch.int64, device=cdev)
        allp = torch.zeros(world * 4, dtype=torch.int64, device=cdev)
        dist.all_gather_into_tensor(allp, me)
        rank_devices = [{"rank": r, "device": int(v[0]), "pci": "%04x:%02x:%02x" % (int(v[1]) & 0xFFFF, int(v[2]) & 0xFF, int(v[3]) & 0xFF)}
                        for r, v in enumerate(allp.view(world, 4).cpu().tolist())]
        collective = "gloo-host (rehearsal)" if rehearsal else "rccl (torch.distributed backend nccl)"
        assert ranks_seen == dist.get_world_size() == world, (ranks_seen, world)
    mode = a.mode
    if mode == "auto":
        mode = "strong" if (world > 1 or force_dist) else "single"
    if mode == "single" and world > 1:
        raise SystemExit("--mode single needs --gpus 1")

    def all_gather_dev(out, inp):
        """One fused all-gather of a device tensor (RCCL over xGMI; host-staged gloo in rehearsal mode)."""
        # flat views: output numel = world x input numel is the one shape contract every backend accepts
        if dist is None:
            out.view(-1).copy_(inp.contiguous().view(-1))
        elif rehearsal:
            o = torch.empty((out.numel(),), dtype=out.dtype)
            dist.all_gather_into_tensor(o, inp.cpu().contiguous().view(-1))
            out.view(-1).copy_(o)
        else:
            dist.all_gather_into_tensor(out.view(-1), inp.contiguous().view(-1))

    def all_reduce(x, op):
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(t, op=op)
        return float(t.item())

    def all_reduce_max(x):
        return all_reduce(x, dist.ReduceOp.MAX)

    dim, k, bq = a.dim, a.k, a.batch
    K, W = a.steps, a.warmup
    if mode == "strong":
        total_rows = a.total_rows
        lo, hi = shard_bounds(total_rows, world, rank)
        n = hi - lo
        row_base = lo
        # every rank holds the SAME query stream (resident in HBM before the timed region), its own row shard
        queries = make_unit_rows(torch, (K + W) * bq, dim, 0xC950002, dev).view(K + W, bq, dim)
    else:
        n = a.rows
        total_rows = n * world
        row_base = rank * n
        queries = make_unit_rows(torch, (K + W) * bq, dim, 0xC950002 + 7919 * rank, dev).view(K + W, bq, dim)
    rows = make_unit_rows(torch, n, dim, 0xC950001 + rank, dev)
    idx = HipIndex.build_from_device(None, rows.data_ptr(), n, dim, device=local_rank, row_base=row_base,
                                     borrow=True, keepalive=rows)
    st = torch.cuda.current_stream().cuda_stream

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    merged = {}
    if mode == "single":
        nq_scan = bq
        out_keys = torch.zeros((K + W, bq, k), dtype=torch.int64, device=dev)
        out_counts = torch.zeros((K + W, bq), dtype=torch.int32, device=dev)

        def step(i):
            idx.search_device(queries[i].data_ptr(), bq, k, out_keys[i].data_ptr(), out_counts[i].data_ptr(), stream=st)

        def finish(lo_, hi_):
            torch.cuda.synchronize()

        def scan_only(i):
            step(i)
    elif mode == "strong":
        # configs[4]: step i = every shard scans the SAME query block; ONE all-gather of the shards' [bq, k] packed
        # keys, issued asynchronously on RCCL's stream so that it runs under the scan of step i+1; the host merge of
        # all steps closes the timed region.  No host sync inside a step.
        nq_scan = bq
        send = torch.zeros((K + W, bq * k), dtype=torch.int64, device=dev)
        recv = torch.zeros((K + W, world, bq * k), dtype=torch.int64, device=dev)
        counts = torch.empty((bq,), dtype=torch.int32, device=dev)
        works = [None] * (K + W)

        def step(i):
            idx.search_device(queries[i].data_ptr(), bq, k, send[i].data_ptr(), counts.data_ptr(), stream=st)
            if dist is None or rehearsal:
                all_gather_dev(recv[i], send[i])
            else:
                works[i] = dist.all_gather_into_tensor(recv[i].view(-1), send[i].view(-1), async_op=True)

        def finish(lo_, hi_):
            for wk in works[lo_:hi_]:
                if wk is not None:
                    wk.wait()
            torch.cuda.synchronize()
            host = recv[lo_:hi_].cpu().numpy().reshape(hi_ - lo_, world, bq, k)
            out = ShardedSearch.merge_host_many(host, k)       # [steps, bq, k]: the host-side merge (north_star)
            for s in range(hi_ - lo_):
                merged[lo_ + s] = [row[row != 0] for row in out[s]]

        def scan_only(i):
            idx.search_device(queries[i].data_ptr(), bq, k, send[i].data_ptr(), counts.data_ptr(), stream=st)
    else:
        # weak (round 1): a rank's payload = its shard's top-k keys for ALL of this step's queries followed by its OWN
        # queries of step i+2 (fp32 viewed as int64): the exchange of step i delivers the query block of step i+2 and
        # runs under the scan of step i+1.
        nq = world * bq
        nq_scan = nq
        qw = dim // 2                                   # int64 words per query row
        pay = nq * k + bq * qw
        send = torch.zeros((K + W, pay), dtype=torch.int64, device=dev)
        recv = torch.zeros((K + W, world, pay), dtype=torch.int64, device=dev)
        qall = torch.empty((K + W + 2, nq, dim), dtype=torch.float32, device=dev)
        counts = torch.empty((nq,), dtype=torch.int32, device=dev)
        works = [None] * (K + W)
        for j0 in range(min(2, K + W)):                                    # prologue: the first two query blocks
            all_gather_dev(qall[j0].view(world, bq, dim), queries[j0])

        def step(i):
            if i >= 2:
                if works[i - 2] is not None:
                    works[i - 2].wait()                 # the compute stream waits; the host does not
                qall[i].view(world, bq, dim).copy_(recv[i - 2, :, nq * k:].view(torch.float32).view(world, bq, dim))
            idx.search_device(qall[i].data_ptr(), nq, k, send[i].data_ptr(), counts.data_ptr(), stream=st)
            if i + 2 < K + W:
                send[i, nq * k:].view(torch.float32).view(bq, dim).copy_(queries[i + 2])
            if dist is None or rehearsal:
                all_gather_dev(recv[i], send[i])
            else:
                works[i] = dist.all_gather_into_tensor(recv[i].view(-1), send[i].view(-1), async_op=True)

        def finish(lo_, hi_):
            for wk in works[lo_:hi_]:
                if wk is not None:
                    wk.wait()
            torch.cuda.synchronize()
            host = recv[lo_:hi_, :, :nq * k].cpu().numpy().reshape(hi_ - lo_, world, nq, k)
            mine = host[:, :, rank * bq:(rank + 1) * bq, :]    # host merge of this rank's own queries, all steps at once
            out = ShardedSearch.merge_host_many(mine, k)       # [steps, bq, k]
            for s in range(hi_ - lo_):
                merged[lo_ + s] = [row[row != 0] for row in out[s]]

        def scan_only(i):
            idx.search_device(qall[i].data_ptr(), nq, k, send[i].data_ptr(), counts.data_ptr(), stream=st)

    # untimed pre-warm beyond the driver's W (clocks, caches, lazy allocations): with a small K the first steps after a
    # cold start would otherwise be the measurement; step 0 is simply repeated, its outputs are rewritten below
    prewarm_steps = 200 if dist is None else 20       # (a FIXED count: every rank must issue the same collectives)
    for _ in range(prewarm_steps):
        step(0)
        finish(0, 1)
    for i in range(W):
        step(i)
    finish(0, W)
    barrier()
    t0 = time.perf_counter()
    for i in range(W, W + K):
        step(i)
    finish(W, W + K)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        elapsed = all_reduce_max(elapsed)

    # ---- the last step's answer, checked outside the timed region ----
    i = W + K - 1
    q_last = queries[i, 0]
    if mode == "single":
        check_topk(torch, np, rows, q_last, out_keys[i, 0].cpu().numpy().view(np.uint64), int(out_counts[i, 0].item()), k,
                   what="headline")
    else:
        r, s = unpack_keys(merged[i][0])
        assert len(r) == k and np.all(np.diff(s) <= 0), "merged top-k not sorted / short"
        mine = [(int(x) - row_base, float(sv)) for x, sv in zip(r, s) if row_base <= int(x) < row_base + n]
        if mine:
            loc = torch.tensor([m[0] for m in mine], device=dev)
            direct = (rows[loc].double() @ q_last.double()).cpu().numpy()
            err = float(np.max(np.abs(direct - np.array([m[1] for m in mine]))))
            assert err <= 1e-5, "rank %d: merged scores differ from a direct fp64 dot by %g" % (rank, err)
        # exhaustive over ALL shards: at most k-1 rows of the whole corpus beat the merged k-th score of each rank's
        # query (strong mode: every rank holds the same query; weak mode: one query per rank)
        q_all = torch.empty((world, dim), dtype=torch.float32, device=dev)
        kth_all = torch.empty((world,), dtype=torch.float32, device=dev)
        all_gather_dev(q_all, q_last.contiguous())
        all_gather_dev(kth_all, torch.tensor([float(s[-1])], dtype=torch.float32, device=dev))
        beat = ((rows @ q_all.T) > (kth_all + 2e-6)[None, :]).sum(dim=0).to(torch.float64)
        if dist is not None:
            bt = beat.cpu() if rehearsal else beat
            dist.all_reduce(bt, op=dist.ReduceOp.SUM)
            beat = bt
        assert float(beat[rank].item()) <= k - 1, "%d rows of the sharded corpus beat the merged k-th score" % int(beat[rank].item())

    # ---- roofline: the scan kernel's own duration, HIP events on the launch stream ----
    idx.set_timing(True)
    for i in range(W, W + K):
        scan_only(i)
    torch.cuda.synchronize()
    launches, scan_ms = idx.scan_time()
    idx.set_timing(False)
    avg_s = scan_ms / max(launches, 1) / 1e3
    alg_bytes = n * dim * 4  # algorithmic bytes per launch = shard rows x dim x 4 B (corpus read once)
    achieved = alg_bytes / avg_s / 1e9
    if nq_scan >= 9 and dim % 32 == 0:
        # query blocks >= 9 run on the f32 matrix cores: compute-bound (2*B*n*dim flops per launch)
        flops = 2.0 * nq_scan * n * dim
        tf = flops / avg_s / 1e12
        roofline = {"bound": "mfma", "kernel": "scan_mfma_kernel", "achieved": round(tf, 2), "peak": MFMA_F32_PEAK_TF,
                    "unit": "TFLOP/s", "frac": round(tf / MFMA_F32_PEAK_TF, 4), "traffic": None,
                    "alg_flops_per_launch": flops, "avg_launch_ms": round(avg_s * 1e3, 5), "launches": launches}
    else:
        # traffic: NOT measured in this run (PMC needs rocprofv3 passes of their own).  The committed counter
        # summary for this exact workload, if any, is quoted with its source; otherwise null.
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "scan_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("alg_bytes_per_launch") == alg_bytes and nq_scan == 1:
                    # NOT measured in this run: the committed PMC summary, with the commit it was measured at and the
                    # kernel instantiation it measured - a scan-kernel change after that commit invalidates it
                    traffic = tj.get("hbm_bytes_per_launch")
                    traffic_src = {"file": "profiles/scan_traffic.json", "how": tj.get("source"),
                                   "measured_at_commit": tj.get("commit"), "kernel": tj.get("kernel"),
                                   "scan_kernels_hip_sha256_then": tj.get("scan_kernels_sha256"),
                                   "scan_kernels_hip_sha256_now": file_sha256(os.path.join(ROOT, "cqs_amd", "csrc", "scan_kernels.hip")),
                                   "measured_in_this_run": False}
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": "scan_gemv_kernel", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "traffic_source": traffic_src, "alg_bytes_per_launch": alg_bytes,
                    "avg_launch_ms": round(avg_s * 1e3, 5), "launches": launches}


    # ---- the auxiliary legs: N = 1 only, each in its own module, each under legs.run ----
    n1 = rank == 0 and world == 1 and mode == "single"
    latency = clients = other = abi = strong_n1 = sparse = aux = e2e = cpu = None
    embed = eng = ecfg = eweights = None

    def build_line():
        """The one JSON line from what has been measured SO FAR (called at the end - and by the watchdog below if a leg hangs)."""
        if mode == "weak":
            total_q = K * bq * world
            workload = ("weak scaling (round-1 variant): %d x %d fp32 unit vectors per GPU, %d quer%s per rank per step, "
                        "brute-force cosine top-%d" % (n, dim, bq, "y" if bq == 1 else "ies", k))
        elif mode == "strong":
            total_q = K * bq
            workload = ("BASELINE configs[4]: %d x %d fp32 unit vectors row-sharded over %d GPU%s, %d quer%s per step scanned "
                        "by every shard, one RCCL all-gather of per-shard top-%d, host merge"
                        % (total_rows, dim, world, "" if world == 1 else "s", bq, "y" if bq == 1 else "ies", k))
        else:
            total_q = K * bq
            workload = ("BASELINE configs[1]: %d x %d fp32 unit vectors, %d quer%s per step, brute-force cosine top-%d"
                        % (n, dim, bq, "y" if bq == 1 else "ies", k))
        line = {
            "metric": "queries/sec @k=%d (brute-force cosine scan + top-k, 768-d fp32)" % k,
            "value": round(total_q / elapsed, 2),
            "unit": "queries/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "prewarm_steps": prewarm_steps,          # untimed repeats of step 0 before the W warm-ups (clocks, caches)
            "ms_per_step": round(elapsed / K * 1e3, 5),
            "higher_is_better": True,
            "scaling": "weak" if mode in ("weak", "single") else "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": workload, "mode": mode, "rows_per_gpu": n, "total_rows": total_rows, "dim": dim, "k": k,
                       "queries_per_step": bq * (world if mode == "weak" else 1),
                       "parallelism": ("row-sharded x%d, %s all-gather of per-shard candidates, host merge"
                                       % (world, "gloo (host-staged, REHEARSAL)" if rehearsal else "RCCL")) if world > 1 else "single GPU",
                       "ranks_seen": ranks_seen if ranks_seen is not None else 1,     # group size as the collective's all-reduce saw it
                       "rank_devices": rank_devices if rank_devices is not None else [{"rank": 0, "device": local_rank}],
                       "collective": collective},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "latency_host_api": latency,
            "concurrent_clients": clients,
            "other_configs": other,
            "abi_sharded": abi,
            "strong_scaling_n1": strong_n1,
            "embed": embed if not isinstance(embed, tuple) else None,
            "e2e": e2e,
            "aux_models": aux,
            "sparse_index": sparse,
            "leg_errors": dict(legs.errors) or None,      # leg name -> message; the same message sits under that leg's own key
            "leg_seconds": dict(legs.seconds),
            "wall_s": round(time.perf_counter() - legs.t0, 1),
        }
        if rehearsal or (force_dist and world == 1):
            # all ranks on one GPU over gloo, or a 1-rank RCCL group: the N > 1 LOGIC ran, nothing here measures N GPUs
            line["rehearsal"] = True
            line["metric"] = "REHEARSAL (logic check, not a measurement): " + line["metric"]
            line["vs_baseline"] = None
        return line

    # A leg that HANGS (rather than raises) must not cost the line either: past --hard-limit-s a watchdog prints what has been
    # measured (the headline and the roofline exist from here on) and ends the process.
    import threading
    print_lock = threading.Lock()
    printed = [False]
    if rank == 0 and a.hard_limit_s > 0:                 # (N > 1 too: rank 0 stuck in a collective still prints what it measured)
        def watchdog():
            while time.perf_counter() - _T_PROCESS < a.hard_limit_s:
                time.sleep(0.5)
                if printed[0]:
                    return
            with print_lock:
                if printed[0]:
                    return
                printed[0] = True
                line = build_line()
                hung = legs.running or "between legs"
                line["incomplete"] = "hard limit of %.0f s reached while `%s` was running; later legs did not run" % (a.hard_limit_s, hung)
                errs = dict(line["leg_errors"] or {})
                errs[hung] = "did not return within the run's hard limit"
                line["leg_errors"] = errs
                print(json.dumps(line), flush=True)
            os._exit(3 if a.strict else 0)
        threading.Thread(target=watchdog, daemon=True).start()
    if n1:
        latency = legs.run("latency_host_api", host_api_latency, idx, torch, k, dim, dev)
    if n1 and a.extras:
        from bench_legs.clients import concurrent_clients_leg
        clients = legs.run("concurrent_clients", lambda: concurrent_clients_leg(
            np, idx, make_unit_rows(torch, 96, dim, 0xC950031, dev).cpu().numpy(), k, dim, rows_ptr=rows.data_ptr()))

    # The CPU baseline runs LAST (after every GPU leg; only its inputs are taken here, while the corpus is still resident):
    # 10-20 s of one AVX-512 worker per physical core, pinned, over a first-touched 3 GB copy leave the host in a state in
    # which the ticketed embedding leg measured 3 % lower (6 072 against 6 198-6 242 chunks/s, same box, four orders tried,
    # round 4) - a baseline must not move the thing it is the baseline of.
    cpu_inputs = None
    if rank == 0 and world == 1 and a.cpu_seconds > 0:
        cpu_inputs = (rows.cpu().numpy(), make_unit_rows(torch, 8, dim, 0xC950033, dev).cpu().numpy())

    if n1 and a.extras:
        from bench_legs.other_configs import other_configs
        other = legs.run("other_configs", other_configs, torch, np, HipIndex, idx, rows, queries, dim, dev, st)
    if isinstance(other, dict) and isinstance(other.get("rows10M"), dict) and "queries_per_sec" in other["rows10M"]:
        # N = 1 (headline = configs[1], 1M rows): the N = 1 point of the strong-scaling curve the N > 1 lines measure
        # (configs[4]: ONE 10M-row corpus) is the 10M-row extra of this very run
        strong_n1 = {"n_gpus": 1, "rows": 10_000_000, "value": other["rows10M"]["queries_per_sec"], "unit": "queries/s",
                     "ms_per_step": other["rows10M"]["ms_per_query"],
                     "note": "= other_configs.rows10M: the corpus of the N > 1 lines (`--mode strong`, BASELINE configs[4]) on ONE GPU; "
                             "scale N > 1 values against THIS number, not against the headline `value` (1M rows)"}
    if n1 and a.abi_devices:
        from bench_legs.sharded import abi_sharded_leg
        abi = legs.run("abi_sharded", lambda: abi_sharded_leg(
            a, torch, np, rows, make_unit_rows(torch, 200, dim, 0xC950034, dev).view(200, 1, dim), k, dim))
    # N > 1 under RCCL: the C ABI's single-process sharded handle (what the Rust daemon binds) gets its first
    # multi-device run here, on rank 0, after every rank has let go of its shard and left the group - see below.
    abi_after_group = dist is not None and not rehearsal and a.abi_after and torch.cuda.device_count() >= world

    if n1 and a.extras and a.sparse_chunks > 0:
        from bench_legs.sparse import sparse_index_leg
        sparse = legs.run("sparse_index", sparse_index_leg, a, np, idx)

    if a.embed_steps > 0:
        from bench_legs.embed import embed_cpu_baseline, embed_leg
        idx.close()
        del rows
        torch.cuda.empty_cache()
        if world == 1:
            got = legs.run("embed", embed_leg, a, rank, world, dist, torch, np, dev, all_reduce_max, budgeted=False)
            if isinstance(got, tuple):
                embed, eng, ecfg, eweights = got
            else:
                embed = got
        else:       # every rank must take part in the leg's barriers: no isolation here
            embed, eng, ecfg, eweights = embed_leg(a, rank, world, dist, torch, np, dev, all_reduce_max)
    if n1 and a.extras and a.embed_steps > 0:
        from bench_legs.aux_models import aux_models_leg
        aux = legs.run("aux_models", aux_models_leg, a, np)   # (before the end-to-end leg: that one ends with a minute of CPU forwards for its recall check)
    if eng is not None:
        if rank == 0 and world == 1 and a.e2e_chunks > 0:
            from bench_legs.e2e import e2e_leg
            e2e = legs.run("e2e", e2e_leg, a, torch, np, dev, eng, ecfg, eweights)
        legs.run("embed_close", eng.close, budgeted=False)

    if cpu_inputs is not None:
        from bench_legs.scan_cpu import cpu_baseline
        cpu = legs.run("cpu_baseline", cpu_baseline, cpu_inputs[0], cpu_inputs[1], k, a.cpu_seconds, budgeted=False)
        cpu_inputs = None
    if isinstance(embed, dict) and "error" not in embed and eweights is not None and rank == 0 and world == 1 and a.cpu_seconds > 0:
        embed["cpu_baseline"] = legs.run("embed_cpu_baseline", embed_cpu_baseline, np, ecfg, eweights, a.cpu_seconds, a.embed_len,
                                         budgeted=False)

    if abi_after_group:
        # every rank drops its shard, the group dissolves, ranks != 0 leave; rank 0 then builds ONE handle over
        # devices 0..N-1 with a corpus of its own (250k rows per device), checks it against the single-device answer
        # and times it through the blocking host API - in a watchdog thread: whatever happens in there (an RCCL clique
        # that cannot form, a hang), the line above is still printed.
        if a.embed_steps <= 0:               # (the embed leg has closed the index and dropped the rows already)
            idx.close()
            del rows
        del queries
        torch.cuda.empty_cache()
        dist.barrier()
        dist.destroy_process_group()
        dist = None
        if rank == 0:
            # in a CHILD process (round 5): the in-library RCCL clique over N distinct devices has never run on hardware - an
            # abort in there would have taken this rank, and the N > 1 line with it, down
            abi, strong_n1 = run_after_group_child(a, world, k, dim, total_rows, mode)
    if rank == 0:
        with print_lock:
            if not printed[0]:
                printed[0] = True
                print(json.dumps(build_line()), flush=True)
    if a.embed_steps <= 0 and not abi_after_group:
        idx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()

    if a.strict and legs.errors:
        sys.exit(3)


if __name__ == "__main__":
    main()
