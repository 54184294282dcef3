#!/usr/bin/env python3
"""bench.py — headline benchmark of the cqs hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W        (N=1 default; N>1 under torch.distributed.run)

Workload (BASELINE.json configs[1]): synthetic 1 000 000 x 768 fp32 unit vectors per GPU,
single-query brute-force cosine top-20.  A "step" is one pass of the hot path over one
batch: at N=1 one query scanned against the 1M-row corpus (3.072 GB of HBM reads); at N>1
the corpus is row-sharded (1M rows per GPU, N x 1M rows in total - weak scaling), every
rank contributes one query per step, the N queries are all-gathered, each rank scans its
shard ONCE for all N queries, per-shard (score,row) candidates are exchanged with one RCCL
all-gather and merged on the host (north_star).  value = whole-job queries/s.

Inputs (corpus, queries) are resident in HBM before the timed region.  Rank 0 prints ONE
JSON line with the driver's contract fields plus `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TF = 157.3  # dense f32-input MFMA peak (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E vendor peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--rows", type=int, default=1_000_000, help="corpus rows per GPU")
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--batch", type=int, default=1, help="queries per rank per step")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of each cpu_baseline leg (0 = skip)")
    ap.add_argument("--extras", type=int, default=1, help="N=1 only: also time the other BASELINE configs (k=500, 17.5k rows, "
                    "256-query blocks, 10M rows) into `other_configs` (0 = skip)")
    ap.add_argument("--embed-steps", type=int, default=8, help="timed embedding batches per rank (0 = skip the embed leg)")
    ap.add_argument("--embed-batch", type=int, default=32, help="sequences per embedding batch (reference: embed_batch_size() = 32)")
    ap.add_argument("--embed-len", type=int, default=512, help="tokens per sequence of the fixed-length embed leg")
    return ap.parse_args()


def make_unit_rows(torch, n, dim, seed, device):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    rows = torch.empty((n, dim), dtype=torch.float32, device=device)
    step = 1 << 18
    for lo in range(0, n, step):  # chunked so the generator scratch stays small
        hi = min(n, lo + step)
        x = torch.randn((hi - lo, dim), generator=g, device=device, dtype=torch.float32)
        x /= x.norm(dim=1, keepdim=True)
        rows[lo:hi] = x
    return rows


def cpu_baseline(rows_host, queries_host, k, seconds):
    """The oracle (C restatement of the reference CPU scan, search/query.rs:453-482 minus SQLite)
    timed on this host: single-threaded = the reference's per-query behaviour; plus one thread
    per core over row shards.  Reported baseline only - never part of `value`."""
    from oracle import oracle
    n = rows_host.shape[0]
    out = {"unit": "queries/s", "kind": "port", "cores": 1}
    t0 = time.perf_counter()
    done = 0
    while True:
        oracle.brute_force(rows_host, queries_host[done % len(queries_host)], k, 0.0)
        done += 1
        el = time.perf_counter() - t0
        if el >= seconds and done >= 2:
            break
    out["value"] = done / el
    cores = os.cpu_count() or 1
    threads = min(cores, 64)
    t0 = time.perf_counter()
    done_mt = 0
    while True:
        oracle.brute_force_mt(rows_host, queries_host[done_mt % len(queries_host)], k, 0.0, threads)
        done_mt += 1
        el = time.perf_counter() - t0
        if el >= seconds and done_mt >= 2:
            break
    out["mt_value"] = done_mt / el
    out["mt_cores"] = threads
    out["host_cores"] = cores
    out["sample"] = (f"{done} single-thread + {done_mt} {threads}-thread queries, each a full scan of the same "
                     f"{n}x{rows_host.shape[1]} fp32 corpus held in RAM, k={k}, threshold 0.0 (oracle: "
                     "simsimd-style AVX2 dot + clamp + BoundedScoreHeap)")
    return out


def embed_leg(a, rank, world, dist, torch, np, dev, all_reduce_max):
    """Index-build leg of the metric ("index embed chunks/sec"): EmbeddingGemma-300m geometry with seeded
    random weights (no network for the real checkpoint), synthetic token ids, batch = the reference's
    embed_batch_size() (32), (a) fixed L and (b) log-normal lengths ("few hundred tokens").
    Data-parallel over ranks: replicated weights, no collective."""
    from cqs_amd.embedder import HipEmbedEngine, default_config
    cfg = default_config()
    eng = HipEmbedEngine(cfg, device=dev.index)
    rng = np.random.default_rng(0xC950003)
    H, D, I, V, NL = 768, 256, 1152, cfg.vocab_size, cfg.layers

    def lin(n, k):
        return rng.standard_normal((n, k), dtype=np.float32) * np.float32(1.0 / np.sqrt(k))

    eng.set_tensor("embed_tokens.weight", rng.standard_normal((V, H), dtype=np.float32) * np.float32(0.05))
    for l in range(NL):
        p = f"layers.{l}."
        for nme in ("input_layernorm", "post_attention_layernorm", "pre_feedforward_layernorm", "post_feedforward_layernorm"):
            eng.set_tensor(p + nme + ".weight", rng.standard_normal(H, dtype=np.float32) * np.float32(0.1))
        eng.set_tensor(p + "self_attn.q_norm.weight", rng.standard_normal(D, dtype=np.float32) * np.float32(0.1))
        eng.set_tensor(p + "self_attn.k_norm.weight", rng.standard_normal(D, dtype=np.float32) * np.float32(0.1))
        eng.set_tensor(p + "self_attn.q_proj.weight", lin(3 * D, H))
        eng.set_tensor(p + "self_attn.k_proj.weight", lin(D, H))
        eng.set_tensor(p + "self_attn.v_proj.weight", lin(D, H))
        eng.set_tensor(p + "self_attn.o_proj.weight", lin(H, 3 * D))
        eng.set_tensor(p + "mlp.gate_proj.weight", lin(I, H))
        eng.set_tensor(p + "mlp.up_proj.weight", lin(I, H))
        eng.set_tensor(p + "mlp.down_proj.weight", lin(H, I))
    eng.set_tensor("norm.weight", rng.standard_normal(H, dtype=np.float32) * np.float32(0.1))
    eng.set_tensor("dense1.weight", lin(3072, H))
    eng.set_tensor("dense2.weight", lin(H, 3072))
    eng.set_weights({})

    def flops_of(lens):
        gemm = 2.0 * NL * (H * 1280 + H * H + H * 2 * I + I * H)           # per token (0.203 GFLOP)
        att = 0.0
        W = cfg.sliding_window // 2 + 1
        for L in lens:
            pos = np.arange(L)
            local = np.minimum(pos + W, L) - np.maximum(pos - W + 1, 0)    # keys with |q-k| < W
            n_full = NL // cfg.sliding_pattern
            att += 4.0 * 3 * D * (n_full * L * L + (NL - n_full) * float(local.sum()))
        head = 2.0 * 2 * H * 3072 * len(lens)
        return gemm * float(np.sum(lens)) + att + head

    def run(lens, steps):
        B, L = len(lens), int(max(lens))
        ids = np.zeros((B, L), np.int64)
        mask = np.zeros((B, L), np.int64)
        for i, n in enumerate(lens):
            ids[i, :n] = rng.integers(1, V, size=n)
            mask[i, :n] = 1
        out = eng.run(ids, mask)                      # warm-up (also sizes the scratch)
        assert np.all(np.isfinite(out))
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dev_ms = 0.0
        for _ in range(steps):
            eng.run(ids, mask)
            dev_ms += eng.last_ms()
        el = time.perf_counter() - t0
        if dist is not None:
            el = all_reduce_max(el)
        toks = int(np.sum(lens))
        tf = flops_of(lens) * steps / (dev_ms / 1e3) / 1e12
        return {"chunks_per_sec": round(B * steps * world / el, 1), "tokens_per_sec": round(toks * steps * world / el, 1),
                "ms_per_batch": round(el / steps * 1e3, 3), "device_ms_per_batch": round(dev_ms / steps, 3),
                "batch": B, "tokens_per_batch": toks,
                "roofline": {"bound": "mfma", "achieved": round(tf, 1), "peak": 2500.0, "unit": "TFLOP/s",
                             "frac": round(tf / 2500.0, 4), "dtype": "bf16"}}

    fixed = run([a.embed_len] * a.embed_batch, a.embed_steps)
    lens = np.clip(np.exp(rng.normal(np.log(300.0), 0.6, size=a.embed_batch)).astype(int), 8, cfg.max_seq)
    ragged = run(list(lens), a.embed_steps)
    # SURVEY §8d asks for "a tuned larger batch" beside the reference's 32, and for the measured GEMM ceiling
    # of the bf16 kernel the forward is built on (a big square GEMM through the same kernel)
    big = run([a.embed_len] * (4 * a.embed_batch), max(2, a.embed_steps // 4)) if rank == 0 or dist is not None else None
    ceiling = None
    try:
        import ctypes as C
        from cqs_amd import _lib
        f = _lib.load().cqs_hip_debug_gemm_ms
        f.restype = C.c_float
        f.argtypes = [C.c_uint32] * 4 + [C.c_int32]
        ms = f(8192, 4096, 4096, 10, 0)
        ceiling = round(2.0 * 8192 * 4096 * 4096 / ms / 1e9, 1) if ms > 0 else None
    except Exception:
        ceiling = None
    eng.close()
    return {"model": "EmbeddingGemma-300m geometry (24 x [768 | 3x256 q, 1 kv | 1152], vocab 262144), seeded weights",
            "steps": a.embed_steps, "fixed_len_%d" % a.embed_len: fixed, "lognormal_len": ragged,
            "fixed_len_%d_batch%d" % (a.embed_len, 4 * a.embed_batch): big,
            "gemm_kernel_ceiling_tflops": ceiling,
            "note": "host-buffer API (ids in, embeddings out per batch, PCIe-inclusive); timed wall-clock, max over ranks"}


def other_configs(torch, np, HipIndex, make_unit_rows, idx, rows, queries, dim, dev, st):
    """The other BASELINE / SURVEY §8d configurations, timed the same way (inputs resident in HBM, device API,
    steps enqueued back to back).  Reported beside the headline, never instead of it."""
    def timed(index, q, b, k, steps, warm):
        keys = torch.zeros((b, k), dtype=torch.int64, device=dev)
        cnt = torch.zeros((b,), dtype=torch.int32, device=dev)
        for _ in range(warm):
            index.search_device(q.data_ptr(), b, k, keys.data_ptr(), cnt.data_ptr(), stream=st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            index.search_device(q.data_ptr(), b, k, keys.data_ptr(), cnt.data_ptr(), stream=st)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    out = {}
    n = rows.shape[0]
    q1 = queries[0].contiguous()
    t = timed(idx, q1, 1, 500, 100, 10)                       # what production asks for (src/limits.rs:315-320)
    out["k500_1M"] = {"queries_per_sec": round(1.0 / t, 1), "ms_per_query": round(t * 1e3, 4)}
    qb = make_unit_rows(torch, 256, dim, 0xC950003, dev)
    t = timed(idx, qb, 256, 20, 10, 2)                        # configs[2]: 256-query blocks on the f32 matrix cores
    tf = 2.0 * 256 * n * dim / t / 1e12
    out["batch256_1M"] = {"queries_per_sec": round(256 / t, 1), "ms_per_batch": round(t * 1e3, 3),
                          "roofline": {"bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                                       "frac": round(tf / MFMA_F32_PEAK_TF, 4), "dtype": "f32"}}
    small = make_unit_rows(torch, 17523, dim, 0xC950004, dev)  # configs[0] shape (cache resident: not judged against HBM)
    si = HipIndex.build_from_device(None, small.data_ptr(), 17523, dim, device=dev.index or 0, borrow=True, keepalive=small)
    t = timed(si, q1, 1, 20, 500, 50)
    out["rows17523"] = {"queries_per_sec": round(1.0 / t, 1), "ms_per_query": round(t * 1e3, 4)}
    si.close()
    del small
    try:
        big_n = 10_000_000
        big = make_unit_rows(torch, big_n, dim, 0xC950005, dev)
        bi = HipIndex.build_from_device(None, big.data_ptr(), big_n, dim, device=dev.index or 0, borrow=True, keepalive=big)
        t = timed(bi, q1, 1, 20, 20, 3)
        gbs = big_n * dim * 4 / t / 1e9
        out["rows10M"] = {"queries_per_sec": round(1.0 / t, 2), "ms_per_query": round(t * 1e3, 3),
                          "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": round(gbs / HBM_PEAK_GBS, 4), "note": "whole step incl. select"}}
        bi.close()
        del big
    except Exception as e:  # e.g. not enough free HBM beside another tenant
        out["rows10M"] = {"skipped": str(e)[:120]}
    torch.cuda.empty_cache()
    return out


def main():
    a = parse()
    import numpy as np
    import torch
    from cqs_amd import HipIndex, unpack_keys
    from cqs_amd.sharded import ShardedSearch, hip_local_search

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

    def all_reduce_max(x):
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    n, dim, k, bq = a.rows, a.dim, a.k, a.batch
    K, W = a.steps, a.warmup
    rows = make_unit_rows(torch, n, dim, 0xC950001 + rank, dev)
    queries = make_unit_rows(torch, (K + W) * bq, dim, 0xC950002 + 7919 * rank, dev).view(K + W, bq, dim)
    idx = HipIndex.build_from_device(None, rows.data_ptr(), n, dim, device=local_rank, row_base=rank * n,
                                     borrow=True, keepalive=rows)
    st = torch.cuda.current_stream().cuda_stream

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    sharded_path = world > 1 or force_dist or os.environ.get("CQS_BENCH_FORCE_SHARDED") == "1"   # the env knob exercises the N>1 code on 1 rank
    if not sharded_path:
        out_keys = torch.zeros((K + W, bq, k), dtype=torch.int64, device=dev)
        out_counts = torch.zeros((K + W, bq), dtype=torch.int32, device=dev)

        def step(i):
            idx.search_device(queries[i].data_ptr(), bq, k, out_keys[i].data_ptr(), out_counts[i].data_ptr(), stream=st)

        def finish(lo, hi):
            torch.cuda.synchronize()
    else:
        # One collective per step, off the critical path: a rank's payload = its shard's top-k keys for ALL of
        # this step's queries followed by its OWN queries of step i+2 (fp32 viewed as int64).  The exchange of
        # step i therefore delivers the query block of step i+2, and - issued asynchronously on RCCL's stream -
        # runs under the scan of step i+1, which only needs the exchange of step i-1.  No host sync.
        nq = world * bq
        qw = dim // 2                                   # int64 words per query row
        pay = nq * k + bq * qw
        send = torch.zeros((K + W, pay), dtype=torch.int64, device=dev)
        recv = torch.zeros((K + W, world, pay), dtype=torch.int64, device=dev)
        qall = torch.empty((K + W + 2, nq, dim), dtype=torch.float32, device=dev)
        counts = torch.empty((nq,), dtype=torch.int32, device=dev)
        works = [None] * (K + W)
        merged = {}
        for j0 in range(min(2, K + W)):                                    # prologue: the first two query blocks
            all_gather_dev(qall[j0].view(world, bq, dim), queries[j0])

        def exchange(i):
            if dist is None or rehearsal:
                all_gather_dev(recv[i], send[i])
                return None
            return dist.all_gather_into_tensor(recv[i].view(-1), send[i].view(-1), async_op=True)

        def step(i):
            if i >= 2:
                if works[i - 2] is not None:
                    works[i - 2].wait()                 # the compute stream waits; the host does not
                qall[i].view(world, bq, dim).copy_(recv[i - 2, :, nq * k:].view(torch.float32).view(world, bq, dim))
            idx.search_device(qall[i].data_ptr(), nq, k, send[i].data_ptr(), counts.data_ptr(), stream=st)
            if i + 2 < K + W:
                send[i, nq * k:].view(torch.float32).view(bq, dim).copy_(queries[i + 2])
            works[i] = exchange(i)

        def finish(lo, hi):
            for wk in works[lo:hi]:
                if wk is not None:
                    wk.wait()
            torch.cuda.synchronize()
            host = recv[lo:hi, :, :nq * k].cpu().numpy().reshape(hi - lo, world, nq, k)
            mine = host[:, :, rank * bq:(rank + 1) * bq, :]    # host merge of this rank's own queries, all steps at once
            out = ShardedSearch.merge_host_many(mine, k)       # [steps, bq, k]
            for s in range(hi - lo):
                merged[lo + s] = [row[row != 0] for row in out[s]]

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

    # ---- sanity of the last step's answer (outside the timed region) ----
    i = W + K - 1
    if not sharded_path:
        r, s = unpack_keys(out_keys[i, 0].cpu().numpy().view(np.uint64))
        assert int(out_counts[i, 0].item()) == k
    else:
        r, s = unpack_keys(merged[i][0])
        assert len(r) == k
    assert np.all(np.diff(s) <= 0), "top-k not sorted"
    local = [(int(x) - rank * n) for x in r if rank * n <= int(x) < (rank + 1) * n]
    if local:
        direct = (rows[torch.tensor(local, device=dev)].double() @ queries[i, 0].double()).cpu().numpy()
        mine = np.array([float(sv) for x, sv in zip(r, s) if rank * n <= int(x) < (rank + 1) * n])
        assert np.max(np.abs(direct - mine)) <= 1e-5, (
            "scores differ from a direct fp64 dot: rank %d max|d|=%g rows=%s got=%s want=%s"
            % (rank, float(np.max(np.abs(direct - mine))), list(r[:6]), list(mine[:4]), list(direct[:4])))

    # ---- roofline: the scan kernel's own duration, HIP events on the launch stream ----
    idx.set_timing(True)
    for i in range(W, W + K):
        if not sharded_path:
            step(i)
        else:
            idx.search_device(qall[i].data_ptr(), nq, k, send[i].data_ptr(), counts.data_ptr(), stream=st)
    torch.cuda.synchronize()
    launches, scan_ms = idx.scan_time()
    idx.set_timing(False)
    avg_s = scan_ms / max(launches, 1) / 1e3
    alg_bytes = n * dim * 4  # algorithmic bytes per launch = shard rows x dim x 4 B (corpus read once)
    achieved = alg_bytes / avg_s / 1e9
    traffic = None
    nq_scan_probe = bq * world
    tpath = os.path.join(ROOT, "profiles", "r01_scan_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            # PMC traffic was measured on the default workload only (rocprofv3 --pmc passes, profiles/)
            if tj.get("alg_bytes_per_launch") == alg_bytes and nq_scan_probe == 1:
                traffic = tj.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    nq_scan = bq * world
    if nq_scan >= 9 and dim % 32 == 0:
        # query blocks >= 9 run on the f32 matrix cores: compute-bound (2*B*n*dim flops per launch)
        flops = 2.0 * nq_scan * n * dim
        tf = flops / avg_s / 1e12
        roofline = {"bound": "mfma", "kernel": "scan_mfma_kernel", "achieved": round(tf, 2), "peak": MFMA_F32_PEAK_TF,
                    "unit": "TFLOP/s", "frac": round(tf / MFMA_F32_PEAK_TF, 4), "traffic": None,
                    "alg_flops_per_launch": flops, "avg_launch_ms": round(avg_s * 1e3, 5), "launches": launches}
    else:
        roofline = {"bound": "hbm", "kernel": "scan_gemv_kernel", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "alg_bytes_per_launch": alg_bytes, "avg_launch_ms": round(avg_s * 1e3, 5), "launches": launches}

    cpu = None
    if rank == 0 and world == 1 and a.cpu_seconds > 0:
        nq = min(8, K)
        cpu = cpu_baseline(rows.cpu().numpy(), queries[W:W + nq, 0].cpu().numpy(), k, a.cpu_seconds)

    other = None
    if rank == 0 and world == 1 and not sharded_path and a.extras:
        other = other_configs(torch, np, HipIndex, make_unit_rows, idx, rows, queries, dim, dev, st)

    embed = None
    if a.embed_steps > 0:
        idx.close()
        del rows
        torch.cuda.empty_cache()
        embed = embed_leg(a, rank, world, dist, torch, np, dev, all_reduce_max)

    if rank == 0:
        total_q = K * bq * world
        line = {
            "metric": "queries/sec @k=%d (brute-force cosine scan + top-k, 768-d fp32)" % k,
            "value": round(total_q / elapsed, 2),
            "unit": "queries/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": round(elapsed / K * 1e3, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: %d x %d fp32 unit vectors per GPU, %d quer%s per rank per step, "
                                   "brute-force cosine top-%d" % (n, dim, bq, "y" if bq == 1 else "ies", k),
                       "rows_per_gpu": n, "total_rows": n * world, "dim": dim, "k": k, "queries_per_step": bq * world,
                       "parallelism": "row-sharded x%d, RCCL all-gather of per-shard candidates, host merge" % world
                       if world > 1 else "single GPU"},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "other_configs": other,
            "embed": embed,
        }
        print(json.dumps(line), flush=True)
    if a.embed_steps <= 0:
        idx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
