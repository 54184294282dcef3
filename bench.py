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


def check_topk(torch, np, rows, q, keys_u64, count, k, row_base=0, exhaustive=True, what=""):
    """Size-independent properties of one query's answer (outside any timed region): full count, sorted by
    (score desc, row asc), scores equal a direct fp64 dot of the returned rows to 1e-5, and - exhaustively -
    no more than k-1 rows of the corpus beat the k-th score by more than 2e-6."""
    from cqs_amd import unpack_keys
    r, s = unpack_keys(np.ascontiguousarray(keys_u64))
    assert int(count) == k and len(r) == k, f"{what}: count {count} != {k}"
    assert np.all(np.diff(s) <= 0), f"{what}: not sorted"
    assert all(s[i] > s[i + 1] or r[i] < r[i + 1] for i in range(k - 1)), f"{what}: ties not ordered by row"
    local = torch.from_numpy((r.astype(np.int64) - row_base)).to(rows.device)
    direct = (rows[local].double() @ q.double()).cpu().numpy()
    err = float(np.max(np.abs(direct - s)))
    assert err <= 1e-5, f"{what}: scores differ from a direct fp64 dot by {err}"
    if exhaustive:
        beat = int(((rows @ q) > float(s[-1]) + 2e-6).sum().item())
        assert beat <= k - 1, f"{what}: {beat} rows beat the k-th score"
    return r, s


def file_sha256(path):
    import hashlib
    try:
        return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
    except OSError:
        return None


def physical_cores():
    """One logical CPU per physical core among the CPUs this process may run on (sysfs thread_siblings_list)."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        allowed = list(range(os.cpu_count() or 1))
    seen, out = set(), []
    for c in allowed:
        try:
            sib = open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c).read().strip()
        except OSError:
            sib = str(c)
        if sib not in seen:
            seen.add(sib)
            out.append(c)
    return out


def cpu_baseline(rows_host, queries_host, k, seconds):
    """The oracle (C restatement of the reference CPU scan, search/query.rs:453-482 minus SQLite) timed on this
    host with the dot body simsimd's run-time dispatch would take here (AVX-512 / AVX2+FMA / scalar - the ISA
    that ran is in `isa`): single-threaded = the reference's per-query behaviour; plus one thread per core over
    row shards.  Reported baseline only - never part of `value`."""
    from oracle import oracle
    n = rows_host.shape[0]
    out = {"unit": "queries/s", "kind": "port", "cores": 1, "isa": oracle.dot_isa(native=True)}
    t0 = time.perf_counter()
    done = 0
    while True:
        oracle.brute_force(rows_host, queries_host[done % len(queries_host)], k, 0.0, oracle.DOT_NATIVE)
        done += 1
        el = time.perf_counter() - t0
        if el >= seconds and done >= 2:
            break
    out["value"] = round(done / el, 3)
    # multi-thread leg: one worker per PHYSICAL core this process may use (SMT siblings share the core's load ports; the
    # scan is memory-bound), at most 64; worker t is pinned to its core, and the corpus copy it scans was FIRST TOUCHED
    # shard by shard by those same pinned workers, so every worker streams from its own NUMA node.  (Round 3 scanned an
    # array one thread had touched: 64 workers read one node's memory, 83 GB/s on a 256-core host.)
    cores = os.cpu_count() or 1
    cpus = physical_cores()
    threads = max(1, min(len(cpus), 64))
    oracle.set_worker_cpus(cpus[:threads])
    local = oracle.first_touch_copy(rows_host, threads)
    t0 = time.perf_counter()
    done_mt = 0
    while True:
        oracle.brute_force_mt(local, queries_host[done_mt % len(queries_host)], k, 0.0, threads, oracle.DOT_NATIVE)
        done_mt += 1
        el = time.perf_counter() - t0
        if el >= seconds and done_mt >= 2:
            break
    oracle.set_worker_cpus([])
    del local
    out["mt_value"] = round(done_mt / el, 3)
    out["mt_cores"] = threads
    out["mt_gb_per_s"] = round(done_mt * n * rows_host.shape[1] * 4 / el / 1e9, 1)
    out["mt_placement"] = "one pinned worker per physical core, corpus shard first-touched by its own worker"
    out["host_cores"] = cores
    out["host_physical_cores_usable"] = len(cpus)
    out["sample"] = (f"{done} single-thread + {done_mt} {threads}-thread queries, each a full scan of the same "
                     f"{n}x{rows_host.shape[1]} fp32 corpus held in RAM, k={k}, threshold 0.0 (oracle: simsimd "
                     f"dot restated, body that ran: {out['isa']}; + clamp + BoundedScoreHeap)")
    return out


# ---------------------------------------------------------------------------------------------------------
# embedding leg
# ---------------------------------------------------------------------------------------------------------
def seeded_embed_weights(np, cfg):
    """EmbeddingGemma-300m geometry, seeded random weights (no network for the real checkpoint)."""
    rng = np.random.default_rng(0xC950003)
    H, D, I, V, NL = cfg.hidden, cfg.head_dim, cfg.intermediate, cfg.vocab_size, cfg.layers

    def lin(n, k):
        return rng.standard_normal((n, k), dtype=np.float32) * np.float32(1.0 / np.sqrt(k))

    w = {"embed_tokens.weight": rng.standard_normal((V, H), dtype=np.float32) * np.float32(0.05)}
    for l in range(NL):
        p = f"layers.{l}."
        for nme in ("input_layernorm", "post_attention_layernorm", "pre_feedforward_layernorm", "post_feedforward_layernorm"):
            w[p + nme + ".weight"] = rng.standard_normal(H, dtype=np.float32) * np.float32(0.1)
        w[p + "self_attn.q_norm.weight"] = rng.standard_normal(D, dtype=np.float32) * np.float32(0.1)
        w[p + "self_attn.k_norm.weight"] = rng.standard_normal(D, dtype=np.float32) * np.float32(0.1)
        w[p + "self_attn.q_proj.weight"] = lin(cfg.heads * D, H)
        w[p + "self_attn.k_proj.weight"] = lin(cfg.kv_heads * D, H)
        w[p + "self_attn.v_proj.weight"] = lin(cfg.kv_heads * D, H)
        w[p + "self_attn.o_proj.weight"] = lin(H, cfg.heads * D)
        w[p + "mlp.gate_proj.weight"] = lin(I, H)
        w[p + "mlp.up_proj.weight"] = lin(I, H)
        w[p + "mlp.down_proj.weight"] = lin(H, I)
    w["norm.weight"] = rng.standard_normal(H, dtype=np.float32) * np.float32(0.1)
    w["dense1.weight"] = lin(cfg.dense_hidden, H)
    w["dense2.weight"] = lin(H, cfg.dense_hidden)
    return w


def embed_flops(np, cfg, lens):
    """2 x 101.5 M non-embedding parameters per token + attention + the dense head per sequence."""
    H, D, I, NL = cfg.hidden, cfg.head_dim, cfg.intermediate, cfg.layers
    nq = (cfg.heads + 2 * cfg.kv_heads) * D
    gemm = 2.0 * NL * (H * nq + H * cfg.heads * D + H * 2 * I + I * H)
    att = 0.0
    W = cfg.sliding_window // 2 + 1
    n_full = NL // cfg.sliding_pattern
    for L in lens:
        pos = np.arange(L)
        local = np.minimum(pos + W, L) - np.maximum(pos - W + 1, 0)    # keys with |q-k| < W
        att += 4.0 * cfg.heads * D * (n_full * L * L + (NL - n_full) * float(local.sum()))
    head = 2.0 * 2 * H * cfg.dense_hidden * len(lens)
    return gemm * float(np.sum(lens)) + att + head


def embed_cpu_baseline(np, cfg, weights, seconds, L):
    """"PyTorch-CPU Gemma3 (same seeded weights) sequences/sec".  oracle/gemma3_ref.forward (fp32, torch
    CPU, padded batch like ORT) on a bounded sample: a 2-sequence probe sizes one batch of up to 32 x L tokens."""
    import torch
    from oracle import gemma3_ref as G
    gc = G.GemmaConfig(vocab_size=cfg.vocab_size, hidden=cfg.hidden, layers=cfg.layers, heads=cfg.heads, kv_heads=cfg.kv_heads,
                       head_dim=cfg.head_dim, intermediate=cfg.intermediate, sliding_window=cfg.sliding_window,
                       sliding_pattern=cfg.sliding_pattern, dense_hidden=cfg.dense_hidden, max_seq=cfg.max_seq)
    rng = np.random.default_rng(0xC950007)

    def run(B):
        ids = rng.integers(1, cfg.vocab_size, size=(B, L)).astype(np.int64)
        mask = np.ones((B, L), np.int64)
        t0 = time.perf_counter()
        out = G.forward(gc, weights, ids, mask)
        assert np.all(np.isfinite(out))
        return time.perf_counter() - t0

    t2 = run(2)
    B = int(max(2, min(32, (seconds / max(t2 / 2, 1e-6)) // 1)))
    tb = run(B) if B > 2 else t2
    return {"kind": "port", "what": "oracle/gemma3_ref.forward (torch CPU fp32, same seeded weights, padded batch)",
            "chunks_per_sec": round(B / tb, 3), "tokens_per_sec": round(B * L / tb, 1), "cores": torch.get_num_threads(),
            "host_cores": os.cpu_count(), "sample": f"one batch of {B} x {L} tokens ({tb:.1f} s) after a 2-sequence probe",
            "tflops": round(embed_flops(np, gc, [L] * B) / tb / 1e12, 3)}


def embed_leg(a, rank, world, dist, torch, np, dev, all_reduce_max):
    """Index-build leg of the metric ("index embed chunks/sec"): EmbeddingGemma-300m geometry with seeded
    random weights, synthetic token ids, batch = the reference's embed_batch_size() (32), (a) fixed L and
    (b) log-normal lengths ("few hundred tokens").  Data-parallel over ranks: replicated weights,
    no collective."""
    from cqs_amd.embedder import HipEmbedEngine, default_config
    cfg = default_config()
    eng = HipEmbedEngine(cfg, device=dev.index)
    weights = seeded_embed_weights(np, cfg)
    for name, t in weights.items():
        eng.set_tensor(name, t)
    eng.set_weights({})
    rng = np.random.default_rng(0xC950004)
    V = cfg.vocab_size

    def run(lens, steps):
        B, L = len(lens), int(max(lens))
        ids = np.zeros((B, L), np.int64)
        mask = np.zeros((B, L), np.int64)
        for i, n in enumerate(lens):
            ids[i, :n] = rng.integers(1, V, size=n)
            mask[i, :n] = 1
        out = eng.run(ids, mask)                      # warm-up (also sizes the scratch of both execution contexts)
        out = eng.run(ids, mask)
        assert np.all(np.isfinite(out))
        # (1) one `session.run` at a time (the reference's Embedder::embed_batch contract): latency per batch
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dev_ms = 0.0
        n_sync = max(2, steps // 2)
        for _ in range(n_sync):
            eng.run(ids, mask)
            dev_ms += eng.last_ms()
        el_sync = time.perf_counter() - t0
        # (2) tickets in flight (what the index pipeline does, cqs_amd/pipeline.py): submit batch i+2 while i, i+1 run
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pend = []
        for _ in range(steps):
            pend.append(eng.submit(ids, mask))
            if len(pend) == 3:
                eng.collect(pend.pop(0), B)
        for t in pend:
            eng.collect(t, B)
        el = time.perf_counter() - t0
        if dist is not None:
            el = all_reduce_max(el)
        toks = int(np.sum(lens))
        flops = embed_flops(np, cfg, lens)
        tf = flops * steps / el / 1e12               # wall clock, host packing and PCIe included
        return {"chunks_per_sec": round(B * steps * world / el, 1), "tokens_per_sec": round(toks * steps * world / el, 1),
                "ms_per_batch": round(el / steps * 1e3, 3),
                "sync_api": {"ms_per_batch": round(el_sync / n_sync * 1e3, 3), "device_ms_per_batch": round(dev_ms / n_sync, 3),
                             "chunks_per_sec": round(B * n_sync / el_sync, 1)},
                "batch": B, "tokens_per_batch": toks, "gflop_per_batch": round(flops / 1e9, 1),
                "roofline": {"bound": "mfma", "achieved": round(tf, 1), "peak": 2500.0, "unit": "TFLOP/s",
                             "frac": round(tf / 2500.0, 4), "dtype": "bf16",
                             "note": "model flops per batch / wall time per batch, 3 tickets in flight on one engine"}}

    fixed = run([a.embed_len] * a.embed_batch, a.embed_steps)
    lens = np.clip(np.exp(rng.normal(np.log(300.0), 0.6, size=a.embed_batch)).astype(int), 8, cfg.max_seq)
    ragged = run(list(lens), a.embed_steps)
    # SURVEY §8d asks for "a tuned larger batch" beside the reference's 32, and for the measured GEMM ceiling
    # of the bf16 kernel the forward is built on (a big square GEMM through the same kernel)
    big = run([a.embed_len] * (4 * a.embed_batch), max(2, a.embed_steps // 4))
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
    # search-time latency: ONE short sequence through the blocking call = what `embed_query` costs before every search
    # (src/embedder/core.rs:768-856 -> src/cli/commands/search/query.rs:595).  Roofline of this shape = weight streaming:
    # the non-embedding parameters + the Dense head read once (bf16) against the 8 TB/s HBM peak.
    qlat = None
    if rank == 0:
        wbytes = 2.0 * sum(int(np.prod(t.shape)) for n, t in weights.items() if n != "embed_tokens.weight" and t.ndim == 2)
        qlat = {"what": "blocking cqs_hip_embed of ONE sequence, wall clock per call incl. H2D / D2H / sync: ms = through the Python mirror "
                        "(`HipEmbedEngine.run`), abi_ms = the C call alone on prepared buffers, both the MEDIAN of 40 calls (mean_ms / max_ms: the same "
                        "calls through the mirror); device_ms = HIP events around the chain, mean",
                "weight_bytes_streamed": wbytes, "by_tokens": {}}
        for n in (8, 16, 32, 64, 65, 128):
            ids = rng.integers(1, V, size=(1, n)).astype(np.int64)
            mask = np.ones((1, n), np.int64)
            for _ in range(6):
                eng.run(ids, mask)                   # (both contexts: eager run, capture, replays)
            reps = 40
            gs0 = eng.query_graph_stats()
            dms, walls = 0.0, []
            for _ in range(reps):
                t0 = time.perf_counter()
                eng.run(ids, mask)
                walls.append(time.perf_counter() - t0)
                dms += eng.last_ms()
            dt = float(np.median(walls))                 # median: one call in a few hundred stalls for 1-40 ms on the host
            # the C call alone (what the Rust shim pays): prepared buffers, no numpy conversions, no last_ms() in the loop
            import ctypes as C
            out = np.zeros((1, eng.dim()), np.float32)
            args = (eng._h, ids.ctypes.data_as(C.c_void_p), mask.ctypes.data_as(C.c_void_p), 1, n, out.ctypes.data_as(C.c_void_p))
            abi = []
            for _ in range(reps):
                t0 = time.perf_counter()
                rc = eng._lib.cqs_hip_embed(*args)
                abi.append(time.perf_counter() - t0)
            dt_abi = float(np.median(abi))
            assert rc == 0 and np.array_equal(out, eng.run(ids, mask))
            gs1 = eng.query_graph_stats()
            calls = 2 * reps + 1
            replays, eager = gs1["replays"] - gs0["replays"], gs1["eager"] - gs0["eager"]
            path = ("search-time kernels, hipGraph replay" if replays == calls else
                    "search-time kernels, EAGER launches (%d of %d calls)" % (eager, calls) if eager else "batch chain")
            qlat["by_tokens"][str(n)] = {"ms": round(dt * 1e3, 4), "abi_ms": round(dt_abi * 1e3, 4), "device_ms": round(dms / reps, 4),
                                         "mean_ms": round(float(np.mean(walls)) * 1e3, 4), "max_ms": round(float(np.max(walls)) * 1e3, 4),
                                         "path": path,          # observed (cqs_hip_embedder_query_graph_stats), not assumed
                                         "weight_stream_frac_of_hbm_peak": round(wbytes / (dms / reps / 1e3) / 1e9 / HBM_PEAK_GBS, 4)}
        # first-call cost: a length the engine has never seen pays an eager chain, then capture +
        # instantiate; `cqs_hip_embedder_warm` moves that to start-up.  Measured on lengths no call above has used.
        import ctypes as C

        def abi_call(n, seed):
            ids = np.random.default_rng(seed).integers(1, V, size=(1, n)).astype(np.int64)
            mask = np.ones((1, n), np.int64)
            out = np.zeros((1, eng.dim()), np.float32)
            t0 = time.perf_counter()
            rc = eng._lib.cqs_hip_embed(eng._h, ids.ctypes.data_as(C.c_void_p), mask.ctypes.data_as(C.c_void_p), 1, n, out.ctypes.data_as(C.c_void_p))
            dt = time.perf_counter() - t0
            assert rc == 0 and np.all(np.isfinite(out))
            return dt * 1e3

        cold = {}
        for n in (11, 23, 47, 90):
            calls = [abi_call(n, 900 + n + j) for j in range(8)]
            cold[str(n)] = {"first_ms": round(calls[0], 4), "second_ms": round(calls[1], 4), "third_ms": round(calls[2], 4),
                            "steady_ms": round(float(np.median(calls[4:])), 4)}
        t0 = time.perf_counter()
        eng.warm(128)
        warm_s = time.perf_counter() - t0
        gsw = eng.query_graph_stats()
        warmed = {}
        for n in (12, 24, 48, 96):
            calls = [abi_call(n, 950 + n + j) for j in range(12)]
            warmed[str(n)] = {"first_ms": round(calls[0], 4), "steady_ms": round(float(np.median(calls[2:])), 4),
                              "first_over_steady": round(calls[0] / float(np.median(calls[2:])), 3)}
        lens = np.random.default_rng(0xC950041).integers(1, 129, size=240)
        walls = np.array([abi_call(int(n), 1000 + j) for j, n in enumerate(lens)])
        gsr = eng.query_graph_stats()
        qlat["first_call_ms"] = {"no_warm": cold, "after_warm": warmed, "warm_seconds": round(warm_s, 3),
                                 "graphs_after_warm": gsw["captured"], "capture_failures": gsw["failed"],
                                 "what": "abi_ms of the FIRST blocking cqs_hip_embed at a token count the engine has not served: "
                                         "no_warm = cold for that length (eager chain; the second call captures + instantiates); "
                                         "after_warm = after cqs_hip_embedder_warm(128)"}
        qlat["random_lengths"] = {"calls": int(len(walls)), "lengths": "uniform 1..128", "p50_ms": round(float(np.percentile(walls, 50)), 4),
                                  "p99_ms": round(float(np.percentile(walls, 99)), 4), "max_ms": round(float(walls.max()), 4),
                                  "eager_chains_during": gsr["eager"] - gsw["eager"], "captures_during": gsr["captured"] - gsw["captured"]}
    cpu = None          # filled in by main() after the last GPU leg (see the note at the scan's cpu_baseline)
    out = {"model": "EmbeddingGemma-300m geometry (24 x [768 | 3x256 q, 1 kv | 1152], vocab 262144), seeded weights",
           "steps": a.embed_steps, "fixed_len_%d" % a.embed_len: fixed, "lognormal_len": ragged,
           "fixed_len_%d_batch%d" % (a.embed_len, 4 * a.embed_batch): big,
           "gemm_kernel_ceiling_tflops": ceiling, "cpu_baseline": cpu, "query_latency": qlat,
           "note": "host-buffer API (ids in, embeddings out per batch, PCIe-inclusive); timed wall-clock, max over ranks; "
                   "chunks_per_sec = submit/collect with 3 tickets in flight, sync_api = one blocking call per batch"}
    return out, eng, cfg, weights


def e2e_leg(a, torch, np, dev, eng, cfg, weights):
    """BASELINE configs[3]: synthetic code chunks (log-normal token lengths) -> GPU EmbeddingGemma forward through the
    index pipeline (length-sorted batches, cqs_amd.pipeline) -> L2 normalise -> HIP index -> 256 queries k-NN.
    R@5 / R@20: the CPU-oracle pipeline (fp32 forward + oracle scan) on a sub-sample at the real geometry."""
    try:
        from cqs_amd.pipeline import EmbedPipeline
    except Exception as e:  # pipeline module not built yet
        return {"skipped": f"pipeline unavailable: {e}"}
    from cqs_amd import HipIndex
    rng = np.random.default_rng(0xC950008)
    n = a.e2e_chunks
    lens = np.clip(np.exp(rng.normal(np.log(300.0), 0.6, size=n)).astype(int), 8, cfg.max_seq)
    V = cfg.vocab_size
    chunks = [rng.integers(1, V, size=int(L)).astype(np.int64) for L in lens]
    pipe = EmbedPipeline(eng)
    pipe.embed_token_lists(chunks[:256])             # warm-up (scratch sizes)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    emb = pipe.embed_token_lists(chunks)             # [n, 768] f32, L2-normalised, input order
    t_embed = time.perf_counter() - t0
    t1 = time.perf_counter()
    idx = HipIndex.build_from_flat(None, emb[:4096])
    for lo in range(4096, n, 32768):
        idx.extend(None, emb[lo:lo + 32768])
    nq, k = 256, 20
    qrows = rng.choice(n, size=nq, replace=False)
    noise = rng.standard_normal((nq, emb.shape[1])).astype(np.float32) * np.float32(0.02)
    queries = emb[qrows] + noise
    queries /= np.linalg.norm(queries, axis=1, keepdims=True)
    got_rows, got_scores, counts = idx.search_batch(queries.astype(np.float32), k)
    t_index = time.perf_counter() - t1
    total = time.perf_counter() - t0
    assert bool(np.all(counts == k)) and np.all(np.isfinite(got_scores))
    hit1 = float(np.mean(got_rows[:, 0] == qrows))   # a query is its chunk's embedding + 2 % noise
    out = {"workload": f"configs[3]: {n} synthetic chunks (log-normal lengths, median ~300 tokens) -> HIP embed pipeline -> "
                       f"extend -> {nq} queries top-{k}",
           "chunks_per_sec_e2e": round(n / total, 1), "embed_chunks_per_sec": round(n / t_embed, 1),
           "embed_tokens_per_sec": round(float(lens.sum()) / t_embed, 1), "embed_s": round(t_embed, 2),
           "index_and_query_s": round(t_index, 3), "self_hit_at_1": round(hit1, 4), "pipeline": pipe.stats()}
    idx.close()
    # recall against the CPU-oracle pipeline on a sub-sample (the fp32 CPU forward is ~1e4 x slower)
    if a.cpu_seconds > 0:
        from oracle import gemma3_ref as G
        from oracle import oracle
        gc = G.GemmaConfig(vocab_size=cfg.vocab_size, hidden=cfg.hidden, layers=cfg.layers, heads=cfg.heads,
                           kv_heads=cfg.kv_heads, head_dim=cfg.head_dim, intermediate=cfg.intermediate,
                           sliding_window=cfg.sliding_window, sliding_pattern=cfg.sliding_pattern,
                           dense_hidden=cfg.dense_hidden, max_seq=cfg.max_seq)
        budget = max(20.0, 6 * a.cpu_seconds)
        order = np.argsort(lens)[: max(64, n // 4)]           # short chunks first: most chunks per CPU-second
        sub, t_cpu0, ref = [], time.perf_counter(), []
        for lo in range(0, len(order), 16):
            sel = order[lo:lo + 16]
            L = int(max(lens[sel]))
            ids = np.zeros((len(sel), L), np.int64)
            mask = np.zeros((len(sel), L), np.int64)
            for i, c in enumerate(sel):
                ids[i, :lens[c]] = chunks[c]
                mask[i, :lens[c]] = 1
            ref.append(G.forward(gc, weights, ids, mask))
            sub.extend(int(c) for c in sel)
            if time.perf_counter() - t_cpu0 > budget and len(sub) >= 64:
                break
        ref = np.concatenate(ref)
        ref /= np.linalg.norm(ref, axis=1, keepdims=True)
        ref = ref.astype(np.float32)
        sub = np.array(sub)
        hip_sub = np.ascontiguousarray(emb[sub])
        cs = np.sum(hip_sub * ref, axis=1)
        nqs = min(64, len(sub) // 2)
        sidx = HipIndex.build_from_flat(None, hip_sub)
        r5 = r20 = 0.0
        kk = min(20, len(sub) - 1)
        gr, _, gc_ = sidx.search_batch(hip_sub[:nqs], kk)
        for qi in range(nqs):
            ref_ids, _ = oracle.index_search(ref, ref[qi], kk)
            got = [int(x) for x in gr[qi, :int(gc_[qi])]]
            r5 += len(set(ref_ids[:5]) & set(got[:5])) / 5.0
            r20 += len(set(ref_ids[:kk]) & set(got[:kk])) / float(kk)
        sidx.close()
        out["recall_vs_cpu_oracle"] = {"sample_chunks": int(len(sub)), "queries": int(nqs), "R@5": round(r5 / nqs, 4),
                                       "R@20": round(r20 / nqs, 4), "min_cosine_hip_vs_fp32": round(float(cs.min()), 6),
                                       "cpu_forward_s": round(time.perf_counter() - t_cpu0, 1),
                                       "note": "seeded weights: parity unpinned w.r.t. the real checkpoint (no weights offline)"}
    return out


# ---------------------------------------------------------------------------------------------------------
# scan legs
# ---------------------------------------------------------------------------------------------------------
def other_configs(torch, np, HipIndex, idx, rows, queries, dim, dev, st):
    """The other BASELINE / SURVEY §8d configurations, timed the same way (inputs resident in HBM, device API,
    steps enqueued back to back) and CHECKED outside the timed region (check_topk).  Reported beside the headline,
    never instead of it."""
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
        return (time.perf_counter() - t0) / steps, keys.cpu().numpy().view(np.uint64), cnt.cpu().numpy()

    out = {}
    n = rows.shape[0]
    q1 = queries[0, 0].contiguous()
    t, hk, hc = timed(idx, q1, 1, 500, 100, 10)               # what production asks for (src/limits.rs:315-320)
    check_topk(torch, np, rows, q1, hk[0], hc[0], 500, what="k500_1M")
    out["k500_1M"] = {"queries_per_sec": round(1.0 / t, 1), "ms_per_query": round(t * 1e3, 4), "checked": True}
    qb = make_unit_rows(torch, 256, dim, 0xC950003, dev)
    t, hk, hc = timed(idx, qb, 256, 20, 60, 15)               # configs[2]: 256-query blocks on the f32 matrix cores
    allsc = rows @ qb.T                                       # exhaustive threshold count for all 256 queries at once
    kth = torch.empty((256,), device=dev)
    for qi in range(256):
        _, s = check_topk(torch, np, rows, qb[qi], hk[qi], hc[qi], 20, exhaustive=False, what="batch256_1M[%d]" % qi)
        kth[qi] = float(s[-1])
    beat = (allsc > (kth + 2e-6)[None, :]).sum(dim=0)
    assert int(beat.max().item()) <= 19, "batch256_1M: rows beat the k-th score"
    del allsc
    tf = 2.0 * 256 * n * dim / t / 1e12
    out["batch256_1M"] = {"queries_per_sec": round(256 / t, 1), "ms_per_batch": round(t * 1e3, 3), "checked": True,
                          "roofline": {"bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                                       "frac": round(tf / MFMA_F32_PEAK_TF, 4), "dtype": "f32"}}
    for bsmall in (64, 32):                                     # smaller blocks on the matrix cores
        qs = qb[:bsmall].contiguous()
        t, hk, hc = timed(idx, qs, bsmall, 20, 100, 15)
        for qi in (0, bsmall // 2, bsmall - 1):
            check_topk(torch, np, rows, qs[qi], hk[qi], hc[qi], 20, what="batch%d_1M[%d]" % (bsmall, qi))
        tf = 2.0 * bsmall * n * dim / t / 1e12
        gbs = n * dim * 4 / t / 1e9
        out["batch%d_1M" % bsmall] = {"queries_per_sec": round(bsmall / t, 1), "ms_per_batch": round(t * 1e3, 3), "checked": True,
                                       "roofline": {"bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                                                    "frac": round(tf / MFMA_F32_PEAK_TF, 4), "dtype": "f32",
                                                    "hbm_frac": round(gbs / HBM_PEAK_GBS, 4),
                                                    "note": "B / 2 flop per corpus byte: at 32 queries the block sits below the ridge (~23 flop/B) and "
                                                            "is HBM-bound (hbm_frac), at 64 just above it"}}
    small = make_unit_rows(torch, 17523, dim, 0xC950004, dev)  # configs[0] shape (cache resident: not judged against HBM)
    si = HipIndex.build_from_device(None, small.data_ptr(), 17523, dim, device=dev.index or 0, borrow=True, keepalive=small)
    t, hk, hc = timed(si, q1, 1, 20, 500, 50)
    check_topk(torch, np, small, q1, hk[0], hc[0], 20, what="rows17523")
    out["rows17523"] = {"queries_per_sec": round(1.0 / t, 1), "ms_per_query": round(t * 1e3, 4), "checked": True}
    si.close()
    del small
    try:
        big_n = 10_000_000
        big = make_unit_rows(torch, big_n, dim, 0xC950005, dev)
        bi = HipIndex.build_from_device(None, big.data_ptr(), big_n, dim, device=dev.index or 0, borrow=True, keepalive=big)
        t, hk, hc = timed(bi, q1, 1, 20, 20, 3)
        check_topk(torch, np, big, q1, hk[0], hc[0], 20, what="rows10M")
        gbs = big_n * dim * 4 / t / 1e9
        out["rows10M"] = {"queries_per_sec": round(1.0 / t, 2), "ms_per_query": round(t * 1e3, 3), "checked": True,
                          "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": round(gbs / HBM_PEAK_GBS, 4), "note": "whole step incl. select"}}
        bi.close()
        del big
    except AssertionError:
        raise
    except Exception as e:  # e.g. not enough free HBM beside another tenant
        out["rows10M"] = {"skipped": str(e)[:120]}
    torch.cuda.empty_cache()
    return out


def aux_models_leg(a, np):
    """the two BERT-family auxiliary models at their real geometry (seeded weights, synthetic token
    ids): SPLADE encode (BERT-base masked-LM + pooling -> sparse vectors) and reranker scoring (MiniLM-L6
    cross-encoder).  Each is checked against the fp32 CPU oracle on a small batch outside the timed region; the CPU
    baseline is that oracle (torch CPU) on a bounded sample."""
    from oracle import bert_ref as R
    from cqs_amd import _lib
    from cqs_amd.splade import HipBertEngine, Reranker, SpladeEncoder, bert_config
    rng = np.random.default_rng(0xC950009)
    out = {}

    def padded(seqs, types=None):
        L = max(len(s) for s in seqs)
        ids = np.zeros((len(seqs), L), np.int64); mask = np.zeros((len(seqs), L), np.int64); tt = np.zeros((len(seqs), L), np.int64)
        for i, s in enumerate(seqs):
            ids[i, :len(s)] = s; mask[i, :len(s)] = 1
            if types is not None:
                tt[i, :len(s)] = types[i]
        return ids, mask, tt

    # SPLADE
    cfg = R.splade_base()
    w = R.seeded_weights(cfg, "mlm", seed=1)
    eng = HipBertEngine(bert_config(_lib.BERT_HEAD_MLM))
    eng.set_weights(w)
    small = [rng.integers(1, cfg.vocab_size, size=n).astype(np.int32) for n in (48, 200, 7)]
    got = eng.splade_dense(small)
    t0 = time.perf_counter()
    _, want = R.splade_encode_batch(cfg, w, *padded(small)[:2], 0.01)
    cpu_s = time.perf_counter() - t0
    err = float(np.max(np.abs(got - want)))
    assert err < 0.08, "splade activations differ from the fp32 oracle: %g" % err
    B, L = 64, 256
    seqs = [rng.integers(1, cfg.vocab_size, size=L).astype(np.int32) for _ in range(B)]
    # seeded weights make half the vocabulary "active"; a trained SPLADE keeps 100-300 entries per document
    # (src/splade/mod.rs:44): put the threshold where ~200 survive so the host-side filter does realistic work
    warm = eng.splade_dense(seqs)
    thr = float(np.sort(warm[0])[-200])
    enc = SpladeEncoder(eng, threshold=thr)
    t_end = time.perf_counter() + 0.5                                  # (let the oracle's CPU threads stop spinning)
    while time.perf_counter() < t_end:
        enc.encode_batch_arrays(seqs)
    steps = max(4, a.embed_steps)
    t0 = time.perf_counter()
    for _ in range(steps):
        sv = enc.encode_batch_arrays(seqs)
    dt_sync = (time.perf_counter() - t0) / steps                       # one blocking encode_batch at a time
    enc.encode_batches_arrays([seqs] * 3)
    t0 = time.perf_counter()
    piped = enc.encode_batches_arrays([seqs] * (2 * steps))             # the index pipeline's form: 3 tickets in flight
    dt = (time.perf_counter() - t0) / (2 * steps)
    assert all(np.array_equal(piped[0][b][0], sv[b][0]) and np.array_equal(piped[-1][b][1], sv[b][1]) for b in range(B)), "tickets != blocking call"
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.splade_dense(seqs)
    dt_dense = (time.perf_counter() - t0) / steps
    flops = 2.0 * B * L * (cfg.layers * (4 * cfg.hidden * cfg.hidden + 2 * cfg.hidden * cfg.intermediate) + cfg.hidden * cfg.hidden
                           + cfg.hidden * cfg.vocab_size) + 4.0 * B * cfg.layers * L * L * cfg.hidden
    out["splade"] = {"model": "BERT-base masked-LM geometry (12 x [768 | 12 x 64 | 3072], vocab 30522), seeded weights",
                     "batch": B, "tokens_per_doc": L, "docs_per_sec": round(B / dt, 1), "tokens_per_sec": round(B * L / dt, 1),
                     "ms_per_batch": round(dt * 1e3, 3), "sync_api": {"docs_per_sec": round(B / dt_sync, 1), "ms_per_batch": round(dt_sync * 1e3, 3)},
                     "ms_per_batch_device_side": round(dt_dense * 1e3, 3),
                     "tflops": round(flops / dt_dense / 1e12, 1), "nnz_per_doc": round(float(np.mean([len(v[0]) for v in sv])), 1),
                     "checked": {"max_abs_err_vs_fp32_oracle": round(err, 4)},
                     "cpu_baseline": {"docs_per_sec": round(3 / cpu_s, 2), "kind": "port", "sample": "oracle/bert_ref (torch CPU fp32), 3 docs / 255 tokens"},
                     "threshold": round(thr, 4),
                     "note": "host API: token ids in, sparse vectors out (threshold filter of src/splade/mod.rs:1049-1062 on the device), 3 tickets in flight "
                             "(cqs_hip_splade_submit_sparse / _collect_sparse; sync_api = one blocking cqs_hip_splade_encode_sparse per batch); "
                             "threshold set where ~200 entries per document survive (seeded weights are not sparse)"}
    eng.close()

    # BERT-family embedder presets of the `Embedder` seam (bge-large = the reference's strongest: src/embedder/models.rs:374-405)
    for name, cfg, B, L in (("bge_large", R.bge_large(), 32, 512), ("e5_base", R.e5_base(), 32, 512)):
        w = R.seeded_weights(cfg, "none", seed=3)
        eng = HipBertEngine(bert_config(_lib.BERT_HEAD_NONE, hidden=cfg.hidden, layers=cfg.layers, heads=cfg.heads,
                                        intermediate=cfg.intermediate))
        eng.set_weights(w)
        small = [rng.integers(1, cfg.vocab_size, size=n).astype(np.int32) for n in (40, 130)]
        got = eng.embed(small, None, "mean")
        t0 = time.perf_counter()
        want = R.pooled_embedding(cfg, w, *padded(small)[:2])
        cpu_s = time.perf_counter() - t0
        cs = min(float(np.dot(got[i], want[i]) / (np.linalg.norm(got[i]) * np.linalg.norm(want[i]))) for i in range(2))
        assert cs > 0.999, "%s embeddings differ from the fp32 oracle: cos %g" % (name, cs)
        seqs = [rng.integers(1, cfg.vocab_size, size=L).astype(np.int32) for _ in range(B)]
        t_end = time.perf_counter() + 0.3
        while time.perf_counter() < t_end:
            eng.embed(seqs, None, "mean")
        steps = max(4, a.embed_steps)
        t0 = time.perf_counter()
        for _ in range(steps):
            ref_out = eng.embed(seqs, None, "mean")
        dt_sync = (time.perf_counter() - t0) / steps
        pend, last = [], None
        t0 = time.perf_counter()
        for _ in range(2 * steps):                                        # tickets: 3 in flight
            pend.append(eng.embed_submit(seqs, None, "mean"))
            if len(pend) == 3:
                last = eng.embed_collect(pend.pop(0))
        for h in pend:
            last = eng.embed_collect(h)
        dt = (time.perf_counter() - t0) / (2 * steps)
        assert np.array_equal(last, ref_out), "tickets != blocking call"
        flops = 2.0 * B * L * cfg.layers * (4 * cfg.hidden * cfg.hidden + 2 * cfg.hidden * cfg.intermediate) + 4.0 * B * cfg.layers * L * L * cfg.hidden
        out["embedder_" + name] = {"model": "%s geometry (%d x [%d | %d x 64 | %d]), seeded weights, mean pooling" % (
                                       name.replace("_", "-"), cfg.layers, cfg.hidden, cfg.heads, cfg.intermediate),
                                   "batch": B, "tokens_per_chunk": L, "chunks_per_sec": round(B / dt, 1), "tokens_per_sec": round(B * L / dt, 1),
                                   "ms_per_batch": round(dt * 1e3, 3), "tflops": round(flops / dt / 1e12, 1),
                                   "sync_api": {"chunks_per_sec": round(B / dt_sync, 1), "ms_per_batch": round(dt_sync * 1e3, 3)},
                                   "note": "3 tickets in flight (cqs_hip_bert_embed_submit / _collect); sync_api = one blocking cqs_hip_bert_embed per batch",
                                   "checked": {"min_cosine_vs_fp32_oracle": round(cs, 6)},
                                   "cpu_baseline": {"chunks_per_sec": round(2 / cpu_s, 2), "kind": "port", "sample": "oracle/bert_ref (torch CPU fp32), 2 chunks / 170 tokens"}}
        eng.close()
        del w

    # reranker
    cfg = R.minilm_l6()
    w = R.seeded_weights(cfg, "classifier", seed=2)
    eng = HipBertEngine(bert_config(_lib.BERT_HEAD_CLASSIFIER))
    eng.set_weights(w)
    rr = Reranker(eng)
    small = [rng.integers(1, cfg.vocab_size, size=n).astype(np.int32) for n in (64, 300, 20, 128)]
    st = [np.r_[np.zeros(12, np.int32), np.ones(len(s) - 12, np.int32)] for s in small]
    got = rr.scores(small, st)
    t0 = time.perf_counter()
    want = R.rerank_scores(cfg, w, *padded(small, st))
    cpu_s = time.perf_counter() - t0
    err = float(np.max(np.abs(got - want)))
    assert err < 0.02, "reranker scores differ from the fp32 oracle: %g" % err
    B, L = 32, 512                                                     # the reference's batch (src/reranker.rs:83)
    seqs = [rng.integers(1, cfg.vocab_size, size=L).astype(np.int32) for _ in range(B)]
    tts = [np.r_[np.zeros(16, np.int32), np.ones(L - 16, np.int32)] for _ in range(B)]
    t_end = time.perf_counter() + 0.5                                  # (let the oracle's CPU threads stop spinning)
    while time.perf_counter() < t_end:
        rr.scores(seqs, tts)
    steps = max(20, 4 * a.embed_steps)
    t0 = time.perf_counter()
    for _ in range(steps):
        rr.scores(seqs, tts)
    dt = (time.perf_counter() - t0) / steps
    out["reranker"] = {"model": "MiniLM-L6-H384 cross-encoder geometry (6 x [384 | 12 x 32 | 1536]), seeded weights",
                       "batch": B, "tokens_per_pair": L, "pairs_per_sec": round(B / dt, 1), "ms_per_batch": round(dt * 1e3, 3),
                       "checked": {"max_abs_err_vs_fp32_oracle": round(err, 4)},
                       "cpu_baseline": {"pairs_per_sec": round(4 / cpu_s, 2), "kind": "port", "sample": "oracle/bert_ref (torch CPU fp32), 4 pairs / 512 tokens"}}
    eng.close()
    return out


def sparse_index_leg(a, np, dense_idx=None, dense_queries=None):
    """The SPLADE retrieval leg (`SpladeIndex::search_with_filter`, src/splade/index.rs:223-290) behind the C ABI:
    1M synthetic chunk vectors (~96 distinct tokens each, skewed token frequencies), 64-term queries, k = 500
    (candidate_count_for(limit), src/limits.rs:315-320) through the blocking host API; every timed answer's chunk order
    and score BITS checked against the oracle, which is also the CPU baseline."""
    from cqs_amd import synth
    from cqs_amd.splade_index import HipSpladeIndex
    from oracle import oracle as O
    n, vocab, k = a.sparse_chunks, 30522, 500
    t0 = time.perf_counter()
    off, tok, w = synth.sparse_corpus(n, vocab)
    t_gen = time.perf_counter() - t0
    t0 = time.perf_counter()
    h = HipSpladeIndex.build_from_csr(None, off, tok, w)
    t_build = time.perf_counter() - t0
    out = {"chunks": n, "postings": h.postings(), "unique_tokens": h.unique_tokens(), "k": k, "build_s": round(t_build, 2),
           "what": "cqs_hip_sparse_index_search, host query terms in / host (chunk, score) out, one call at a time; "
                   "accumulate = HIP events around the scoring launch (the exact select and the copies are the rest)"}
    ora = O.SpladeIndex(off, tok, w)
    for terms in (64, 200):
        qs = synth.sparse_queries(40, terms, vocab, seed=0x5BA2DF + terms)
        for qt, qw in qs[:5]:
            h.search_raw(qt, qw, k)
        res, acc, touched = [], [], []
        t0 = time.perf_counter()
        for qt, qw in qs:
            res.append(h.search_raw(qt, qw, k))
            ms, tp = h.last_search()
            acc.append(ms)
            touched.append(tp)
        el = time.perf_counter() - t0
        t0 = time.perf_counter()
        ncpu = 0
        for (qt, qw), (hc, hs, rc) in zip(qs, res):
            oc, os_ = ora.search_raw(qt, qw, k)
            ncpu += 1
            assert rc == 0 and np.array_equal(hc, oc) and np.array_equal(hs.view(np.uint32), os_.view(np.uint32)), "sparse leg differs from the oracle"
            if time.perf_counter() - t0 > max(2.0, a.cpu_seconds / 2):
                break
        cpu_el = time.perf_counter() - t0
        acc = np.asarray(acc, dtype=np.float64) * 1e-3
        alg = np.asarray(touched, dtype=np.float64) * 8.0 + n * 4.0 + (n / 64.0) * 4.0
        gbs = float(np.mean(alg / acc)) / 1e9
        out["terms%d" % terms] = {
            "queries_per_sec": round(len(qs) / el, 1), "ms_per_query": round(el / len(qs) * 1e3, 4),
            "accumulate_ms": round(float(np.mean(acc)) * 1e3, 4), "touched_postings": int(np.mean(touched)),
            "checked": ncpu, "checked_bit_exact": True,
            "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4),
                         "alg_bytes": int(np.mean(alg)),
                         "note": "algorithmic bytes = 8 B per touched posting + the 4 B/chunk score row and its maxima; the query touches "
                                 "~1.5-3.5 % of the index: the launch is short (15-40 us) and latency- rather than bandwidth-shaped"},
            "cpu_baseline": {"kind": "port", "cores": 1, "queries_per_sec": round(ncpu / cpu_el, 2), "ms_per_query": round(cpu_el / ncpu * 1e3, 3),
                             "sample": "%d of the timed queries through oracle.SpladeIndex.search_raw (dense score array in place of the HashMap)" % ncpu},
        }
    # persistence: what a daemon restart costs instead of the rebuild (own format, cqs_hip_sparse_index_save / _load)
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        pth = os.path.join(td, "splade.hip.bin")
        t0 = time.perf_counter()
        h.save(pth, 1)
        t_save = time.perf_counter() - t0
        t0 = time.perf_counter()
        h2 = HipSpladeIndex.load(pth, 1)
        t_load = time.perf_counter() - t0
        qt, qw = synth.sparse_queries(1, 64, vocab, seed=0x5BA2E9)[0]
        a1, a2 = h.search_raw(qt, qw, k), h2.search_raw(qt, qw, k)
        assert np.array_equal(a1[0], a2[0]) and np.array_equal(a1[1].view(np.uint32), a2[1].view(np.uint32))
        out["persist"] = {"file_mb": round(os.path.getsize(pth) / 1e6, 1), "save_s": round(t_save, 2), "load_s": round(t_load, 2),
                          "build_s": round(t_build, 2), "checked_bit_exact": True}
        h2.close()
    # several queries per call (cqs_hip_sparse_index_search_batch): evaluation runs, or a caller that gathers its clients
    qs = synth.sparse_queries(64, 64, vocab, seed=0x5BA2E3)
    out["batched_64_terms"] = {}
    for bsz in (8, 32):
        groups = [qs[i:i + bsz] for i in range(0, 64, bsz)]
        h.search_batch_raw(groups[0], k)
        t0 = time.perf_counter()
        res = [h.search_batch_raw(g, k) for g in groups]
        el = time.perf_counter() - t0
        ch, scs, cnt, rc = res[0]
        oc, os_ = ora.search_raw(groups[0][3][0], groups[0][3][1], k)
        assert rc == 0 and np.array_equal(ch[3, :cnt[3]], oc) and np.array_equal(scs[3, :cnt[3]].view(np.uint32), os_.view(np.uint32))
        out["batched_64_terms"][str(bsz)] = {"queries_per_sec": round(64 / el, 1), "ms_per_call": round(el / len(groups) * 1e3, 4),
                                             "checked_bit_exact": True}
    # concurrent callers of the single-query entry point (the daemon's threads): combined into shared batches
    import threading
    qs = synth.sparse_queries(48, 64, vocab, seed=0x5BA2E5)
    lone = [h.search_raw(qt, qw, k) for qt, qw in qs]
    out["concurrent_clients"] = {"what": "N Python threads, each one blocking cqs_hip_sparse_index_search at a time (ctypes releases the "
                                         "interpreter lock inside the call); every answer bit-identical to the lone call's (checked)"}
    for nthreads in (1, 8):
        p0, q0 = h.combine_stats()
        ok = [True] * nthreads

        def work(t):
            for rep in range(25):
                i = (t * 5 + rep) % len(qs)
                c, sc_, rc = h.search_raw(qs[i][0], qs[i][1], k)
                ok[t] &= rc == 0 and np.array_equal(c, lone[i][0]) and np.array_equal(sc_.view(np.uint32), lone[i][1].view(np.uint32))

        th = [threading.Thread(target=work, args=(t,)) for t in range(nthreads)]
        t0 = time.perf_counter()
        [t.start() for t in th]; [t.join() for t in th]
        el = time.perf_counter() - t0
        p1, q1 = h.combine_stats()
        assert all(ok), "a combined sparse search differs from the lone call"
        out["concurrent_clients"][str(nthreads)] = {"queries_per_sec": round(25 * nthreads / el, 1), "ms_per_call": round(el / 25 * 1e3, 4),
                                                    "mean_callers_per_pass": round((q1 - q0) / max(1, p1 - p0), 2), "checked": True}
    # the same from NATIVE threads (the interpreter lock out of the way): cqs_hip_debug_sparse_client_storm
    import ctypes as C
    from cqs_amd import _lib
    storm = _lib.load().cqs_hip_debug_sparse_client_storm
    storm.restype = C.c_double
    storm.argtypes = [C.c_void_p] * 4 + [C.c_uint32] * 4 + [C.c_void_p] * 3
    nq = len(qs)
    q_off = np.zeros(nq + 1, np.uint64)
    for i, (qt, _qw) in enumerate(qs):
        q_off[i + 1] = q_off[i] + qt.size
    qt_all = np.concatenate([qt for qt, _ in qs]).astype(np.uint32)
    qw_all = np.concatenate([qw for _, qw in qs]).astype(np.float32)
    out["concurrent_clients"]["native_threads"] = {}
    for nthreads in (1, 2, 4, 8, 16):
        oc = np.zeros((nq, k), np.uint64); osc = np.zeros((nq, k), np.float32); ocn = np.zeros(nq, np.uint32)
        p0, q0 = h.combine_stats()
        per = 120
        el = storm(h._h, q_off.ctypes.data, qt_all.ctypes.data, qw_all.ctypes.data, nq, k, nthreads, per, oc.ctypes.data, osc.ctypes.data,
                   ocn.ctypes.data)
        p1, q1 = h.combine_stats()
        assert el > 0
        for i in range(nq):
            if ocn[i]:
                assert np.array_equal(oc[i, :ocn[i]], lone[i][0]) and np.array_equal(osc[i, :ocn[i]].view(np.uint32), lone[i][1].view(np.uint32))
        out["concurrent_clients"]["native_threads"][str(nthreads)] = {
            "queries_per_sec": round(nthreads * per / el, 1), "ms_per_call": round(el / per * 1e3, 4),
            "mean_callers_per_pass": round((q1 - q0) / max(1, p1 - p0), 2), "checked": True}
    out["corpus_gen_s"] = round(t_gen, 1)
    if dense_idx is not None and len(dense_idx) == n:
        # Both retrieval legs of `search_hybrid_inner` (src/search/query.rs:879-901) for one query at k = candidate_count = 500
        # on the same 1M chunks: the dense scan and the sparse index, one after the other and from two threads (the handles
        # are independent: different streams, different mutexes); then the fusion mirror (in cqs it stays in Rust).
        import threading
        from cqs_amd.index import IndexResult
        from cqs_amd.splade_index import fuse_hybrid
        qs = synth.sparse_queries(40, 64, vocab, seed=0x5BA2E1)
        dq = dense_queries[:40]
        for i in range(5):
            dense_idx.search_batch(dq[i], k); h.search_raw(qs[i][0], qs[i][1], k)
        t0 = time.perf_counter()
        for i in range(40):
            dense_idx.search_batch(dq[i], k)
        t_d = (time.perf_counter() - t0) / 40
        t0 = time.perf_counter()
        for i in range(40):
            h.search_raw(qs[i][0], qs[i][1], k)
        t_s = (time.perf_counter() - t0) / 40
        res = [None, None]

        def dense_side():
            res[0] = [dense_idx.search_batch(dq[i], k) for i in range(40)]

        def sparse_side():
            res[1] = [h.search_raw(qs[i][0], qs[i][1], k) for i in range(40)]

        t0 = time.perf_counter()
        th = [threading.Thread(target=dense_side), threading.Thread(target=sparse_side)]
        [t.start() for t in th]; [t.join() for t in th]
        t_both = (time.perf_counter() - t0) / 40
        rows_d, sc_d, cnt_d = res[0][0]
        hc, hs, _rc = res[1][0]
        d = [IndexResult(str(int(r)), float(x)) for r, x in zip(rows_d[0, :cnt_d[0]], sc_d[0, :cnt_d[0]])]
        sres = [IndexResult(str(int(c)), float(x)) for c, x in zip(hc, hs)]
        t0 = time.perf_counter()
        fused = fuse_hybrid(d, sres, 0.7, k)
        t_f = time.perf_counter() - t0
        out["hybrid"] = {"k": k, "dense_leg_ms": round(t_d * 1e3, 4), "sparse_leg_ms": round(t_s * 1e3, 4),
                         "one_after_the_other_ms": round((t_d + t_s) * 1e3, 4), "two_threads_ms_per_query": round(t_both * 1e3, 4),
                         "fused_candidates": len(fused), "fusion_python_mirror_ms": round(t_f * 1e3, 3),
                         "what": "search_hybrid_inner's two retrieval legs for one query, k = 500 each, 1M chunks, blocking host APIs; "
                                 "two_threads = 40 dense and 40 sparse searches issued from one thread each, wall time / 40; the fusion "
                                 "(query.rs:909-1010) stays in Rust in cqs - the Python mirror's time is listed for completeness"}
    h.close()
    return out


def concurrent_clients_leg(np, idx, qh, k, dim):
    """What N daemon client threads see (src/cli/watch/daemon.rs:273: one thread per client, all calling `search` on one
    Arc<dyn VectorIndex>): N threads, each one blocking `cqs_hip_index_search` call at a time, one query per call, on the
    headline corpus.  `native`: the threads are std::threads inside the library calling the public entry point (a Rust
    daemon has no interpreter lock); `python`: Python threads through ctypes (GIL released during the call, taken between
    calls).  Every answer is compared bit for bit with the same query asked alone."""
    import ctypes as C
    import threading
    lib = idx._lib
    storm = lib.cqs_hip_debug_client_storm
    storm.restype = C.c_double
    storm.argtypes = [C.c_void_p, C.c_void_p] + [C.c_uint32] * 5 + [C.c_void_p] * 3
    nq = 48                                               # a multiple of every thread count below
    q = np.ascontiguousarray(qh[:nq], dtype=np.float32)
    assert q.shape[0] == nq
    want = [idx.search_batch(q[i], k) for i in range(nq)]
    want_r = np.stack([w[0][0] for w in want])
    want_s = np.stack([w[1][0] for w in want])
    out = {"what": "N threads, each one blocking cqs_hip_index_search(b = 1) at a time on the headline corpus; queries/s over all "
                   "threads; every answer bit-identical to the lone call's (checked)", "k": k, "native_threads": {}, "python_threads": {}}
    for T in (1, 2, 4, 8, 16):
        per = max(60, 1920 // T)
        rows = np.zeros((nq, k), np.uint64)
        scores = np.zeros((nq, k), np.float32)
        counts = np.zeros((nq,), np.uint32)
        storm(idx._h, q.ctypes.data, nq, dim, k, T, 24, rows.ctypes.data, scores.ctypes.data, counts.ctypes.data)   # warm
        p0, q0 = idx.combine_stats()
        el = storm(idx._h, q.ctypes.data, nq, dim, k, T, per, rows.ctypes.data, scores.ctypes.data, counts.ctypes.data)
        p1, q1 = idx.combine_stats()
        assert el > 0, "a client call failed"
        assert np.all(counts == k) and np.array_equal(rows, want_r) and np.array_equal(scores, want_s), "combined answers differ from the lone call's"
        out["native_threads"][str(T)] = {"queries_per_sec": round(T * per / el, 1), "ms_per_call": round(el / per * 1e3, 4),
                                          "mean_callers_per_pass": round((q1 - q0) / max(p1 - p0, 1), 2), "checked": True}
    for T in (1, 8):
        per = max(60, 960 // T)
        bufs = [(np.zeros((1, k), np.uint64), np.zeros((1, k), np.float32), np.zeros((1,), np.uint32)) for _ in range(T)]
        bad = []

        def work(t):
            r, s_, c = bufs[t]
            qi = t % nq
            for _ in range(per):
                rc = lib.cqs_hip_index_search(idx._h, q[qi].ctypes.data, 1, dim, k, None, 0, 0.0, r.ctypes.data, s_.ctypes.data, c.ctypes.data)
                if rc != 0 or not (np.array_equal(r[0], want_r[qi]) and np.array_equal(s_[0], want_s[qi])):
                    bad.append((t, qi, rc))
                    return
                qi = (qi + T) % nq

        th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
        t0 = time.perf_counter()
        [x.start() for x in th]
        [x.join() for x in th]
        el = time.perf_counter() - t0
        assert not bad, bad
        out["python_threads"][str(T)] = {"queries_per_sec": round(T * per / el, 1), "ms_per_call": round(el / per * 1e3, 4), "checked": True}
    return out


def abi_sharded_leg(a, torch, np, rows, queries, k, dim):
    """Single-process multi-GPU path behind the C ABI (cqs_hip_index_create_sharded): synchronous host-API queries
    on the same corpus, checked against the single-device answer."""
    from cqs_amd import HipIndex
    devs = [int(x) for x in a.abi_devices.split(",") if x != ""]
    host = rows.cpu().numpy()
    t0 = time.perf_counter()
    sh = HipIndex.build_sharded(None, host, devs)
    t_build = time.perf_counter() - t0
    qh = queries[:, 0].cpu().numpy()
    nq = min(200, qh.shape[0])
    for i in range(min(10, nq)):
        sh.search_batch(qh[i], k)
    t0 = time.perf_counter()
    res = [sh.search_batch(qh[i], k) for i in range(nq)]
    el = time.perf_counter() - t0
    single = HipIndex.build_from_device(None, rows.data_ptr(), rows.shape[0], dim, borrow=True, keepalive=rows)
    for i in range(min(10, nq)):
        single.search_batch(qh[i], k)
    t0 = time.perf_counter()
    ref = [single.search_batch(qh[i], k) for i in range(nq)]
    el1 = time.perf_counter() - t0
    for i in range(nq):                                 # EVERY timed query against the single-device answer
        r1, s1, c1 = ref[i]
        assert c1[0] == res[i][2][0] and np.max(np.abs(s1 - res[i][1])) <= 2e-6
        if np.all(np.abs(np.diff(s1[0])) > 4e-6):
            assert np.array_equal(r1, res[i][0])
    info = sh.shards()
    single.close()
    sh.close()
    return {"devices": devs, "shards": [{"device": d, "rows": r, "rccl": rc} for d, _f, r, rc in info],
            "queries_per_sec_host_api": round(nq / el, 1), "ms_per_query": round(el / nq * 1e3, 4),
            "single_device_same_queries": {"queries_per_sec_host_api": round(nq / el1, 1), "ms_per_query": round(el1 / nq * 1e3, 4)},
            "vs_single_device": round(el1 / el, 4), "queries": nq,
            "build_s": round(t_build, 2), "checked_vs_single_device": True, "checked": True,
            "what": "cqs_hip_index_create_sharded -> per-shard scan + select -> gather -> host merge, blocking host API, "
                    "one query per call; a device named more than once gathers without RCCL (one-GPU form)"}


def abi_after_group_leg(a, torch, np, world, k, dim, rows_per_device=250_000, budget_s=150.0):
    """Rank 0, after the process group is gone: cqs_hip_index_create_sharded over devices 0..world-1 (RCCL clique inside
    the library), checked against the single-device answer, timed through the blocking host API.  Runs in a thread
    with a wall-clock budget; any failure becomes an `error` field, never a lost bench line."""
    import threading
    box = {}

    def work():
        try:
            dev0 = torch.device("cuda", 0)
            torch.cuda.set_device(0)
            n = rows_per_device * world
            rows = make_unit_rows(torch, n, dim, 0xC950011, dev0)
            queries = make_unit_rows(torch, 64, dim, 0xC950012, dev0).view(64, 1, dim)
            ns = argparse.Namespace(abi_devices=",".join(str(d) for d in range(world)))
            box["res"] = abi_sharded_leg(ns, torch, np, rows, queries, k, dim)
            box["res"]["rows"] = n
        except BaseException as e:      # noqa: BLE001 - the line must survive
            box["res"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}

    t = threading.Thread(target=work, daemon=True)
    t.start()
    t.join(budget_s)
    if t.is_alive():
        return {"error": "abi sharded leg exceeded its %.0f s budget (left running in a daemon thread)" % budget_s,
                "_hung": True}
    return box.get("res")


def strong_n1_leg(a, torch, k, dim, total_rows, budget_s=120.0):
    """Rank 0, after the process group is gone (N > 1 strong mode): the SAME corpus size on ONE GPU, one query per step -
    the N = 1 point of the strong-scaling curve, measured in the same run (the driver's own N = 1 run is the headline
    configs[1] workload, 1M rows: not comparable with a 10M-row strong-scaling line).  Watchdog thread, never loses the line."""
    import threading
    box = {}

    def work():
        try:
            from cqs_amd import HipIndex
            dev0 = torch.device("cuda", 0)
            torch.cuda.set_device(0)
            rows = make_unit_rows(torch, total_rows, dim, 0xC950021, dev0)
            q = make_unit_rows(torch, 32, dim, 0xC950022, dev0)
            idx = HipIndex.build_from_device(None, rows.data_ptr(), total_rows, dim, borrow=True, keepalive=rows)
            keys = torch.zeros((1, k), dtype=torch.int64, device=dev0)
            cnt = torch.zeros((1,), dtype=torch.int32, device=dev0)
            st = torch.cuda.current_stream().cuda_stream
            for i in range(8):
                idx.search_device(q[i % 32].data_ptr(), 1, k, keys.data_ptr(), cnt.data_ptr(), stream=st)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            steps = 24
            e0.record()
            for i in range(steps):
                idx.search_device(q[i % 32].data_ptr(), 1, k, keys.data_ptr(), cnt.data_ptr(), stream=st)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / steps
            assert int(cnt.item()) == k
            idx.close()
            box["res"] = {"n_gpus": 1, "rows": total_rows, "value": round(1e3 / ms, 2), "unit": "queries/s", "ms_per_step": round(ms, 4),
                          "steps": steps, "note": "the same corpus size on ONE GPU (rank 0, after the timed region): the N = 1 point of this "
                                                  "line's strong-scaling curve"}
        except BaseException as e:      # noqa: BLE001 - the line must survive
            box["res"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}

    t = threading.Thread(target=work, daemon=True)
    t.start()
    t.join(budget_s)
    if t.is_alive():
        return {"error": "strong-scaling N = 1 leg exceeded its %.0f s budget" % budget_s, "_hung": True}
    return box.get("res")


def main():
    a = parse()
    import numpy as np
    import torch
    from cqs_amd import HipIndex, unpack_keys
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

    # ---- what `VectorIndex::search` sees: the synchronous host-buffer entry point (query H2D + results D2H + sync) ----
    latency = None
    if rank == 0 and world == 1 and mode == "single":
        qh = queries[W:W + min(K, 200), 0].cpu().numpy()
        for j in range(min(10, len(qh))):
            idx.search_batch(qh[j], k)
        t1 = time.perf_counter()
        for j in range(len(qh)):
            idx.search_batch(qh[j], k)
        el = time.perf_counter() - t1
        latency = {"queries_per_sec": round(len(qh) / el, 1), "ms_per_query": round(el / len(qh) * 1e3, 4),
                   "what": "cqs_hip_index_search, host query in / host results out, one call at a time"}

    clients = None
    if rank == 0 and world == 1 and mode == "single" and a.extras:
        qc = make_unit_rows(torch, 48, dim, 0xC950031, dev).cpu().numpy()
        clients = concurrent_clients_leg(np, idx, qc, k, dim)

    # The CPU baseline runs LAST (after every GPU leg; only its inputs are taken here, while the corpus is still resident):
    # 10-20 s of one AVX-512 worker per physical core, pinned, over a first-touched 3 GB copy leave the host in a state in
    # which the ticketed embedding leg measured 3 % lower (6 072 against 6 198-6 242 chunks/s, same box, four orders tried:
    # tools/r04_embed_order.sh) - a baseline must not move the thing it is the baseline of.
    cpu = None
    cpu_inputs = None
    if rank == 0 and world == 1 and a.cpu_seconds > 0:
        nqc = min(8, K)
        cpu_inputs = (rows.cpu().numpy(), queries[W:W + nqc, 0].cpu().numpy())

    other = None
    if rank == 0 and world == 1 and mode == "single" and a.extras:
        other = other_configs(torch, np, HipIndex, idx, rows, queries, dim, dev, st)

    abi = None
    strong_n1 = None
    if other and isinstance(other.get("rows10M"), dict) and "queries_per_sec" in other["rows10M"]:
        # N = 1 (headline = configs[1], 1M rows): the N = 1 point of the strong-scaling curve the N > 1 lines measure
        # (configs[4]: ONE 10M-row corpus) is the 10M-row extra of this very run
        strong_n1 = {"n_gpus": 1, "rows": 10_000_000, "value": other["rows10M"]["queries_per_sec"], "unit": "queries/s",
                     "ms_per_step": other["rows10M"]["ms_per_query"],
                     "note": "= other_configs.rows10M: the corpus of the N > 1 lines (`--mode strong`, BASELINE configs[4]) on ONE GPU; "
                             "scale N > 1 values against THIS number, not against the headline `value` (1M rows)"}
    if rank == 0 and world == 1 and mode == "single" and a.abi_devices:
        abi = abi_sharded_leg(a, torch, np, rows, queries[W:], k, dim)
    # N > 1 under RCCL: the C ABI's single-process sharded handle (what the Rust daemon binds) gets its first
    # multi-device run here, on rank 0, after every rank has let go of its shard and left the group - see below.
    abi_after_group = dist is not None and not rehearsal and a.abi_after and torch.cuda.device_count() >= world

    sparse = None
    if rank == 0 and world == 1 and mode == "single" and a.extras and a.sparse_chunks > 0:
        sparse = sparse_index_leg(a, np, idx, queries[W:, 0].cpu().numpy())

    embed = e2e = None
    if a.embed_steps > 0:
        idx.close()
        del rows
        torch.cuda.empty_cache()
        embed, eng, ecfg, eweights = embed_leg(a, rank, world, dist, torch, np, dev, all_reduce_max)
    aux = None
    if rank == 0 and world == 1 and mode == "single" and a.extras and a.embed_steps > 0:
        aux = aux_models_leg(a, np)       # (before the end-to-end leg: that one ends with a minute of CPU forwards for its recall check)
    if a.embed_steps > 0:
        if rank == 0 and world == 1 and a.e2e_chunks > 0:
            e2e = e2e_leg(a, torch, np, dev, eng, ecfg, eweights)
        eng.close()

    if cpu_inputs is not None:
        cpu = cpu_baseline(cpu_inputs[0], cpu_inputs[1], k, a.cpu_seconds)
        cpu_inputs = None
    if embed is not None and rank == 0 and world == 1 and a.cpu_seconds > 0:
        embed["cpu_baseline"] = embed_cpu_baseline(np, ecfg, eweights, a.cpu_seconds, a.embed_len)

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
            abi = abi_after_group_leg(a, torch, np, world, k, dim)
            if mode == "strong" and not (abi or {}).get("_hung"):
                strong_n1 = strong_n1_leg(a, torch, k, dim, total_rows)
    if rank == 0:
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
            "embed": embed,
            "e2e": e2e,
            "aux_models": aux,
            "sparse_index": sparse,
        }
        if rehearsal or (force_dist and world == 1):
            # all ranks on one GPU over gloo, or a 1-rank RCCL group: the N > 1 LOGIC ran, nothing here measures N GPUs
            line["rehearsal"] = True
            line["metric"] = "REHEARSAL (logic check, not a measurement): " + line["metric"]
            line["vs_baseline"] = None
        print(json.dumps(line), flush=True)
    if a.embed_steps <= 0 and not abi_after_group:
        idx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
