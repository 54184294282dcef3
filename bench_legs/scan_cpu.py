"""`cpu_baseline` of the headline: the oracle's scan timed on this host's cores (reported beside `value`, never part of it)."""
import os
import time

from .common import physical_cores


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
