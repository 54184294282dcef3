"""The single-process sharded handle (`cqs_hip_index_create_sharded`) and the N = 1 point of the strong-scaling curve."""
import argparse
import time

from .common import make_unit_rows


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
    clients = None
    if getattr(a, "extras", 1):
        clients = sharded_clients(np, sh, qh[:48], [r[0][0] for r in res[:48]], [r[1][0] for r in res[:48]], k, dim)
    sh.close()
    return {"devices": devs, "shards": [{"device": d, "rows": r, "rccl": rc} for d, _f, r, rc in info],
            "queries_per_sec_host_api": round(nq / el, 1), "ms_per_query": round(el / nq * 1e3, 4),
            "single_device_same_queries": {"queries_per_sec_host_api": round(nq / el1, 1), "ms_per_query": round(el1 / nq * 1e3, 4)},
            "vs_single_device": round(el1 / el, 4), "queries": nq,
            "build_s": round(t_build, 2), "checked_vs_single_device": True, "checked": True, "concurrent_clients": clients,
            "what": "cqs_hip_index_create_sharded -> per-shard scan + select -> gather -> host merge, blocking host API, "
                    "one query per call; a device named more than once gathers without RCCL (one-GPU form)"}


def sharded_clients(np, sh, q, want_rows, want_scores, k, dim):
    """Concurrent single-query callers on the SHARDED handle (round 5: its own combining queue): N native
    threads, each one blocking cqs_hip_index_search(b = 1) at a time; every answer compared bit for bit with the lone call's."""
    import ctypes as C
    storm = sh._lib.cqs_hip_debug_client_storm
    storm.restype = C.c_double
    storm.argtypes = [C.c_void_p, C.c_void_p] + [C.c_uint32] * 5 + [C.c_void_p] * 3
    q = np.ascontiguousarray(q, dtype=np.float32)
    nq = q.shape[0]
    want_r, want_s = np.stack(want_rows), np.stack(want_scores)
    out = {"what": "N native threads on the sharded handle, one blocking cqs_hip_index_search(b = 1) each at a time; every answer "
                   "bit-identical to the lone call's (checked)", "native_threads": {}}
    for T in (1, 4, 8, 16):
        per = max(40, 960 // T)
        rows = np.zeros((nq, k), np.uint64)
        scores = np.zeros((nq, k), np.float32)
        counts = np.zeros((nq,), np.uint32)
        storm(sh._h, q.ctypes.data, nq, dim, k, T, 16, rows.ctypes.data, scores.ctypes.data, counts.ctypes.data)   # warm
        p0, q0 = sh.combine_stats()
        el = storm(sh._h, q.ctypes.data, nq, dim, k, T, per, rows.ctypes.data, scores.ctypes.data, counts.ctypes.data)
        p1, q1 = sh.combine_stats()
        assert el > 0, "a client call failed"
        assert np.all(counts == k) and np.array_equal(rows, want_r) and np.array_equal(scores, want_s), "combined sharded answers differ from the lone call's"
        out["native_threads"][str(T)] = {"queries_per_sec": round(T * per / el, 1), "ms_per_call": round(el / per * 1e3, 4),
                                          "mean_callers_per_pass": round((q1 - q0) / max(p1 - p0, 1), 2), "checked": True}
    return out


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
