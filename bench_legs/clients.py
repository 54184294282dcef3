"""What N daemon client threads see on one dense index handle (the combining queue)."""
import time


def concurrent_clients_leg(np, idx, qh, k, dim, rows_ptr=None):
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
    nq = 96 if len(qh) >= 96 else 48                      # a multiple of every thread count below
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
    # Opt-in throughput mode (CQS_HIP_COMBINE_BITS=relaxed, read at create): blocks of >= 9 callers run on the matrix cores,
    # 32 queries per corpus sweep instead of 8 - answers within the parity tolerance of the lone call's (scores <= 2e-6 apart,
    # ids equal wherever neighbouring scores are further apart than that), not bit-identical.  A second handle over the same rows.
    if rows_ptr is None:                                  # (the caller did not say where the corpus lies: strict mode only)
        return out
    import os
    from cqs_amd import HipIndex
    os.environ["CQS_HIP_COMBINE_BITS"] = "relaxed"
    try:
        rx = HipIndex.build_from_device(None, rows_ptr, len(idx), dim, borrow=True, keepalive=idx)
    finally:
        del os.environ["CQS_HIP_COMBINE_BITS"]
    out["native_threads_relaxed_bits"] = {"what": "the same storm on a handle created under CQS_HIP_COMBINE_BITS=relaxed: combined blocks of >= 9 "
                                                  "callers use the matrix-core kernel (32 queries per sweep); every answer within 2e-6 of the lone "
                                                  "call's scores, same ids outside near-ties (checked) - NOT bit-identical"}
    for T in (8, 16, 32):
        per = max(60, 1920 // T)
        rows = np.zeros((nq, k), np.uint64)
        scores = np.zeros((nq, k), np.float32)
        counts = np.zeros((nq,), np.uint32)
        nqt = nq if nq % T == 0 else 48
        storm(rx._h, q.ctypes.data, nqt, dim, k, T, 24, rows.ctypes.data, scores.ctypes.data, counts.ctypes.data)   # warm
        p0, q0 = rx.combine_stats()
        el = storm(rx._h, q.ctypes.data, nqt, dim, k, T, per, rows.ctypes.data, scores.ctypes.data, counts.ctypes.data)
        p1, q1 = rx.combine_stats()
        assert el > 0, "a client call failed"
        for i in range(nqt):
            assert counts[i] == k and np.max(np.abs(scores[i] - want_s[i])) <= 2e-6, "relaxed answer outside the tolerance"
            far = np.ones(k, bool)
            gap = np.abs(np.diff(want_s[i])) <= 4e-6
            far[:-1] &= ~gap
            far[1:] &= ~gap
            assert np.array_equal(rows[i][far], want_r[i][far]), "relaxed answer: ids differ outside near-ties"
        out["native_threads_relaxed_bits"][str(T)] = {"queries_per_sec": round(T * per / el, 1), "ms_per_call": round(el / per * 1e3, 4),
                                                       "mean_callers_per_pass": round((q1 - q0) / max(p1 - p0, 1), 2), "checked_within_tolerance": True}
    rx.close()
    return out
