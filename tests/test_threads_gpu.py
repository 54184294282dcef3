"""The header promises that calls on one handle may come from several host threads (include/cqs_hip.h, threading):
each handle serialises its entry points internally.  Two / three threads hammer one index, one embedding engine and one
BERT engine; every result must equal the single-threaded answer bit for bit."""
import threading

import numpy as np
import pytest

from cqs_amd import HipIndex, synth
from cqs_amd.index import HipError

pytestmark = pytest.mark.gpu


def _run_threads(fn, n_threads):
    errs = []

    def wrap(t):
        try:
            fn(t)
        except BaseException as e:  # noqa: BLE001 - surfaced below
            errs.append((t, repr(e)))

    th = [threading.Thread(target=wrap, args=(t,)) for t in range(n_threads)]
    [x.start() for x in th]
    [x.join() for x in th]
    assert not errs, errs


def test_index_search_from_three_threads(hip):
    rows = synth.gaussian_unit(60_000, seed=5)
    qs = synth.gaussian_unit(24, seed=6)
    idx = HipIndex.build_from_flat(None, rows)
    want = [idx.search_batch(qs[i], 50) for i in range(len(qs))]
    want_blk = idx.search_batch(qs, 20)

    def work(t):
        for rep in range(6):
            for i in range(t, len(qs), 3):
                r, s, c = idx.search_batch(qs[i], 50)
                assert np.array_equal(r, want[i][0]) and np.array_equal(s, want[i][1]) and np.array_equal(c, want[i][2]), (t, rep, i)
            r, s, c = idx.search_batch(qs, 20)
            assert np.array_equal(r, want_blk[0]) and np.array_equal(s, want_blk[1])

    _run_threads(work, 3)
    idx.close()


def test_combining_queue_hands_every_caller_its_lone_answer(hip):
    """concurrent single-query callers share passes over the corpus (include/cqs_hip.h, search: "Concurrent
    callers").  8 and 16 threads, one query per call; every answer must equal, bit for bit, what the same call returns
    alone - and the queue must really have combined (more queries than passes)."""
    rows = synth.gaussian_unit(300_000, seed=15)
    qs = synth.gaussian_unit(64, seed=16)
    idx = HipIndex.build_from_flat(None, rows)
    want = [idx.search_batch(qs[i], 20) for i in range(len(qs))]
    p0, q0 = idx.combine_stats()
    assert p0 == q0 == len(qs)                 # a lone caller: one query per pass

    for n_threads in (8, 16):
        def work(t):
            for rep in range(12):
                for i in range(t, len(qs), n_threads):
                    r, s, c = idx.search_batch(qs[i], 20)
                    assert np.array_equal(r, want[i][0]) and np.array_equal(s, want[i][1]) and np.array_equal(c, want[i][2]), (t, rep, i)
        _run_threads(work, n_threads)
    p1, q1 = idx.combine_stats()
    assert q1 - q0 == 2 * 12 * len(qs)
    assert (q1 - q0) > 1.5 * (p1 - p0), f"the queue did not combine: {q1 - q0} queries in {p1 - p0} passes"
    idx.close()


def test_combining_queue_mixed_callers(hip):
    """Callers with different k, PIPELINE mode, a bitset, a multi-query block and a non-finite query all at once: each
    group rides its own passes (or the serial path) and gets the answer it gets alone."""
    from cqs_amd import _lib
    rows = synth.gaussian_unit(120_000, seed=25)
    qs = synth.gaussian_unit(32, seed=26)
    idx = HipIndex.build_from_flat(None, rows)
    keep = np.random.default_rng(27).integers(0, 2**32, size=(len(rows) + 31) // 32, dtype=np.uint64).astype(np.uint32)
    kinds = [dict(k=20), dict(k=50), dict(k=20, mode=_lib.MODE_PIPELINE, threshold=0.05), dict(k=10, keep_bitset=keep)]
    want = {(j, i): idx.search_batch(qs[i], **kw) for j, kw in enumerate(kinds) for i in range(len(qs))}
    want_blk = idx.search_batch(qs[:12], 20)
    nanq = qs[0].copy()
    nanq[5] = np.nan

    def work(t):
        j = t % len(kinds)
        for rep in range(6):
            for i in range(t % 8, len(qs), 8):
                r, s, c = idx.search_batch(qs[i], **kinds[j])
                w = want[(j, i)]
                assert np.array_equal(r, w[0]) and np.array_equal(s, w[1]) and np.array_equal(c, w[2]), (t, rep, i)
            if t == 0:
                r, s, c = idx.search_batch(qs[:12], 20)
                assert np.array_equal(r, want_blk[0]) and np.array_equal(s, want_blk[1])
            if t == 1:
                assert int(idx.search_batch(nanq, 20)[2][0]) == 0

    _run_threads(work, 12)
    idx.close()


def test_poisoned_handle_wakes_every_parked_caller(hip):
    """A pass that fails poisons the handle (src/cagra.rs:472-489): the call that led it reports the device error, every
    caller parked behind it - and every later call - returns CQS_HIP_ERR_POISONED; nobody hangs."""
    import ctypes as C
    from cqs_amd import _lib
    rows = synth.gaussian_unit(200_000, seed=35)
    qs = synth.gaussian_unit(16, seed=36)
    idx = HipIndex.build_from_flat(None, rows)
    lib = _lib.load()
    lib.cqs_hip_debug_index_fail_next.argtypes = [C.c_void_p]
    lib.cqs_hip_debug_index_fail_next.restype = None
    codes = []
    lock = threading.Lock()
    start = threading.Barrier(9)

    def work(t):
        start.wait()
        for rep in range(40):
            try:
                idx.search_batch(qs[(t + rep) % len(qs)], 20)
            except HipError as e:
                with lock:
                    codes.append(e.code)

    th = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    [x.start() for x in th]
    start.wait()
    lib.cqs_hip_debug_index_fail_next(idx._h)
    [x.join(timeout=60) for x in th]
    assert not any(x.is_alive() for x in th), "a caller is still parked on a poisoned handle"
    assert idx.is_poisoned()
    assert codes.count(_lib.ERR_DEVICE) == 1, codes
    assert codes.count(_lib.ERR_POISONED) == len(codes) - 1 and len(codes) >= 8, codes
    idx.close()


def test_sharded_handle_takes_the_combining_queue(hip):
    """the row-sharded parent (what a multi-GPU daemon binds; one-GPU form: device 0 named three times)
    combines concurrent single-query callers too.  Every answer equals the lone call's bit for bit, the queue really
    combined, and mixed parameters / a multi-query block / a bitset still get their own answers."""
    rows = synth.gaussian_unit(200_000, seed=45)
    qs = synth.gaussian_unit(48, seed=46)
    sh = HipIndex.build_sharded(None, rows, [0, 0, 0])
    single = HipIndex.build_from_flat(None, rows)
    want = [sh.search_batch(qs[i], 20) for i in range(len(qs))]
    want50 = [sh.search_batch(qs[i], 50) for i in range(len(qs))]
    for i in range(len(qs)):                                 # the sharded answer is the single-device answer (scores: same kernels per row)
        r1, s1, c1 = single.search_batch(qs[i], 20)
        assert np.array_equal(c1, want[i][2]) and np.max(np.abs(s1 - want[i][1])) <= 2e-6
    keep = np.random.default_rng(47).integers(0, 2**32, size=(len(rows) + 31) // 32, dtype=np.uint64).astype(np.uint32)
    want_keep = sh.search_batch(qs[3], 10, keep_bitset=keep)
    want_blk = sh.search_batch(qs[:12], 20)
    p0, q0 = sh.combine_stats()
    assert p0 == q0 == 2 * len(qs)                          # lone callers: one query per pass

    def work(t):
        for rep in range(8):
            for i in range(t, len(qs), 8):
                w = want50 if t % 4 == 3 else want
                r, s, c = sh.search_batch(qs[i], 50 if t % 4 == 3 else 20)
                assert np.array_equal(r, w[i][0]) and np.array_equal(s, w[i][1]) and np.array_equal(c, w[i][2]), (t, rep, i)
            if t == 0:
                r, s, c = sh.search_batch(qs[:12], 20)
                assert np.array_equal(r, want_blk[0]) and np.array_equal(s, want_blk[1])
            if t == 1:
                r, s, c = sh.search_batch(qs[3], 10, keep_bitset=keep)
                assert np.array_equal(r, want_keep[0]) and np.array_equal(s, want_keep[1])

    _run_threads(work, 8)
    p1, q1 = sh.combine_stats()
    assert q1 - q0 == 8 * len(qs)
    assert (q1 - q0) > 1.5 * (p1 - p0), f"the sharded queue did not combine: {q1 - q0} queries in {p1 - p0} passes"
    single.close()
    sh.close()


def test_poisoned_sharded_handle_wakes_every_parked_caller(hip):
    """The poisoned case on the sharded parent: the leading call reports the device error, everybody parked behind it and
    every later call gets CQS_HIP_ERR_POISONED; nobody hangs."""
    import ctypes as C
    from cqs_amd import _lib
    rows = synth.gaussian_unit(200_000, seed=55)
    qs = synth.gaussian_unit(16, seed=56)
    sh = HipIndex.build_sharded(None, rows, [0, 0])
    lib = _lib.load()
    lib.cqs_hip_debug_index_fail_next.argtypes = [C.c_void_p]
    lib.cqs_hip_debug_index_fail_next.restype = None
    codes = []
    lock = threading.Lock()
    start = threading.Barrier(9)

    def work(t):
        start.wait()
        for rep in range(40):
            try:
                sh.search_batch(qs[(t + rep) % len(qs)], 20)
            except HipError as e:
                with lock:
                    codes.append(e.code)

    th = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    [x.start() for x in th]
    start.wait()
    lib.cqs_hip_debug_index_fail_next(sh._h)
    [x.join(timeout=60) for x in th]
    assert not any(x.is_alive() for x in th), "a caller is still parked on a poisoned sharded handle"
    assert sh.is_poisoned()
    assert codes.count(_lib.ERR_DEVICE) == 1, codes
    assert codes.count(_lib.ERR_POISONED) == len(codes) - 1 and len(codes) >= 8, codes
    sh.close()


def test_a_lone_caller_after_a_burst_does_not_wait_for_stragglers(hip, monkeypatch):
    """The leader's straggler window is anchored at the end of the previous pass (round 5): with a 50 ms window, a caller
    that comes alone 200 ms after an 8-thread burst must not pay it (round 4: it paid the whole window once)."""
    import time
    monkeypatch.setenv("CQS_HIP_COMBINE_WAIT_US", "50000")
    rows = synth.gaussian_unit(100_000, seed=65)
    qs = synth.gaussian_unit(8, seed=66)
    idx = HipIndex.build_from_flat(None, rows)
    monkeypatch.delenv("CQS_HIP_COMBINE_WAIT_US")
    for i in range(8):
        idx.search_batch(qs[i], 20)

    def work(t):
        for rep in range(10):
            idx.search_batch(qs[t], 20)

    _run_threads(work, 8)
    p0, q0 = idx.combine_stats()
    time.sleep(0.2)
    t0 = time.perf_counter()
    idx.search_batch(qs[0], 20)
    dt = time.perf_counter() - t0
    p1, q1 = idx.combine_stats()
    assert (p1 - p0, q1 - q0) == (1, 1)
    assert dt < 0.025, f"a lone caller waited {dt * 1e3:.1f} ms behind a burst that ended 200 ms earlier"
    idx.close()


@pytest.mark.parametrize("sharded", [False, True])
def test_relaxed_combining_mode_stays_inside_the_parity_tolerance(hip, monkeypatch, sharded):
    """CQS_HIP_COMBINE_BITS=relaxed (opt-in, read at create): a combined block of >= 9 callers may run on the matrix cores -
    32 queries per corpus sweep instead of 8.  Answers are then NOT the lone call's bits (another summation order) but must
    stay inside the parity tolerance: scores within 2e-6, ids equal wherever neighbouring scores are further apart than
    that, and the queue must really have formed blocks of > 8.  The default mode on the same corpus stays bit-identical
    (the tests above)."""
    rows = synth.gaussian_unit(200_000, seed=75)
    qs = synth.gaussian_unit(64, seed=76)
    strict = HipIndex.build_from_flat(None, rows)
    want = [strict.search_batch(qs[i], 20) for i in range(len(qs))]
    strict.close()
    monkeypatch.setenv("CQS_HIP_COMBINE_BITS", "relaxed")
    idx = HipIndex.build_sharded(None, rows, [0, 0]) if sharded else HipIndex.build_from_flat(None, rows)
    monkeypatch.delenv("CQS_HIP_COMBINE_BITS")
    # native threads (the interpreter lock keeps Python threads from ever parking more than a few callers at once)
    import ctypes as C
    storm = idx._lib.cqs_hip_debug_client_storm
    storm.restype = C.c_double
    storm.argtypes = [C.c_void_p, C.c_void_p] + [C.c_uint32] * 5 + [C.c_void_p] * 3
    nq, k, dim = len(qs), 20, rows.shape[1]
    q = np.ascontiguousarray(qs, dtype=np.float32)
    got_r = np.zeros((nq, k), np.uint64); got_s = np.zeros((nq, k), np.float32); got_c = np.zeros((nq,), np.uint32)
    p0, q0 = idx.combine_stats()
    el = storm(idx._h, q.ctypes.data, nq, dim, k, 16, 40, got_r.ctypes.data, got_s.ctypes.data, got_c.ctypes.data)
    p1, q1 = idx.combine_stats()
    assert el > 0 and q1 - q0 == 16 * 40
    for i in range(nq):
        wr, ws, wc = want[i]
        assert got_c[i] == wc[0] and np.max(np.abs(got_s[i] - ws[0])) <= 2e-6, i
        gap = np.abs(np.diff(ws[0])) <= 4e-6
        far = np.ones(k, bool); far[:-1] &= ~gap; far[1:] &= ~gap
        assert np.array_equal(got_r[i][far], wr[0][far]), i
    assert (q1 - q0) > 8.5 * (p1 - p0), f"no block of more than 8 callers formed: {q1 - q0} queries in {p1 - p0} passes"
    idx.close()


def test_embed_engine_from_two_threads(hip):
    from test_embed_gpu import SMALL, batch, make
    eng, _ = make(SMALL, seed=91)
    batches = [batch(SMALL, [int(x) for x in np.random.default_rng(92 + j).integers(1, SMALL.max_seq, size=6)], seed=93 + j) for j in range(4)]
    want = [eng.run(i, m).copy() for i, m in batches]

    def work(t):
        for rep in range(8):
            j = (2 * rep + t) % len(batches)
            ids, mask = batches[j]
            if rep % 2:
                got = eng.run(ids, mask)
            else:
                got = eng.collect(eng.submit(ids, mask), len(ids))
            assert np.array_equal(got, want[j]), (t, rep, j)

    _run_threads(work, 2)          # (3 slots: two threads never exhaust them)
    eng.close()


def test_bert_engine_from_two_threads(hip):
    from oracle import bert_ref as R
    from test_bert_gpu import _engine, _seqs
    cfg = R.BertConfig(vocab_size=800, hidden=384, layers=2, heads=6, intermediate=768, max_pos=100)
    eng, _ = _engine(cfg, "mlm", seed=95)
    sets = [_seqs(cfg, [30, 100, 7], seed=96), _seqs(cfg, [64, 65], seed=97)]
    want = [eng.splade_dense(s).copy() for s in sets]

    def work(t):
        for rep in range(8):
            j = (rep + t) % 2
            assert np.array_equal(eng.splade_dense(sets[j]), want[j]), (t, rep)

    _run_threads(work, 2)
    eng.close()
