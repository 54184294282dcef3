"""Parity of the HIP sparse index (cqs_hip_sparse_index_*, through the C ABI via cqs_amd.splade_index) with the oracle
restatement of `SpladeIndex` (src/splade/index.rs:177-290): chunk order and score BITS identical (integer / f32-sum work:
the bar is bit-exact), on the reference's own known-answer cases, on seeded corpora from 1 chunk to 1M, and through
size-independent properties."""
import numpy as np
import pytest

import sparse_cases as sc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S(hip):
    from cqs_amd import splade_index
    return splade_index


def _same(oracle_ix, hip_ix, qt, qw, k, keep=None):
    oc, os_ = oracle_ix.search_raw(qt, qw, k, keep)
    hc, hs, rc = hip_ix.search_raw(qt, qw, k, keep)
    assert rc == 0
    assert hc.size == oc.size, (hc.size, oc.size)
    assert np.array_equal(hs.view(np.uint32), os_.view(np.uint32)), "score bits differ"
    assert np.array_equal(hc, oc), "chunk order differs"
    return hc, hs


def test_reference_kats_through_the_abi(S, oracle):    # src/splade/index.rs:1113-1242
    from test_sparse_oracle import _check_case
    k = sc.kats()
    chunks = [(cid, [(int(t), float(w)) for t, w in sv]) for cid, sv in k["index"]]
    ix = S.HipSpladeIndex.build(chunks)
    assert len(ix) == k["len"] and not ix.is_empty() and ix.unique_tokens() == 5 and ix.postings() == 8
    for case in k["cases"]:
        _check_case(ix, case)
    empty = S.HipSpladeIndex.build([])
    assert empty.is_empty() and empty.unique_tokens() == 0 and empty.search([(1, 1.0)], 10) == []
    ix.close(); empty.close()


@pytest.mark.parametrize("n,vocab,nnz,terms,k", [
    (1, 50, (1, 8), 6, 10), (63, 80, (0, 12), 20, 500), (64, 80, (1, 12), 20, 64), (65, 200, (1, 30), 70, 1024),
    (1000, 300, (5, 40), 65, 500), (5000, 2000, (20, 120), 130, 500), (70001, 30522, (40, 160), 90, 1000),
])
def test_seeded_corpora_bit_exact(S, oracle, n, vocab, nnz, terms, k):
    rng = np.random.default_rng(n * 7919 + terms)
    off, tok, w = sc.corpus(rng, n, vocab, nnz[0], nnz[1], dup_frac=0.3, special=True)
    rank = rng.permutation(n).astype(np.uint32)
    o = oracle.SpladeIndex(off, tok, w, id_rank=rank)
    h = S.HipSpladeIndex.build_from_csr(None, off, tok, w, id_rank=rank)
    assert len(h) == n and h.unique_tokens() == o.unique_tokens()
    for trial in range(4):
        qt, qw = sc.query(rng, vocab, terms, shuffle=trial != 1, dups=3 if trial == 2 else 0, absent=4)
        if trial == 3:
            qw[::5] *= np.float32(-1.0)
        _same(o, h, qt, qw, k)
        keep = rng.random(n) < (0.5 if trial % 2 else 0.05)
        _same(o, h, qt, qw, k, keep)
    _same(o, h, qt, qw, 1)
    _same(o, h, qt, qw, k, np.zeros(n, bool))            # nothing kept
    h.close()


def test_equal_scores_follow_id_order_and_string_ids(S, oracle):
    """Many identical documents (duplicated code chunks): every score ties; the cut at k must keep the smallest ids
    (BoundedScoreHeap, candidate.rs:299-334) - with string ids, and with duplicate ids too (the reference allows them)."""
    rng = np.random.default_rng(5)
    base = [(3, 0.5), (9, 1.25), (11, 0.75)]
    ids = ["c%05d" % i for i in rng.permutation(3000)]
    ids[10] = ids[20]                                      # a duplicate id
    chunks = [(cid, base if i % 3 else base + [(40, 0.1)]) for i, cid in enumerate(ids)]
    o = oracle.SpladeIndex.build(chunks)
    h = S.HipSpladeIndex.build(chunks)
    q = [(9, 2.0), (3, 1.0), (40, 5.0)]
    for k in (1, 7, 500, 1024):
        got = h.search(q, k)
        want = o.search(q, k)
        assert [(r.id, r.score) for r in got] == want
    flt = lambda cid: cid.endswith("7")
    assert [(r.id, r.score) for r in h.search_with_filter(q, 300, flt)] == o.search_with_filter(q, 300, flt)
    h.close()


def test_non_finite_weights_and_guards(S, oracle, hip):
    rng = np.random.default_rng(11)
    off, tok, w = sc.corpus(rng, 4000, 500, 5, 40)
    w[rng.integers(0, w.size, 50)] = np.float32("inf")
    w[rng.integers(0, w.size, 50)] = np.float32("nan")
    w[rng.integers(0, w.size, 50)] = np.float32("-inf")
    o = oracle.SpladeIndex(off, tok, w)
    h = S.HipSpladeIndex.build_from_csr(None, off, tok, w)
    for trial in range(3):
        qt, qw = sc.query(rng, 500, 40)
        if trial == 1:
            qw[3] = np.float32("nan")
        if trial == 2:
            qw[5] = np.float32("inf"); qw[6] = np.float32(0.0)
        hc, hs = _same(o, h, qt, qw, 500)
        assert np.all(np.isfinite(hs))
    # guards: k = 0, empty query, k over the cap (refused, the mirror returns nothing and keeps the message)
    assert h.search_raw(qt, qw, 0)[0].size == 0 and h.search_raw([], [], 10)[0].size == 0
    _c, _s, rc = h.search_raw(qt, qw, 1025)
    assert rc == -1 and "MAX_K" in h.last_error
    # the reserved NaN payload is refused at build and in a query
    bad = w.copy(); bad.view(np.uint32)[7] = 0xFFFFFFFF
    with pytest.raises(S.HipError):
        S.HipSpladeIndex.build_from_csr(None, off, tok, bad)
    qbad = qw.copy(); qbad.view(np.uint32)[0] = 0xFFFFFFFF
    assert h.search_raw(qt, qbad, 10)[2] == -1
    # not a permutation
    with pytest.raises(S.HipError):
        S.HipSpladeIndex.build_from_csr(None, off, tok, w, id_rank=np.zeros(4000, np.uint32))
    h.close()


def test_full_u32_token_ids(S, oracle):
    """Token ids are any u32 (the reference's proptest draws from the whole range, index.rs:1750-1753): the sort path of
    the token table."""
    rng = np.random.default_rng(21)
    off, tok, w = sc.corpus(rng, 3000, 400, 3, 30, dup_frac=0.2)
    remap = rng.choice(np.arange(0, 2 ** 32, dtype=np.uint64), size=400, replace=False).astype(np.uint32)
    remap[0] = 0; remap[1] = 0xFFFFFFFF
    o = oracle.SpladeIndex(off, remap[tok], w)
    h = S.HipSpladeIndex.build_from_csr(None, off, remap[tok], w)
    for _ in range(3):
        qt, qw = sc.query(rng, 400, 50)
        _same(o, h, remap[qt], qw, 500)
    h.close()


def test_one_million_chunks_against_the_oracle_and_properties(S, oracle):
    """BASELINE-size corpus (1M chunks, ~96 postings each): the oracle still answers in seconds, so this is parity proper;
    plus properties that need no oracle: linearity in the query weights (exact for powers of two), a filter that keeps
    exactly the winners returns them unchanged, sortedness."""
    rng = np.random.default_rng(1234)
    n, vocab = 1_000_000, 30522
    lens = rng.integers(64, 129, size=n)
    off = np.zeros(n + 1, np.uint64); off[1:] = np.cumsum(lens)
    P = int(off[-1])
    # Zipf-like token draw without per-document loops: ids = floor(vocab * u^3); duplicates inside a document are left in
    # (each is its own posting, as in the reference)
    tok = np.minimum((vocab * rng.random(P) ** 3).astype(np.uint32), vocab - 1)
    w = (rng.random(P, dtype=np.float32) * 2.0 + 0.01).astype(np.float32)
    o = oracle.SpladeIndex(off, tok, w)
    h = S.HipSpladeIndex.build_from_csr(None, off, tok, w)
    qt = np.unique(np.minimum((vocab * rng.random(80) ** 3).astype(np.uint32), vocab - 1))
    rng.shuffle(qt)
    qw = (rng.random(qt.size, dtype=np.float32) + 0.1).astype(np.float32)
    hc, hs = _same(o, h, qt, qw, 500)
    assert np.all(hs[:-1] >= hs[1:])
    ms, touched = h.last_search()
    assert touched == o.touched(qt) and ms == 0     # nobody had asked for the launch's time yet: that search was not bracketed by events
    h.search_raw(qt, qw, 500)
    ms, touched = h.last_search()
    assert touched == o.touched(qt) and ms > 0      # ... from the first request on, every search is
    hc2, hs2, _ = h.search_raw(qt, qw * np.float32(4.0), 500)
    assert np.array_equal(hc2, hc) and np.array_equal(hs2, hs * np.float32(4.0))
    keep = np.zeros(n, bool); keep[hc.astype(np.int64)] = True
    hc3, hs3, _ = h.search_raw(qt, qw, 500, keep)
    assert np.array_equal(hc3, hc) and np.array_equal(hs3, hs)
    _same(o, h, qt, qw, 1000, rng.random(n) < 0.3)
    h.close()


def test_many_terms_and_concurrent_callers(S, oracle):
    """A query far longer than one 64-term group (with repeats: the same list is walked again, in order), a term table that
    has to grow between calls, and four threads on one handle (the handle serialises them; every answer must still be the
    oracle's)."""
    import threading
    rng = np.random.default_rng(77)
    off, tok, w = sc.corpus(rng, 6000, 900, 10, 50, dup_frac=0.1)
    o = oracle.SpladeIndex(off, tok, w)
    h = S.HipSpladeIndex.build_from_csr(None, off, tok, w)
    for terms in (10, 300, 1500, 40):
        qt = rng.integers(0, 900, size=terms).astype(np.uint32)            # repeats included
        qw = (rng.random(terms, dtype=np.float32) - np.float32(0.3)).astype(np.float32)
        _same(o, h, qt, qw, 500)
    queries = [sc.query(rng, 900, 30 + 5 * i) for i in range(16)]
    want = [o.search_raw(qt, qw, 200) for qt, qw in queries]
    got = [None] * len(queries)

    def work(t):
        for i in range(t, len(queries), 4):
            for _ in range(5):
                got[i] = h.search_raw(queries[i][0], queries[i][1], 200)

    th = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    [t.start() for t in th]; [t.join() for t in th]
    for (oc, os_), (hc, hs, rc) in zip(want, got):
        assert rc == 0 and np.array_equal(hc, oc) and np.array_equal(hs.view(np.uint32), os_.view(np.uint32))
    h.close()


def test_the_input_space_of_the_reference_proptest(S, oracle):
    """The reference's own generators for `SpladeIndex` inputs (src/splade/index.rs:1717-1762): 0..8 chunks, 0..6 entries
    each, token ids over the whole u32 range, DUPLICATE tokens inside a vector, weights from every f32 class (ordinary,
    +-0, f32::MIN / MAX, MIN_POSITIVE, the smallest subnormal, +-inf, NaN), ids empty / ASCII / multi-byte and not unique.
    There they feed the save / load round trip; here every such index is searched on both sides - same ids, same score bits."""
    rng = np.random.default_rng(2024)
    special = np.array([0.0, -0.0, np.finfo(np.float32).min, np.finfo(np.float32).max, np.finfo(np.float32).tiny,
                        np.float32(1e-45), np.inf, -np.inf, np.nan], dtype=np.float32)
    id_pool = ["", "a", "chunk_a", "src/lib.rs:12", "a/b-c.d_e", "été", "中文", "\U0001F600x", "zz", "a"]

    def weight():
        if rng.random() < 0.45:
            return float(np.float32(rng.standard_normal() * 10.0 ** rng.integers(-3, 4)))
        return float(special[rng.integers(0, special.size)])

    def token(pool):
        return int(pool[rng.integers(0, len(pool))])

    checked = 0
    for case in range(150):
        pool = [0, 0xFFFFFFFF] + [int(x) for x in rng.integers(0, 2 ** 32, size=4, dtype=np.uint64)]
        chunks = []
        for _ in range(int(rng.integers(0, 9))):
            vec = [(token(pool), weight()) for _ in range(int(rng.integers(0, 7)))]
            chunks.append((id_pool[rng.integers(0, len(id_pool))], vec))
        o = oracle.SpladeIndex.build(chunks)
        h = S.HipSpladeIndex.build(chunks)
        assert len(h) == len(o) and h.unique_tokens() == o.unique_tokens()
        for _q in range(3):
            query = [(token(pool), weight() if rng.random() < 0.5 else float(np.float32(rng.random() + 0.1)))
                     for _ in range(int(rng.integers(0, 6)))]
            k = int(rng.integers(0, 12))
            got = [(r.id, r.score) for r in h.search(query, k)]
            want = o.search(query, k)
            # equal (score, id) pairs - duplicate ids - may come out in either order: compare as sorted lists of bit patterns
            key = lambda t: (t[0], np.float32(t[1]).view(np.uint32).item())
            assert sorted(map(key, got)) == sorted(map(key, want)), (case, chunks, query, k, got, want)
            assert [key(t)[1] for t in got] == [key(t)[1] for t in want]
            checked += 1
        h.close()
    assert checked == 450


def test_built_from_the_postings_map_equals_built_from_the_documents(S, oracle):
    """`cqs_hip_sparse_index_create_inverted`: the reference's in-memory form (token -> [(chunk, weight)] in push order,
    index.rs:177-212) as input - keys in arbitrary order, one list handed over descending, a posting that names a chunk past
    the end (skipped at search time by the reference, index.rs:252).  Same answers, bit for bit, as the index built from
    the documents, and as the oracle."""
    rng = np.random.default_rng(404)
    n = 5000
    off, tok, w = sc.corpus(rng, n, 700, 5, 40, dup_frac=0.3, special=True)
    ids = ["id%05d" % i for i in rng.permutation(n)]
    postings = {}
    for d in range(n):                                                       # SpladeIndex::build's loop (index.rs:197-202)
        for e in range(int(off[d]), int(off[d + 1])):
            postings.setdefault(int(tok[e]), []).append((d, float(w[e])))
    keys = list(postings.keys())
    rng.shuffle(keys)
    shuffled = {k: postings[k] for k in keys}
    some = keys[0]
    shuffled[some] = sorted(shuffled[some], key=lambda cw: -cw[0])           # descending chunks; no chunk twice under this token?
    if len({c for c, _ in shuffled[some]}) != len(shuffled[some]):           # (a repeated chunk's postings must keep their order)
        shuffled[some] = postings[some]
    shuffled[keys[1]] = shuffled[keys[1]] + [(n + 7, 3.0)]                   # corrupt posting: chunk past the end
    a = S.HipSpladeIndex.build_from_csr(ids, off, tok, w)
    b = S.HipSpladeIndex.build_from_postings(ids, shuffled)
    o = oracle.SpladeIndex(off, tok, w, ids=ids)
    assert len(b) == n and b.unique_tokens() == a.unique_tokens() == o.unique_tokens() and b.postings() == a.postings()
    for trial in range(4):
        qt, qw = sc.query(rng, 700, 45, dups=2 if trial == 1 else 0, absent=3)
        keep = None if trial < 2 else rng.random(n) < 0.4
        ac, as_, _ = a.search_raw(qt, qw, 500, keep)
        bc, bs, rc = b.search_raw(qt, qw, 500, keep)
        oc, os_ = o.search_raw(qt, qw, 500, keep)
        assert rc == 0 and np.array_equal(ac, bc) and np.array_equal(as_.view(np.uint32), bs.view(np.uint32))
        assert np.array_equal(bc, oc) and np.array_equal(bs.view(np.uint32), os_.view(np.uint32))
    with pytest.raises(S.HipError):                                          # a key twice is not a map
        lib = S._lib.load()
        t2 = np.array([5, 5], np.uint32); o2 = np.array([0, 1, 2], np.uint64); c2 = np.array([0, 1], np.uint32); w2 = np.ones(2, np.float32)
        hh = S.C.c_void_p()
        rc = lib.cqs_hip_sparse_index_create_inverted(t2.ctypes.data, o2.ctypes.data, c2.ctypes.data, w2.ctypes.data, 2, 10, None, 0, S.C.byref(hh))
        if rc != 0:
            raise S.HipError(rc, "refused")
    a.close(); b.close()


def test_batched_queries_equal_the_single_calls(S, oracle):
    """`cqs_hip_sparse_index_search_batch`: 1, 7, 33 and 64 queries per call (one of them empty, one naming only absent
    tokens, very different lengths), with and without a filter - every row identical, bit for bit, to that query's own
    call and to the oracle; 65 queries are refused."""
    rng = np.random.default_rng(909)
    n = 30000
    off, tok, w = sc.corpus(rng, n, 1200, 8, 60, dup_frac=0.1)
    rank = rng.permutation(n).astype(np.uint32)
    o = oracle.SpladeIndex(off, tok, w, id_rank=rank)
    h = S.HipSpladeIndex.build_from_csr(None, off, tok, w, id_rank=rank)
    for b in (1, 7, 33, 64):
        qs = [sc.query(rng, 1200, int(rng.integers(1, 150)), absent=2) for _ in range(b)]
        if b >= 7:
            qs[2] = (np.zeros(0, np.uint32), np.zeros(0, np.float32))
            qs[5] = (np.array([5000, 5001], np.uint32), np.ones(2, np.float32))
        for keep in (None, rng.random(n) < 0.3):
            ch, scs, cnt, rc = h.search_batch_raw(qs, 200, keep)
            assert rc == 0
            for i, (qt, qw) in enumerate(qs):
                oc, os_ = o.search_raw(qt, qw, 200, keep)
                assert cnt[i] == oc.size
                assert np.array_equal(ch[i, :cnt[i]], oc) and np.array_equal(scs[i, :cnt[i]].view(np.uint32), os_.view(np.uint32))
                if i % 9 == 0:
                    hc, hs, _ = h.search_raw(qt, qw, 200, keep)
                    assert np.array_equal(hc, oc) and np.array_equal(hs.view(np.uint32), os_.view(np.uint32))
    qs = [sc.query(rng, 1200, 5) for _ in range(65)]
    assert h.search_batch_raw(qs, 10)[3] == -1 and "64" in h.last_error
    h.close()


def test_concurrent_single_query_callers_share_batches(S, oracle):
    """The combining queue of `cqs_hip_sparse_index_search`: eight threads on one handle - every answer the oracle's, bit
    for bit, whatever batch it rode in; the queue's counters show shared batches; mixed k, a filtered caller (serial path) and
    a caller with an invalid weight (refused alone, nobody else disturbed) in the mix."""
    import threading
    rng = np.random.default_rng(31)
    n = 40000
    off, tok, w = sc.corpus(rng, n, 1500, 8, 60)
    o = oracle.SpladeIndex(off, tok, w)
    h = S.HipSpladeIndex.build_from_csr(None, off, tok, w)
    queries = [sc.query(rng, 1500, int(rng.integers(5, 90))) for _ in range(24)]
    keep = rng.random(n) < 0.5
    want = {}
    for i, (qt, qw) in enumerate(queries):
        for k in (100, 500):
            want[(i, k, False)] = o.search_raw(qt, qw, k)
        want[(i, 100, True)] = o.search_raw(qt, qw, 100, keep)
    bad = []

    def work(t):
        for rep in range(30):
            i = (7 * t + rep) % len(queries)
            qt, qw = queries[i]
            k = 500 if (t + rep) % 3 == 0 else 100
            flt = t == 5 and rep % 4 == 0
            if t == 6 and rep % 10 == 0:                          # reserved NaN payload: refused, alone
                qbad = qw.copy(); qbad.view(np.uint32)[0] = 0xFFFFFFFF
                if h.search_raw(qt, qbad, k)[2] != -1:
                    bad.append(("not refused", t, rep))
                continue
            c, s_, rc = h.search_raw(qt, qw, 100 if flt else k, keep if flt else None)
            oc, os_ = want[(i, 100 if flt else k, flt)]
            if rc != 0 or not np.array_equal(c, oc) or not np.array_equal(s_.view(np.uint32), os_.view(np.uint32)):
                bad.append((t, rep, rc))

    p0, q0 = h.combine_stats()
    th = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    [t.start() for t in th]; [t.join() for t in th]
    p1, q1 = h.combine_stats()
    assert not bad, bad[:5]
    assert q1 - q0 >= 200 and p1 - p0 <= q1 - q0
    h.close()


def test_nan_payloads_next_to_the_marker(S, oracle):
    """The kernel marks "not scored yet" with the bit pattern 0xFFFFFFFF (refused as an input).  A NaN weight with every
    other payload - 0x7FFFFFFF included, times a negative query weight - must behave like any NaN: the chunk's sum stays NaN
    through later adds and the chunk is dropped (`would_accept`), never mistaken for an unscored one that starts over."""
    f = lambda bits: float(np.array([bits], dtype=np.uint32).view(np.float32)[0])
    w = np.array([f(0x7FFFFFFF), 2.0, 0.5, 1.0, f(0xFFFFFFFE), 3.0], dtype=np.float32)
    w.view(np.uint32)[0] = 0x7FFFFFFF
    w.view(np.uint32)[4] = 0xFFFFFFFE
    off = np.array([0, 2, 4, 6], dtype=np.uint64)
    tok = np.array([1, 2, 1, 2, 1, 2], dtype=np.uint32)
    o = oracle.SpladeIndex(off, tok, w)
    h = S.HipSpladeIndex.build_from_csr(None, off, tok, w)
    for qw in ([-1.0, 1.0], [1.0, 1.0], [-2.5, -1.0]):
        qt = np.array([1, 2], np.uint32)
        oc, os_ = o.search_raw(qt, np.array(qw, np.float32), 10)
        hc, hs, rc = h.search_raw(qt, np.array(qw, np.float32), 10)
        assert rc == 0 and np.array_equal(hc, oc) and np.array_equal(hs.view(np.uint32), os_.view(np.uint32))
        assert list(hc) == [1]                                 # only the chunk without a NaN weight survives
    h.close()


def test_persistence_round_trip_and_rejections(S, oracle, tmp_path):
    """save / load / load_or_build in the library's own format, after the reference's persistence tests (index.rs:1243-1375,
    :1646-1700): a round trip answers bit for bit like the index it came from (ranked and unranked); another generation, a
    bad magic, a flipped body byte, a truncated file and a missing file all load as "rebuild"; load_or_build persists on the
    first call and loads on the second; a save over a live file replaces it atomically."""
    rng = np.random.default_rng(612)
    n = 7000
    off, tok, w = sc.corpus(rng, n, 900, 5, 50, dup_frac=0.2, special=True)
    ids = ["k%05d" % i for i in rng.permutation(n)]
    o = oracle.SpladeIndex(off, tok, w, ids=ids)
    a = S.HipSpladeIndex.build_from_csr(ids, off, tok, w)
    path = tmp_path / "splade.hip.bin"
    ck = a.save(path, generation=41)
    assert ck != 0 and path.exists() and not (tmp_path / "splade.hip.bin.tmp").exists()
    b = S.HipSpladeIndex.load(path, 41, ids)
    assert b is not None and len(b) == n and b.unique_tokens() == a.unique_tokens() and b.postings() == a.postings()
    for _ in range(3):
        qt, qw = sc.query(rng, 900, 40)
        keep = rng.random(n) < 0.5
        for kp in (None, keep):
            bc, bs, rc = b.search_raw(qt, qw, 300, kp)
            oc, os_ = o.search_raw(qt, qw, 300, kp)
            assert rc == 0 and np.array_equal(bc, oc) and np.array_equal(bs.view(np.uint32), os_.view(np.uint32))
    b.close()
    assert S.HipSpladeIndex.load(path, 42, ids) is None                          # generation mismatch (index.rs:1279-1297)
    assert S.HipSpladeIndex.load(path, 41, ids[:-1]) is None                     # another chunk count
    assert S.HipSpladeIndex.load(tmp_path / "nothing.bin", 41, ids) is None      # missing file (index.rs:1333-1338)
    raw = path.read_bytes()
    (tmp_path / "magic.bin").write_bytes(b"NOTASPDX" + raw[8:])
    assert S.HipSpladeIndex.load(tmp_path / "magic.bin", 41, ids) is None        # bad magic (index.rs:1300-1310)
    flipped = bytearray(raw); flipped[len(raw) // 2] ^= 0x40
    (tmp_path / "corrupt.bin").write_bytes(bytes(flipped))
    assert S.HipSpladeIndex.load(tmp_path / "corrupt.bin", 41, ids) is None      # corrupt body (index.rs:1313-1330)
    (tmp_path / "short.bin").write_bytes(raw[:-16])
    assert S.HipSpladeIndex.load(tmp_path / "short.bin", 41, ids) is None        # truncated
    # unranked index (integer ids) round trip
    u = S.HipSpladeIndex.build_from_csr(None, off, tok, w)
    u.save(tmp_path / "u.bin", 1)
    u2 = S.HipSpladeIndex.load(tmp_path / "u.bin", 1)
    qt, qw = sc.query(rng, 900, 30)
    assert all(np.array_equal(x, y) for x, y in zip(u.search_raw(qt, qw, 100)[:2], u2.search_raw(qt, qw, 100)[:2]))
    u.close(); u2.close()
    # load_or_build (index.rs:1341-1375): first call builds + persists, second call loads
    chunks = [(ids[i], [(int(tok[e]), float(w[e])) for e in range(int(off[i]), int(off[i + 1]))]) for i in range(300)]
    calls = []
    def rows():
        calls.append(1)
        return chunks
    rows.ids = [c for c, _ in chunks]
    p2 = tmp_path / "lob.bin"
    i1, rebuilt1 = S.HipSpladeIndex.load_or_build(p2, 7, rows)
    i2, rebuilt2 = S.HipSpladeIndex.load_or_build(p2, 7, rows)
    i3, rebuilt3 = S.HipSpladeIndex.load_or_build(p2, 8, rows)                   # the store moved on: rebuild + replace
    assert (rebuilt1, rebuilt2, rebuilt3) == (True, False, True) and len(calls) == 2
    q = [(int(tok[0]), 1.0), (int(tok[5]), 0.5)]
    r1 = [(r.id, r.score) for r in i1.search(q, 50)]
    assert r1 == [(r.id, r.score) for r in i2.search(q, 50)] == [(r.id, r.score) for r in i3.search(q, 50)] and r1
    assert S.HipSpladeIndex.load(p2, 8, rows.ids) is not None and S.HipSpladeIndex.load(p2, 7, rows.ids) is None   # atomic replace
    for x in (a, i1, i2, i3):
        x.close()


def _forge_sparse_file(path, chunks, tok, off, post, rank=None, generation=1, fix_checksum=True):
    """A file in the library's sparse-index format with a VALID checksum (the multiply-rotate hash of persist_util.h restated
    here), whatever its sections hold - to reach the loader's structure checks behind the checksum."""
    import struct
    M = (1 << 64) - 1
    P1, P2 = 0x9E3779B185EBCA87, 0xC2B2AE3D27D4EB4F

    def pad8(b):
        return b + b"\0" * (-len(b) % 8)

    secs = [pad8(np.asarray(tok, np.uint32).tobytes()), pad8(np.asarray(off, np.uint64).tobytes()),
            pad8(np.asarray(post, np.uint32).reshape(-1, 2).tobytes() if len(post) else b""),
            pad8(np.asarray(rank, np.uint32).tobytes()) if rank is not None else b""]
    body = b"".join(secs)
    h = 0x27D4EB2F165667C5 ^ len(body)
    for (w,) in struct.iter_unpack("<Q", body):
        h ^= (w * P1) & M
        h = ((((h << 31) | (h >> 33)) & M) * P2) & M
    h ^= h >> 29; h = (h * P2) & M; h ^= h >> 32
    if not fix_checksum:
        h ^= 1
    header = struct.pack("<8sIIQQQQQ8s", b"CQSHIPS1", 1, 1 if rank is not None else 0, chunks, len(tok), len(post), generation, h, b"\0" * 8)
    assert len(header) == 64
    open(path, "wb").write(header + body)


def test_loader_rejects_forged_structure(S, tmp_path):
    """The loader is handed a file from disk: behind a matching checksum it still re-checks everything the kernels rely on
    (they index LDS by posting position).  A well-formed forged file loads and answers; each single defect - a position past
    the end, a descending list, offsets that do not tile the postings, a repeated token, an id order that is not a
    permutation, the reserved weight pattern - makes the load answer "rebuild", never a handle."""
    one = np.float32(1.0).view(np.uint32).item()
    good = dict(chunks=6, tok=[3, 9], off=[0, 3, 5], post=[(0, one), (2, one), (5, one), (1, one), (2, one)])
    p = tmp_path / "ok.bin"
    _forge_sparse_file(p, **good)
    h = S.HipSpladeIndex.load(p, 1)
    assert h is not None and len(h) == 6 and h.unique_tokens() == 2 and h.postings() == 5
    c, s_, rc = h.search_raw([3, 9], [1.0, 2.0], 10)
    assert rc == 0 and list(c) == [2, 1, 0, 5] and list(s_) == [3.0, 2.0, 1.0, 1.0]
    h.close()
    _forge_sparse_file(p, **good, rank=[5, 4, 3, 2, 1, 0])
    h = S.HipSpladeIndex.load(p, 1)
    assert h is not None and list(h.search_raw([3], [1.0], 10)[0]) == [5, 3, 0]     # positions 0, 2, 5 are chunks 5, 3, 0; ties by position
    h.close()
    bad = {
        "position past the end": dict(good, post=[(0, one), (2, one), (6, one), (1, one), (2, one)]),
        "descending list": dict(good, post=[(2, one), (0, one), (5, one), (1, one), (2, one)]),
        "offsets do not tile": dict(good, off=[0, 4, 4]),
        "offsets run past the postings": dict(good, off=[0, 6, 5]),
        "repeated token": dict(good, tok=[3, 3]),
        "descending tokens": dict(good, tok=[9, 3]),
        "reserved weight pattern": dict(good, post=[(0, one), (2, 0xFFFFFFFF), (5, one), (1, one), (2, one)]),
    }
    for name, spec in bad.items():
        _forge_sparse_file(p, **spec)
        assert S.HipSpladeIndex.load(p, 1) is None, name
    _forge_sparse_file(p, **good, rank=[0, 1, 2, 3, 4, 4])
    assert S.HipSpladeIndex.load(p, 1) is None                                       # not a permutation
    _forge_sparse_file(p, **good, rank=[0, 1, 2, 3, 4, 9])
    assert S.HipSpladeIndex.load(p, 1) is None
    _forge_sparse_file(p, **good, fix_checksum=False)
    assert S.HipSpladeIndex.load(p, 1) is None                                       # and the checksum itself
    _forge_sparse_file(p, **good)
    assert S.HipSpladeIndex.load(p, 1) is not None                                   # (the forger is sound)
