"""Pins the CPU oracle against the known-answer tests the reference's own suite holds
for this path.  Each test names the reference test it replays
(file:line in the cqs repo, v1.51.0).  CPU only."""
import math

import numpy as np
import pytest

from cqs_amd import synth

DIM = 768
NAN, INF = float("nan"), float("inf")


# ---- src/math.rs:69-298 --------------------------------------------------------
def test_cosine_identical(oracle):  # math.rs:83-88
    a = np.full(DIM, 0.5, np.float32)
    assert oracle.cosine_similarity(a, a) > 0.99


def test_cosine_orthogonal(oracle):  # math.rs:90-96
    a = np.zeros(DIM, np.float32); a[0] = 1
    b = np.zeros(DIM, np.float32); b[1] = 1
    assert abs(oracle.cosine_similarity(a, b)) < 0.01


def test_cosine_symmetric(oracle):  # math.rs:98-109
    i = np.arange(DIM, dtype=np.float32)
    a, b = i / DIM, 1.0 - i / DIM
    assert abs(oracle.cosine_similarity(a, b) - oracle.cosine_similarity(b, a)) < 1e-6


def test_cosine_range_finite(oracle):  # math.rs:111-121
    i = np.arange(DIM)
    a = ((i * 7) % 100).astype(np.float32) / 100.0
    b = ((i * 13) % 100).astype(np.float32) / 100.0
    assert math.isfinite(oracle.cosine_similarity(a, b))


def test_cosine_dimension_mismatch(oracle):  # math.rs:123-135
    a = np.full(100, 0.5, np.float32)
    b = np.full(DIM, 0.5, np.float32)
    assert oracle.cosine_similarity(a, b) is None
    assert oracle.cosine_similarity(a, a) is not None
    assert oracle.cosine_similarity(np.zeros(0, np.float32), np.zeros(0, np.float32)) is None  # math.rs:12


def test_cosine_nan_inf(oracle):  # math.rs:138-183
    normal = np.full(DIM, 0.5, np.float32)
    nan = np.full(DIM, NAN, np.float32)
    assert oracle.cosine_similarity(nan, normal) is None
    assert oracle.cosine_similarity(normal, nan) is None
    inf = normal.copy(); inf[42] = INF
    assert oracle.cosine_similarity(inf, normal) is None
    ninf = normal.copy(); ninf[0] = -INF
    assert oracle.cosine_similarity(ninf, normal) is None
    z = oracle.cosine_similarity(np.zeros(DIM, np.float32), normal)
    assert z is None or math.isfinite(z)
    sub = np.full(DIM, np.finfo(np.float32).tiny / 2, np.float32)
    s = oracle.cosine_similarity(sub, sub)
    assert s is None or math.isfinite(s)


def test_full_cosine(oracle):  # math.rs:218-296
    assert abs(oracle.full_cosine_similarity([1, 2, 3], [4, 5, 6]) - 0.9746) < 0.001
    assert abs(oracle.full_cosine_similarity([1, 0, 0], [0, 1, 0])) < 1e-6
    assert abs(oracle.full_cosine_similarity([3, 4, 5], [3, 4, 5]) - 1.0) < 1e-6
    assert oracle.full_cosine_similarity([0, 0, 0], [1, 2, 3]) is None
    assert oracle.full_cosine_similarity([1, 2, 3], [0, 0, 0]) is None
    assert oracle.full_cosine_similarity([0, 0, 0], [0, 0, 0]) is None
    assert oracle.full_cosine_similarity([NAN, 1, 2], [1, 2, 3]) is None
    assert oracle.full_cosine_similarity([1, 2, 3], [1, 2]) is None


def test_dot_kinds_agree(oracle):
    """simsimd-style, f64 fallback and sequential f32 agree to ~1e-6 on unit vectors (SURVEY §8a caveat 4)."""
    x = synth.gaussian_unit(64, seed=7)
    for i in range(0, 64, 2):
        d0 = oracle.dot(x[i], x[i + 1], 0)
        d1 = oracle.dot(x[i], x[i + 1], 1)
        d2 = oracle.dot(x[i], x[i + 1], 2)
        assert abs(d0 - d1) < 2e-6 and abs(d2 - d1) < 2e-6
    # tail handling (n % 8 != 0) against f64
    a = np.arange(1, 14, dtype=np.float32) / 13
    b = np.arange(13, 0, -1, dtype=np.float32) / 7
    assert abs(oracle.dot(a, b, 0) - float(np.dot(a.astype(np.float64), b.astype(np.float64)))) < 1e-5


def test_dot_body_is_chosen_by_cpu_features_and_bit_identical(oracle):
    """The timed CPU baseline runs the vector body wherever the CPU has AVX2 + FMA (feature bits, not the
    CPU model: round 1's `target_clones("arch=haswell")` fell to one `fmaf` call per element everywhere but
    on Intel Haswell), and both bodies give bit-identical dots, so the KATs do not depend on the host."""
    flags = set()
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    flags = set(line.split(":", 1)[1].split())
                    break
    except OSError:
        pass
    isa = oracle.dot_isa()
    if {"avx2", "fma"} <= flags:
        assert isa == "avx2+fma", isa
        assert oracle.dot_isa(native=True) == ("avx512f" if "avx512f" in flags else "avx2+fma")
    else:
        assert isa == "scalar-fmaf"
    rng = np.random.default_rng(11)
    try:
        for n in (1, 7, 8, 9, 15, 16, 17, 31, 100, 768, 769, 2048):
            a = rng.standard_normal(n).astype(np.float32)
            b = rng.standard_normal(n).astype(np.float32)
            oracle.dot_force_scalar(False)
            v_vec, v_nat = oracle.dot(a, b, 0), oracle.dot(a, b, 3)
            oracle.dot_force_scalar(True)
            assert oracle.dot_isa() == "scalar-fmaf"
            assert oracle.dot(a, b, 0) == v_vec, n              # bit-identical
            ref = float(np.dot(a.astype(np.float64), b.astype(np.float64)))
            assert abs(v_nat - ref) <= 2e-6 * max(1.0, float(np.sum(np.abs(a * b)))), n
    finally:
        oracle.dot_force_scalar(False)


# ---- src/embedder/core.rs normalize_l2 KATs ---------------------------------------
def test_normalize_l2_kats(oracle):
    v = oracle.normalize_l2([3.0, 4.0])  # core.rs:1484-1494
    assert abs(v[0] - 0.6) < 1e-6 and abs(v[1] - 0.8) < 1e-6
    assert abs(math.sqrt(float(np.sum(v * v))) - 1.0) < 1e-6
    assert list(oracle.normalize_l2([0.0, 0.0, 0.0])) == [0.0, 0.0, 0.0]  # core.rs:1496-1501
    assert oracle.normalize_l2(np.zeros(0, np.float32)).size == 0  # core.rs:1503-1507
    v = oracle.normalize_l2([1.0, NAN, 3.0])  # core.rs:1518-1531: passes through verbatim
    assert v[0] == 1.0 and math.isnan(v[1]) and v[2] == 3.0
    assert np.all(np.isnan(oracle.normalize_l2([NAN] * 4)))  # core.rs:1533-1542
    assert np.all(np.isnan(oracle.normalize_l2([INF, INF, INF])))  # core.rs:1544-1556


def test_normalize_l2_properties(oracle):
    """proptests core.rs:1830 (self-dot within 1±1e-3) and :1853 (idempotent 1e-5)."""
    rng = np.random.default_rng(0)
    for _ in range(50):
        v = rng.standard_normal(rng.integers(1, 1025)).astype(np.float32) * np.float32(rng.uniform(0.01, 100))
        u = oracle.normalize_l2(v)
        assert abs(float(np.dot(u.astype(np.float64), u.astype(np.float64))) - 1.0) < 1e-3
        assert np.max(np.abs(oracle.normalize_l2(u) - u)) < 1e-5


def test_synth_normalize_matches_oracle(oracle):
    """The numpy generator's normalisation is the reference formula up to f32 summation order."""
    x = synth.gaussian_unit(8, seed=3)
    for r in x:
        assert abs(float(np.dot(r.astype(np.float64), r.astype(np.float64))) - 1.0) < 1e-5


# ---- pooling KATs (core.rs:1643-1712) -------------------------------------------------
def test_pooling_kats(oracle):
    h = np.array([[[1, 2], [3, 4], [100, 200]]], np.float32)
    p = oracle.mean_pool(h, np.array([[1, 1, 0]]))
    assert abs(p[0, 0] - 2.0) < 1e-6 and abs(p[0, 1] - 3.0) < 1e-6
    p = oracle.mean_pool(np.array([[[5, 5], [6, 6]]], np.float32), np.array([[0, 0]]))
    assert list(p[0]) == [0.0, 0.0]
    h = np.array([[[1, 2], [9.9, 9.9]], [[3, 4], [7.7, 7.7]]], np.float32)
    p = oracle.cls_pool(h)
    assert list(p[0]) == [1.0, 2.0] and list(p[1]) == [3.0, 4.0]
    h = np.array([[[0, 0], [0, 0], [42, 43], [9, 9]], [[11, 12], [0, 0], [0, 0], [0, 0]]], np.float32)
    p = oracle.last_token_pool(h, np.array([[1, 1, 1, 0], [1, 0, 0, 0]]))
    assert list(p[0]) == [42.0, 43.0] and list(p[1]) == [11.0, 12.0]
    p = oracle.last_token_pool(np.array([[[7, 8], [9, 10]]], np.float32), np.array([[0, 0]]))
    assert list(p[0]) == [7.0, 8.0]


# ---- BoundedScoreHeap KATs (candidate.rs:585-707) --------------------------------------
def test_heap_equal_scores(oracle):  # candidate.rs:588-600
    h = oracle.BoundedScoreHeap(2)
    for s in "abc":
        h.push(s, 0.5)
    r = h.into_sorted_vec()
    assert len(r) == 2 and {x[0] for x in r} == {"a", "b"}


def test_heap_evicts_lowest(oracle):  # candidate.rs:602-612
    h = oracle.BoundedScoreHeap(2)
    h.push("low", 0.1); h.push("mid", 0.5); h.push("high", 0.9)
    r = h.into_sorted_vec()
    assert [x[0] for x in r] == ["high", "mid"]


def test_heap_ignores_non_finite(oracle):  # candidate.rs:614-624
    h = oracle.BoundedScoreHeap(5)
    h.push("nan", NAN); h.push("inf", INF); h.push("neginf", -INF); h.push("ok", 0.5)
    r = h.into_sorted_vec()
    assert [x[0] for x in r] == ["ok"]


def test_heap_empty(oracle):  # candidate.rs:626-630
    assert oracle.BoundedScoreHeap(5).into_sorted_vec() == []


def test_heap_deterministic_reverse_and_forward(oracle):  # candidate.rs:632-665
    h = oracle.BoundedScoreHeap(2)
    for s in "cba":
        h.push(s, 0.5)
    assert [x[0] for x in h.into_sorted_vec()] == ["a", "b"]
    h = oracle.BoundedScoreHeap(2)
    for s in "abc":
        h.push(s, 0.5)
    assert [x[0] for x in h.into_sorted_vec()] == ["a", "b"]


def test_heap_would_accept(oracle):  # candidate.rs:677-707
    h = oracle.BoundedScoreHeap(2)
    assert h.would_accept(0.1)
    h.push("a", 0.5)
    assert h.would_accept(-1.0)
    assert not h.would_accept(NAN)
    h.push("b", 0.9)
    assert h.would_accept(0.7)
    assert not h.would_accept(0.1)
    assert h.would_accept(0.5)
    assert not oracle.BoundedScoreHeap(0).would_accept(1.0)


def test_heap_capacity_zero_and_string_order(oracle):
    h = oracle.BoundedScoreHeap(0)
    h.push("x", 1.0)
    assert h.into_sorted_vec() == []
    # ids compare as UTF-8 bytes like Rust String: "a:10" < "a:9"
    h = oracle.BoundedScoreHeap(1)
    h.push("a:9", 0.5); h.push("a:10", 0.5)
    assert [x[0] for x in h.into_sorted_vec()] == ["a:10"]


# ---- scoring pipeline default (candidate.rs:538-562, 506-520) -----------------------
def test_apply_scoring_default(oracle):
    assert oracle.apply_scoring_default(0.7, 0.3) == pytest.approx(0.7)
    assert oracle.apply_scoring_default(-0.2, 0.0) == 0.0       # clamp to 0, `>=` passes at threshold 0
    assert oracle.apply_scoring_default(1.0000005, 0.0) == 1.0  # clamp to 1
    assert oracle.apply_scoring_default(0.29, 0.3) is None
    assert oracle.apply_scoring_default(0.3, 0.3) is not None   # `>=`
    assert oracle.apply_scoring_default(NAN, 0.0) is None       # NaN >= t is false


# ---- blobs (helpers/embeddings.rs:59-155) ------------------------------------------------
def test_blob_roundtrip_and_mismatch(oracle):
    data = np.zeros(DIM, np.float32)
    assert oracle.bytes_to_embedding(data.tobytes(), DIM) is not None
    assert oracle.bytes_to_embedding(np.ones(1024, np.float32).tobytes(), 1024) is not None
    assert oracle.bytes_to_embedding(data.tobytes(), 1024) is None  # EmbeddingBlobMismatch
    v = np.full(DIM, 0.5, np.float32); v[0] = NAN
    r = oracle.bytes_to_embedding(v.tobytes(), DIM)  # NaN passes through (embeddings.rs:120-155)
    assert math.isnan(r[0]) and r[1] == 0.5


# ---- DistDotClamped (hnsw/mod.rs:1026-1049), CAGRA score (cagra.rs:656-661) ------------------
def test_dist_dot_clamped(oracle):
    L = oracle.lib()
    a = np.array([1.0, 0.0007], np.float32)
    d = L.cqs_oracle_dist_dot_clamped(a.ctypes.data, a.ctypes.data, 2)
    assert d >= 0.0 and abs(d) < 1e-6
    x = np.array([0.6, 0.0], np.float32); y = np.array([0.5, 0.0], np.float32)
    assert abs(L.cqs_oracle_dist_dot_clamped(x.ctypes.data, y.ctypes.data, 2) - 0.70) < 1e-6
    p = np.array([1.0, 0.0], np.float32); q = np.array([0.0, 1.0], np.float32)
    assert abs(L.cqs_oracle_dist_dot_clamped(p.ctypes.data, q.ctypes.data, 2) - 1.0) < 1e-6
    assert L.cqs_oracle_cagra_cosine_from_l2sq(0.0) == 1.0
    assert L.cqs_oracle_cagra_cosine_from_l2sq(-1e-6) == 1.0  # .min(1.0)
    assert abs(L.cqs_oracle_cagra_cosine_from_l2sq(2.0)) < 1e-7


# ---- limits (limits.rs:913-956), embed_batch_size (models.rs:1434-1482) ------------------------
def test_dim_scaled_batch_table(oracle):
    f = oracle.lib().cqs_oracle_dim_scaled_batch
    assert f(10_000, 1024, 500, 50_000) == 10_000
    assert f(5_000, 1024, 500, 50_000) == 5_000
    assert f(10_000, 2048, 500, 50_000) == 5_000
    assert f(10_000, 4096, 500, 50_000) == 2_500
    assert f(10_000, 768, 500, 50_000) == 13_333
    assert f(10_000, 65_536, 500, 50_000) == 500
    assert f(10_000, 64, 500, 50_000) == 50_000
    assert f(10_000, 0, 500, 50_000) == 10_000
    assert f(50, 0, 500, 50_000) == 500
    assert f(99_999, 0, 500, 50_000) == 50_000
    assert f(5000, 768, 500, 50_000) == 6666  # brute-force batch (search/query.rs:426-432)
    from cqs_amd.index import dim_scaled_batch
    for args in [(10_000, 768, 500, 50_000), (10_000, 0, 500, 50_000), (10_000, 64, 500, 50_000)]:
        assert dim_scaled_batch(*args) == f(*args)


def test_candidate_count_for(oracle):  # limits.rs:315-320, :905-908
    f = oracle.lib().cqs_oracle_candidate_count_for
    assert f(20, 500) == 500
    assert f(200, 500) == 1000
    assert f((2**64 - 1) // 4, 500) == 2**64 - 1


def test_embed_batch_size(oracle):
    f = oracle.lib().cqs_oracle_embed_batch_size
    assert f(1024, 512) == 64     # bge-large (models.rs:1434-1439)
    assert f(768, 512) == 128     # e5-base (models.rs:1441-1451)
    assert f(768, 2048) == 32     # gemma / nomic shape (models.rs:1453-1481)


# ---- prepare_index_data skip rule (hnsw/mod.rs:717-731) ----------------------------------------
def test_prepare_index_keep(oracle):
    rows = synth.gaussian_unit(6, 16, seed=1)
    rows[1] = 0.0
    rows[3, 5] = NAN
    rows[4, 0] = INF
    keep, kept = oracle.prepare_index_keep(rows)
    assert list(keep) == [True, False, True, False, False, True] and kept == 3
    from cqs_amd import prepare_index_data
    ids, flat, n = prepare_index_data([(str(i), rows[i]) for i in range(6)], 16)
    assert ids == ["0", "2", "5"] and n == 3 and flat.shape == (3, 16)
    with pytest.raises(ValueError):
        prepare_index_data([], 16)
    with pytest.raises(ValueError):
        prepare_index_data([("a", rows[0][:8])], 16)
    with pytest.raises(ValueError):
        prepare_index_data([("z", np.zeros(16, np.float32))], 16)


# ---- brute force / index search / neighbors on the reference's generators -----------------------
def test_brute_force_matches_numpy_f64(oracle):
    rows = synth.sin_corpus(300)
    q = synth.sin_embedding(17)
    ids, sc = oracle.brute_force(rows, q, 10, 0.0)
    exact = rows.astype(np.float64) @ q.astype(np.float64)
    assert ids[0] == 17 and abs(sc[0] - 1.0) < 1e-5
    assert np.max(np.abs(sc - exact[ids.astype(int)])) < 1e-5
    assert np.all(np.diff(sc) <= 0)


def test_brute_force_clamp_ties_by_id(oracle):
    """SURVEY §8a caveat 2: all-negative cosines clamp to 0.0 and tie -> ordered by id asc."""
    rows = -synth.gaussian_unit(50, seed=5) ** 2  # all components negative
    rows = rows / np.linalg.norm(rows, axis=1, keepdims=True)
    q = np.abs(synth.gaussian_unit(1, seed=6)[0])
    ids, sc = oracle.brute_force(rows.astype(np.float32), q, 5, 0.0)
    assert list(ids) == [0, 1, 2, 3, 4] and np.all(sc == 0.0)
    ids, sc = oracle.brute_force(rows.astype(np.float32), q, 5, 0.3)  # threshold gate drops all
    assert len(ids) == 0


def test_brute_force_dim_mismatch_and_nan_row(oracle):
    rows = synth.gaussian_unit(20, seed=2)
    rows[7, 3] = NAN
    q = rows[0].copy()
    ids, _ = oracle.brute_force(rows, q, 20, 0.0)
    assert 7 not in ids  # cosine -> None -> skipped (math.rs:23-27)
    assert len(oracle.brute_force(rows, q[:100], 5, 0.0)[0]) == 0


def test_index_search_guards(oracle):  # cagra.rs:443-470
    rows = synth.gaussian_unit(40, seed=9)
    q = rows[3]
    assert len(oracle.index_search(rows, q, 0)[0]) == 0
    assert len(oracle.index_search(rows[:0], q, 5)[0]) == 0
    assert len(oracle.index_search(rows, q[:10], 5)[0]) == 0
    bad = q.copy(); bad[0] = NAN
    assert len(oracle.index_search(rows, bad, 5)[0]) == 0
    ids, sc = oracle.index_search(rows, q, 5)
    assert ids[0] == 3 and np.all(np.diff(sc) <= 0)
    # bitset: none -> empty; all -> unfiltered; k capped at included (cagra.rs:760-775)
    n_words = (40 + 31) // 32
    assert len(oracle.index_search(rows, q, 5, np.zeros(n_words, np.uint32))[0]) == 0
    allb = np.full(n_words, 0xFFFFFFFF, np.uint32)
    assert list(oracle.index_search(rows, q, 5, allb)[0]) == list(ids)
    some = np.zeros(n_words, np.uint32); some[0] = (1 << 3) | (1 << 9)
    ids2, _ = oracle.index_search(rows, q, 5, some)
    assert len(ids2) == 2 and set(ids2) == {3, 9}


def test_find_neighbors(oracle):  # neighbors.rs:86-132
    rows = synth.sin_corpus(50)  # sin(seed*0.1 + ...) repeats every ~62.8 seeds: stay inside one period
    ids, sc = oracle.find_neighbors(rows, 20, 5)
    assert 20 not in ids and len(ids) == 5 and np.all(np.diff(sc) <= 0)
    assert set(ids[:2]) == {19, 21}
    assert len(oracle.find_neighbors(synth.sin_corpus(150), 20, 1000)[0]) == 100  # SIMILAR_LIMIT_MAX clamp
    assert len(oracle.find_neighbors(rows, 20, 0)[0]) == 1       # clamp(1, ..)


def test_guided_equals_brute_set(oracle):
    """tests/search_test.rs:344-390: index-guided top-k == brute-force top-k as a set (exact backend: same order)."""
    rows = synth.gaussian_unit(500, seed=11)
    q = synth.gaussian_unit(1, seed=12)[0]
    a, sa = oracle.index_search(rows, q, 20)
    b, sb = oracle.brute_force(rows, q, 20, 0.0)
    pos = sa > 0
    assert list(a[pos]) == list(b[: pos.sum()])


def test_brute_force_mt_equals_single(oracle):
    rows = synth.gaussian_unit(3000, seed=21)
    q = synth.gaussian_unit(1, seed=22)[0]
    a, sa = oracle.brute_force(rows, q, 20, 0.0)
    b, sb = oracle.brute_force_mt(rows, q, 20, 0.0, 4)
    assert list(a) == list(b) and np.array_equal(sa, sb)


# ---- tests/golden/reference_kats.json: the tabular KATs of the reference suite as data -------------------
def _golden():
    import json, os
    with open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")) as f:
        return json.load(f)


def _num(x):
    return {"nan": NAN, "inf": INF, "-inf": -INF}.get(x, x) if isinstance(x, str) else x


def test_golden_fixture_float_functions(oracle):
    g = _golden()
    for c in g["full_cosine_similarity"]["cases"]:
        got = oracle.full_cosine_similarity([_num(v) for v in c["a"]], [_num(v) for v in c["b"]])
        if c["expect"] is None:
            assert got is None, c
        else:
            assert got is not None and abs(got - c["expect"]) < c["tol"], (c, got)
    for c in g["normalize_l2"]["cases"]:
        got = oracle.normalize_l2(np.array([_num(v) for v in c["in"]], np.float32))
        want = [_num(v) for v in c["expect"]]
        assert len(got) == len(want)
        for a, b in zip(got, want):
            assert (math.isnan(a) and math.isnan(b)) or abs(a - b) <= c["tol"], (c, list(got))
    L = oracle.lib()
    for c in g["dist_dot_clamped"]["cases"]:
        a, b = np.array(c["a"], np.float32), np.array(c["b"], np.float32)
        d = L.cqs_oracle_dist_dot_clamped(a.ctypes.data, b.ctypes.data, len(a))
        assert d >= 0.0 and abs(d - c["expect"]) < c["tol"], (c, d)


def test_golden_fixture_heap_and_tables(oracle):
    g = _golden()
    for c in g["bounded_score_heap"]["cases"]:
        h = oracle.BoundedScoreHeap(c["capacity"])
        for name, score in c["push"]:
            h.push(name, _num(score))
        assert [x[0] for x in h.into_sorted_vec()] == c["expect_ids"], c
    L = oracle.lib()
    for args, want in g["dim_scaled_batch"]["cases"]:
        assert L.cqs_oracle_dim_scaled_batch(*args) == want, args
    for args, want in g["candidate_count_for"]["cases"]:
        assert L.cqs_oracle_candidate_count_for(*args) == want, args
    for args, want in g["embed_batch_size"]["cases"]:
        assert L.cqs_oracle_embed_batch_size(*args) == want, args
