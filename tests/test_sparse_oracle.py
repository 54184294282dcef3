"""The sparse leg's CPU oracle (`oracle.SpladeIndex`, restating src/splade/index.rs:177-290) against the known-answer
cases of the reference's own unit tests (tests/golden/splade_index_kats.json), plus the host-side fusion mirror.  CPU only."""
import math

import numpy as np

import sparse_cases as sc


def _check_case(ix, case):
    q = sc.kat_query(case["query"])
    if "filter_equals" in case:
        res = ix.search_with_filter(q, case["k"], lambda cid, want=case["filter_equals"]: cid == want)
    else:
        res = ix.search(q, case["k"])
    ids = [r[0] if isinstance(r, tuple) else r.id for r in res]
    scores = [r[1] if isinstance(r, tuple) else r.score for r in res]
    if "ids" in case:
        assert ids == case["ids"], case["name"]
    if "len" in case:
        assert len(ids) == case["len"], case["name"]
    if "scores" in case:
        for got, want in zip(scores, case["scores"]):
            assert abs(got - want) < case["tolerance"], case["name"]
    if "ids_subset_of" in case:
        assert set(ids) <= set(case["ids_subset_of"]), case["name"]


def test_oracle_replays_the_reference_kats(oracle):   # src/splade/index.rs:1113-1242
    k = sc.kats()
    ix = oracle.SpladeIndex.build([(cid, [(int(t), float(w)) for t, w in sv]) for cid, sv in k["index"]])
    assert len(ix) == k["len"] and not ix.is_empty()
    for case in k["cases"]:
        _check_case(ix, case)
    empty = oracle.SpladeIndex.build([])                  # test_build_empty (index.rs:1122-1127)
    assert empty.is_empty() and empty.unique_tokens() == 0
    assert empty.search([(1, 1.0)], 10) == []


def test_oracle_sums_in_query_order_and_breaks_ties_by_id(oracle):
    """index.rs:248-258: per chunk the sum is 0.0 + q1*d1 + q2*d2 ... in QUERY order (f32, no fma) - a different term order
    gives different bits; equal scores come out in id order (BoundedScoreHeap, candidate.rs:299-334)."""
    f = np.float32
    docs = [("z", [(1, 0.1), (2, 0.2), (3, 0.3)]), ("a", [(1, 0.1), (2, 0.2), (3, 0.3)]), ("m", [(7, 1.0)])]
    ix = oracle.SpladeIndex.build(docs)
    q = [(3, 1e8), (2, -1.5e8), (1, 3.0)]
    got = ix.search(q, 10)
    want = f(0.0)
    for t, w in q:
        want = f(want + f(f(w) * f(dict(docs[0][1])[t])))
    assert [g[0] for g in got] == ["a", "z"] and got[0][1] == got[1][1] == float(want)
    rev = ix.search(list(reversed(q)), 10)
    want_rev = f(0.0)
    for t, w in reversed(q):
        want_rev = f(want_rev + f(f(w) * f(dict(docs[0][1])[t])))
    assert rev[0][1] == float(want_rev) and float(want_rev) != float(want)
    # a token a document names twice is two postings, added one after the other (index.rs:198-200)
    ix2 = oracle.SpladeIndex.build([("d", [(5, 1e8), (5, 1.0), (6, -1e8)])])
    s = ix2.search([(5, 1.0), (6, 1.0)], 1)[0][1]
    assert s == float(f(f(f(0.0) + f(1e8)) + f(1.0)) + f(-1e8))


def test_oracle_filter_and_candidates(oracle):
    """A chunk no posting reaches is not a candidate; one that sums to 0.0 or below is; the filter hides chunks from scoring."""
    ix = oracle.SpladeIndex.build([("a", [(1, 1.0), (2, -1.0)]), ("b", [(3, 1.0)]), ("c", [(1, -2.0)])])
    got = ix.search([(1, 1.0), (2, 1.0)], 10)
    assert got == [("a", 0.0), ("c", -2.0)]
    assert ix.search_with_filter([(1, 1.0), (2, 1.0)], 10, lambda cid: cid != "a") == [("c", -2.0)]
    assert ix.unique_tokens() == 3 and ix.touched([1, 2, 9]) == 3


def test_fuse_hybrid_mirror():
    """`search_hybrid_inner`'s fusion (src/search/query.rs:909-1010) on hand-computed cases.  The reference holds no
    fixture for this arithmetic (its tests run it against a SQLite store): parity unpinned, restated line by line."""
    from cqs_amd.index import IndexResult as R
    from cqs_amd.splade_index import fuse_hybrid
    f = np.float32
    dense = [R("a", 0.9), R("b", 0.5), R("c", 0.2)]
    sparse = [R("c", 8.0), R("d", 4.0), R("a", 2.0)]
    out = fuse_hybrid(dense, sparse, 0.7, 10)
    want = {"a": f(0.7) * f(0.9) + (f(1) - f(0.7)) * (f(2.0) / f(8.0)), "b": f(0.7) * f(0.5) + (f(1) - f(0.7)) * f(0),
            "c": f(0.7) * f(0.2) + (f(1) - f(0.7)) * f(1.0), "d": f(0.7) * f(0) + (f(1) - f(0.7)) * f(0.5)}
    assert [r.id for r in out] == sorted(want, key=lambda i: (-want[i], i))
    for r in out:
        assert r.score == float(want[r.id])
    # alpha <= 0: dense + 0.1 * sparse (query.rs:982-990); truncation; a non-positive sparse maximum zeroes the leg
    out0 = fuse_hybrid(dense, sparse, 0.0, 2)
    assert [r.id for r in out0] == ["a", "b"] and out0[0].score == float(f(0.9) + f(0.25) * f(0.1))
    neg = fuse_hybrid(dense, [R("x", -1.0), R("y", -3.0)], 0.5, 10)
    assert {r.id: r.score for r in neg}["x"] == 0.0 and [r.id for r in neg][:3] == ["a", "b", "c"]
    # equal fused scores: id order (query.rs:1003)
    tie = fuse_hybrid([R("q", 0.5), R("p", 0.5)], [], 1.0, 10)
    assert [r.id for r in tie] == ["p", "q"]
    assert math.isclose(tie[0].score, 0.5)


def test_fuse_hybrid_mirror_equals_the_oracle_restatement(oracle):
    """cqs_amd.splade_index.fuse_hybrid (product mirror) against oracle.hybrid_fuse (written separately from the same
    lines of query.rs) on random pools: overlapping ids, negative sparse pools, tied scores, every alpha branch."""
    from cqs_amd.index import IndexResult as R
    from cqs_amd.splade_index import fuse_hybrid
    rng = np.random.default_rng(0)
    ids = ["c%03d" % i for i in range(60)]
    for trial in range(300):
        nd, ns = int(rng.integers(0, 30)), int(rng.integers(0, 30))
        d = [(ids[i], float(np.float32(rng.uniform(-1, 1)))) for i in rng.choice(60, nd, replace=False)]
        s = [(ids[i], float(np.float32(rng.uniform(-2, 20)))) for i in rng.choice(60, ns, replace=False)]
        if trial % 7 == 0:
            s = [(i, -abs(x)) for i, x in s]
        if trial % 5 == 0 and len(d) > 1:
            d[0] = (d[0][0], d[-1][1])
        alpha = [0.0, 0.3, 0.7, 1.0, -0.5][trial % 5]
        want = oracle.hybrid_fuse(d, s, alpha, 25)
        got = [(r.id, r.score) for r in fuse_hybrid([R(*x) for x in d], [R(*x) for x in s], alpha, 25)]
        assert got == want, trial
