"""The hybrid retrieval of `search_hybrid_inner` (src/search/query.rs:879-1010) end to end on the HIP path - dense leg
(`VectorIndex::search`, k = candidate_count), sparse leg (`SpladeIndex::search_with_filter`), fusion mirror - against the
same pipeline on the oracle (dense scan + SpladeIndex + fusion restatement).  The sparse leg is bit-exact; the dense
leg's scores may differ from the CPU's by <= 1e-5 (summation order), so the fused lists are compared with the dense
parity rule: same ids wherever adjacent oracle scores are further apart than the tolerance, |score difference| <= 2e-5."""
import numpy as np
import pytest

import sparse_cases as sc

pytestmark = pytest.mark.gpu


def _compare(got, want, tol=2e-5):
    assert len(got) == len(want)
    ws = np.array([s for _i, s in want], dtype=np.float64)
    for (gi, gs), (wi, w) in zip(got, want):
        assert abs(gs - w) <= tol, (gi, gs, wi, w)
    i = 0
    while i < len(want):                                   # groups of near-equal oracle scores must hold the same ids
        j = i + 1
        while j < len(want) and ws[j - 1] - ws[j] <= 2 * tol:
            j += 1
        # the group's last member may tie with candidates just past the cut; then only require containment of the clear ones
        gset, wset = {g for g, _ in got[i:j]}, {w for w, _ in want[i:j]}
        if j < len(want):
            assert gset == wset, (i, j, sorted(gset ^ wset))
        i = j


@pytest.mark.parametrize("alpha", [0.7, 0.0, 1.0])
def test_hybrid_search_matches_the_oracle_pipeline(hip, oracle, alpha):
    from cqs_amd import HipIndex, synth
    from cqs_amd.index import IndexResult
    from cqs_amd.splade_index import HipSpladeIndex, fuse_hybrid
    n, dim, cand = 20000, 768, 500
    rows = synth.gaussian_unit(n, seed=91)
    ids = ["chunk%06d" % i for i in np.random.default_rng(3).permutation(n)]
    off, tok, w = synth.sparse_corpus(n, 3000, 10, 60, seed=17)
    dense_ix = HipIndex.build_from_flat(ids, rows)
    sparse_ix = HipSpladeIndex.build_from_csr(ids, off, tok, w)
    sparse_or = oracle.SpladeIndex(off, tok, w, ids=ids)
    rng = np.random.default_rng(5)
    for qi, (qt, qw) in enumerate(synth.sparse_queries(4, 40, 3000, seed=23)):
        q = rows[rng.integers(0, n)] + 0.3 * synth.gaussian_unit(1, seed=100 + qi)[0]
        q = (q / np.linalg.norm(q)).astype(np.float32)
        pred = (lambda cid: int(cid[-1]) % 3 != 0) if qi == 3 else None
        # HIP path, as search_hybrid_inner calls it
        d = dense_ix.search(q, cand) if pred is None else dense_ix.search_with_filter(q, cand, pred)
        s = sparse_ix.search_with_filter(list(zip(qt.tolist(), qw.tolist())), cand, pred)
        fused = fuse_hybrid(d, s, alpha, cand)
        # oracle path
        if pred is None:
            drows, dsc = oracle.index_search(rows, q, cand)
            od = [(ids[int(r)], float(x)) for r, x in zip(drows, dsc)]
        else:                                             # the trait's default: over-fetch 3x, post-filter, take k (src/index.rs:167-193)
            drows, dsc = oracle.index_search(rows, q, min(cand * 3, 1024))
            od = [(ids[int(r)], float(x)) for r, x in zip(drows, dsc) if pred(ids[int(r)])][:cand]
        keep = None if pred is None else np.array([1 if pred(c) else 0 for c in ids], dtype=np.uint8)
        oc, osc = sparse_or.search_raw(qt, qw, cand, keep)
        os_ = [(ids[int(c)], float(x)) for c, x in zip(oc, osc)]
        assert [(r.id, r.score) for r in s] == os_                       # the sparse leg: identical, bit for bit
        want = oracle.hybrid_fuse(od, os_, alpha, cand)
        _compare([(r.id, r.score) for r in fused], want)
    dense_ix.close(); sparse_ix.close()
