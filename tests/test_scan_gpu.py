"""GPU parity tests: the HIP scan + top-k (through the C ABI) against the CPU oracle on the
same seeded inputs, plus the edge cases the reference tests.
Run on an MI355X with `pytest -m gpu`."""
import numpy as np
import pytest

from cqs_amd import DistanceMetric, HipIndex, _lib, merge_keys, synth, unpack_keys
from parity import assert_topk_parity

pytestmark = pytest.mark.gpu
NAN, INF = float("nan"), float("inf")
MARGIN = 64


def check(oracle, idx, rows, q, k, keep=None, mode=0, thr=0.0):
    got_rows, got_scores, counts = idx.search_batch(q, k, keep_bitset=keep, mode=mode, threshold=thr)
    q2 = np.atleast_2d(q)
    for i in range(q2.shape[0]):
        ext_ids, ext_scores = oracle.index_search(rows, q2[i], k + MARGIN, keep, mode, thr)
        ref_ids, _ = oracle.index_search(rows, q2[i], k, keep, mode, thr)
        c = int(counts[i])
        assert_topk_parity(got_rows[i, :c], got_scores[i, :c], ext_ids, ext_scores, len(ref_ids))


@pytest.mark.parametrize("n", [1, 63, 255, 256, 257, 1000, 4097])
@pytest.mark.parametrize("k", [1, 20])
def test_small_corpora(hip, oracle, n, k):
    rows = synth.gaussian_unit(n, seed=100 + n)
    q = synth.gaussian_unit(1, seed=200 + n)[0]
    idx = HipIndex.build_from_flat(None, rows)
    check(oracle, idx, rows, q, k)
    idx.close()


@pytest.mark.parametrize("k", [1, 20, 100, 500, 1024])
def test_reference_self_index_size(hip, oracle, k):
    """BASELINE configs[0] shape: N = 17 523 rows, D = 768, single query."""
    rows = synth.gaussian_unit(17523, seed=synth.SEED_CORPUS)
    qs = synth.gaussian_unit(2, seed=synth.SEED_QUERY)
    idx = HipIndex.build_from_flat(None, rows)
    for q in qs:
        check(oracle, idx, rows, q, k)
    idx.close()


def test_k_larger_than_corpus(hip, oracle):
    rows = synth.gaussian_unit(37, seed=1)
    q = synth.gaussian_unit(1, seed=2)[0]
    idx = HipIndex.build_from_flat(None, rows)
    r, s, c = idx.search_batch(q, 100)
    assert c[0] == 37
    check(oracle, idx, rows, q, 100)
    idx.close()


def test_100k_rows_k20_k500(hip, oracle):
    rows = synth.gaussian_unit(100_000, seed=31)
    q = synth.gaussian_unit(1, seed=32)[0]
    idx = HipIndex.build_from_flat(None, rows)
    check(oracle, idx, rows, q, 20)
    check(oracle, idx, rows, q, 500)
    idx.close()


def test_reference_sin_generators(hip, oracle):
    """The reference's deterministic generators (hnsw/mod.rs:928-940, cagra.rs:1815-1825):
    near-duplicate rows every ~63 seeds -> dense near-ties."""
    for scale in (0.1, 10.0):
        rows = synth.sin_corpus(700, scale=scale)
        idx = HipIndex.build_from_flat(None, rows)
        for seed in (0, 17, 333):
            check(oracle, idx, rows, synth.sin_embedding(seed, scale=scale), 20)
        idx.close()


def test_exact_self_match_first(hip, oracle):
    """cagra.rs:1893-1899 asserts containment for ANN; the exact backend must rank the query's own row first."""
    rows = synth.gaussian_unit(3000, seed=41)
    idx = HipIndex.build_from_flat([f"chunk:{i}" for i in range(3000)], rows)
    for r in (0, 1234, 2999):
        res = idx.search(rows[r], 5)
        assert res[0].id == f"chunk:{r}" and abs(res[0].score - 1.0) < 1e-5
        assert all(res[i].score >= res[i + 1].score for i in range(4))  # cagra.rs:1924-1940
    idx.close()


def test_duplicate_rows_exact_ties_by_row(hip, oracle):
    """Bit-identical scores (duplicate chunks) order by row asc (candidate.rs:327; SURVEY §8a caveat 1)."""
    base = synth.gaussian_unit(40, seed=51)
    rows = np.concatenate([base, base, base, base[:7]])
    q = base[3]
    idx = HipIndex.build_from_flat(None, rows)
    for k in (1, 2, 3, 4, 10, 127):
        r, s, c = idx.search_batch(q, k)
        ids, sc = oracle.index_search(rows, q, k)
        assert list(r[0, :c[0]]) == list(ids), f"k={k}"
        assert np.allclose(s[0, :c[0]], sc, atol=1e-5)
    idx.close()


def test_all_rows_identical_heavy_ties(hip, oracle):
    """Every score equal: the radix threshold search cannot separate by score; ties resolve by row asc."""
    v = synth.gaussian_unit(1, seed=61)[0]
    rows = np.tile(v, (20000, 1))
    idx = HipIndex.build_from_flat(None, rows)
    for k in (1, 20, 500):
        r, s, c = idx.search_batch(v, k)
        assert c[0] == k and list(r[0]) == list(range(k))
    idx.close()


@pytest.mark.parametrize("n", [9000, 150_000, 420_000])
def test_select_index_direct_and_gathered_groups(hip, oracle, n):
    """Round 5: from k = 100 on the gemv scan leaves (lane of the maximum, runner-up) beside each task maximum and the select
    takes a selected group's maximum WITHOUT reading its scores when the runner-up misses the threshold bin; groups with a
    second entry at or above it are still gathered.  Both kinds at once, in every task tier (16-row tasks at 9 000 rows,
    64-row tasks at 150 000, 64- then 32-row tasks at 420 000): a random corpus (mostly direct groups) with planted
    near-duplicates of the query - pairs inside one group, exact duplicates (ties at the maximum: runner-up = maximum), a
    run of 40 consecutive rows - against the oracle, ids exact; k = 100 (the first k that uses the index), 150, 500, 1000,
    with a bitset (maxima over kept rows only) and in PIPELINE mode (clamped scores tie at 1.0 / 0.0)."""
    from cqs_amd import _lib
    rows = synth.gaussian_unit(n, seed=91 + n % 7)
    q = synth.gaussian_unit(1, seed=92)[0]
    rng = np.random.default_rng(93)

    def near(eps):
        v = q + eps * rng.standard_normal(768).astype(np.float32)
        return (v / np.linalg.norm(v)).astype(np.float32)

    for base in (64 * 11, 64 * 57 + 3, n - 70):           # two strong rows inside one group, twice an exact duplicate
        rows[base] = near(0.3); rows[base + 5] = near(0.3)
        rows[base + 9] = rows[base]
    run0 = (n // 2) // 64 * 64 + 20
    for i in range(40):                                     # a run of strong rows across a group boundary
        rows[run0 + i] = near(0.5)
    rows[7] = q; rows[n - 1] = q                            # exact score ties far apart
    idx = HipIndex.build_from_flat(None, rows)
    keep = rng.integers(0, 2**32, size=(n + 31) // 32, dtype=np.uint64).astype(np.uint32)
    for k in (100, 150, 500, 1000):
        check(oracle, idx, rows, q, k)
    check(oracle, idx, rows, q, 500, keep=keep)
    check(oracle, idx, rows, q, 300, mode=_lib.MODE_PIPELINE, thr=0.0)
    check(oracle, idx, rows, q, 300, keep=keep, mode=_lib.MODE_PIPELINE, thr=0.05)
    idx.close()


def test_crowded_bin_forces_level2(hip, oracle):
    """Scores packed inside one 12-bit radix bin (near-duplicate corpus) exercise the second histogram level."""
    rng = np.random.default_rng(5)
    v = synth.gaussian_unit(1, seed=71)[0]
    rows = v[None, :] + 1e-3 * rng.standard_normal((30000, 768)).astype(np.float32)
    rows = (rows / np.linalg.norm(rows, axis=1, keepdims=True)).astype(np.float32)
    idx = HipIndex.build_from_flat(None, rows)
    check(oracle, idx, rows, v, 20)
    check(oracle, idx, rows, v, 500)
    idx.close()


def test_pipeline_mode_clamp_and_threshold(hip, oracle):
    """Brute-force semantics (candidate.rs:550 clamp, :513-519 gate): negatives tie at 0.0 ordered by row."""
    rows = -(synth.gaussian_unit(5000, seed=81) ** 2)
    rows = (rows / np.linalg.norm(rows, axis=1, keepdims=True)).astype(np.float32)
    q = np.abs(synth.gaussian_unit(1, seed=82)[0])
    idx = HipIndex.build_from_flat(None, rows)
    r, s, c = idx.search_batch(q, 20, mode=_lib.MODE_PIPELINE, threshold=0.0)
    assert c[0] == 20 and list(r[0]) == list(range(20)) and np.all(s[0] == 0.0)
    ids, sc = oracle.brute_force(rows, q, 20, 0.0)
    assert list(ids) == list(r[0])
    r, s, c = idx.search_batch(q, 20, mode=_lib.MODE_PIPELINE, threshold=0.3)
    assert c[0] == 0
    idx.close()
    # mixed signs with the CLI default threshold 0.3 (cli/definitions.rs:174-183)
    rows = synth.sin_corpus(400)
    q = synth.sin_embedding(5)
    idx = HipIndex.build_from_flat(None, rows)
    check(oracle, idx, rows, q, 50, mode=_lib.MODE_PIPELINE, thr=0.3)
    r, s, c = idx.search_batch(q, 50, mode=_lib.MODE_PIPELINE, threshold=0.3)
    ids, sc = oracle.brute_force(rows, q, 50, 0.3)
    assert c[0] == len(ids) and np.all(s[0, :c[0]] >= 0.3) and np.all(s[0, :c[0]] <= 1.0)
    idx.close()


def test_non_finite_rows_never_emitted(hip, oracle):
    rows = synth.gaussian_unit(2000, seed=91)
    rows[5, 10] = NAN
    rows[77, 0] = INF
    rows[1999, 767] = -INF
    q = rows[6].copy()
    idx = HipIndex.build_from_flat(None, rows)
    r, s, c = idx.search_batch(q, 1024)
    assert c[0] == 1024
    got = set(int(x) for x in r[0, :c[0]])
    assert not ({5, 77, 1999} & got) and np.all(np.isfinite(s[0, :c[0]]))
    check(oracle, idx, rows, q, 20)
    idx.close()


def test_query_guards(hip, oracle):
    """cagra.rs:443-470: k==0, dim mismatch, non-finite query -> empty; never an error."""
    rows = synth.gaussian_unit(500, seed=101)
    idx = HipIndex.build_from_flat(None, rows)
    q = rows[0]
    assert idx.search(q, 0) == []
    assert idx.search(q[:100], 5) == []
    bad = q.copy(); bad[3] = NAN
    assert idx.search(bad, 5) == []
    r, s, c = idx.search_batch(np.stack([q, bad, rows[1]]), 5)  # C ABI: bad query -> count 0, others fine
    assert list(c) == [5, 0, 5] and r[0, 0] == 0 and r[2, 0] == 1
    r, s, c = idx.search_batch(q[:100], 5)  # C ABI: dim mismatch -> OK + count 0
    assert c[0] == 0
    with pytest.raises(Exception):
        idx.search_batch(q, 5000)  # k > max_k is an argument error at the C ABI
    assert len(idx.search(q, 5000)) == 500  # the trait mirror caps k at max_k first (query.rs:232-245)
    assert not idx.is_poisoned() and idx.max_k() == 1024 and idx.index_scores_are_cosine()
    assert idx.name() == "HIP" and idx.dim() == 768 and len(idx) == 500 and not idx.is_empty()
    idx.close()
    empty = HipIndex.build_from_flat(None, np.zeros((0, 768), np.float32))
    assert empty.is_empty() and empty.search(q, 5) == []
    empty.close()


def test_bitset_filter(hip, oracle):
    """cagra.rs:747-775 semantics."""
    n = 3000
    rows = synth.gaussian_unit(n, seed=111)
    q = synth.gaussian_unit(1, seed=112)[0]
    idx = HipIndex.build_from_flat(None, rows)
    rng = np.random.default_rng(1)
    words = (n + 31) // 32
    keep = rng.integers(0, 2**32, size=words, dtype=np.uint64).astype(np.uint32)
    check(oracle, idx, rows, q, 20, keep=keep)
    r, s, c = idx.search_batch(q, 20, keep_bitset=np.zeros(words, np.uint32))
    assert c[0] == 0
    allb = np.full(words, 0xFFFFFFFF, np.uint32)
    r1, s1, c1 = idx.search_batch(q, 20, keep_bitset=allb)
    r0, s0, c0 = idx.search_batch(q, 20)
    assert list(r1[0]) == list(r0[0])
    few = np.zeros(words, np.uint32); few[1] = 0b1011; few[90] = 1 << 31
    r, s, c = idx.search_batch(q, 20, keep_bitset=few)
    assert c[0] == 4 and set(r[0, :4]) == {32, 33, 35, 90 * 32 + 31}
    check(oracle, idx, rows, q, 20, keep=few)
    # whole 64-row groups filtered out (the kernel skips their HBM reads)
    blocky = np.zeros(words, np.uint32); blocky[10:14] = 0xFFFFFFFF; blocky[40] = 0x00010000
    check(oracle, idx, rows, q, 50, keep=blocky)
    idx.close()
    # predicate form through the trait mirror
    ids = [f"src/{'a' if i % 3 else 'b'}.rs:{i}" for i in range(n)]
    idx = HipIndex.build_from_flat(ids, rows)
    res = idx.search_with_filter(q, 10, lambda s: s.startswith("src/b"))
    assert len(res) == 10 and all(x.id.startswith("src/b") for x in res)
    keepb = np.packbits(np.array([i % 3 == 0 for i in range(n)]), bitorder="little")
    keepb = np.concatenate([keepb, np.zeros((-len(keepb)) % 4, np.uint8)]).view(np.uint32)
    oid, _ = oracle.index_search(rows, q, 10, keepb)
    assert [x.id for x in res] == [ids[int(i)] for i in oid]
    idx.close()


@pytest.mark.parametrize("b", [2, 3, 4, 5, 8, 9, 13, 33])
def test_query_blocks_match_single(hip, oracle, b):
    rows = synth.gaussian_unit(6000, seed=121)
    qs = synth.gaussian_unit(b, seed=122 + b)
    idx = HipIndex.build_from_flat(None, rows)
    check(oracle, idx, rows, qs, 20)
    idx.close()


KSPLIT_CASE = """
import sys
sys.path.insert(0, %r)
sys.path.insert(0, %r)
import numpy as np
from cqs_amd import HipIndex, synth
from oracle import oracle
from parity import assert_topk_parity
for b, n, dim, k in [(9, 700, 768, 20), (32, 20000, 768, 20), (33, 9000, 768, 50), (64, 40000, 768, 20), (48, 5000, 1024, 10),
                     (20, 3000, 512, 10), (64, 2000, 256, 10)]:
    rows = synth.gaussian_unit(n, seed=300 + b, dim=dim)
    qs = synth.gaussian_unit(b, seed=400 + b, dim=dim)
    keep = np.random.default_rng(b).integers(0, 2**32, size=(n + 31) // 32, dtype=np.uint64).astype(np.uint32)
    idx = HipIndex.build_from_flat(None, rows)
    for kb in (None, keep):
        r, s, c = idx.search_batch(qs, k, keep_bitset=kb)
        for i in range(b):
            ids, sc = oracle.index_search(rows, qs[i], k + 64, kb)
            assert_topk_parity(r[i, :c[i]], s[i, :c[i]], ids, sc, k)
    idx.close()
print("ok")
"""


@pytest.mark.parametrize("ksplit", ["1", "2", "0"])
def test_k_split_matrix_core_kernel_all_forms(hip, ksplit):
    """Round 5: query blocks of 9-32 at 256 / 512 / 768 / 1024 dimensions run `scan_mfma_ks_kernel` (K split over the waves,
    the query tile resident in registers, 256-byte row pieces through wave-private LDS images).  Its eight-wave form for
    33-64 queries is measured slower than the LDS-tiled kernel and stays behind CQS_HIP_SCAN_MFMA_KSPLIT=2; =0 disables
    the kernel.  The switch is read once per process, so each setting runs in a process of its own: every form against the
    oracle, with and without a bitset, tiles that end inside the last row tile, the work queue (40 000 rows = 625 tiles)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CQS_HIP_SCAN_MFMA_KSPLIT=ksplit)
    p = subprocess.run([sys.executable, "-c", KSPLIT_CASE % (root, os.path.join(root, "tests"))], env=env, capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0 and p.stdout.strip().endswith("ok"), p.stderr[-3000:]


@pytest.mark.parametrize("b,n,dim,k", [(16, 5000, 768, 20), (64, 9000, 768, 20), (100, 4097, 768, 50),
                                       (256, 20000, 768, 20), (300, 3000, 768, 20), (40, 2000, 384, 10),
                                       (32, 2500, 1024, 10), (24, 1500, 100, 10)])
def test_batched_matrix_core_path(hip, oracle, b, n, dim, k):
    """Query blocks >= 9 go through the f32-MFMA kernel (dim % 32 == 0), else the gemv passes;
    either way every query must match the oracle."""
    rows = synth.gaussian_unit(n, dim, seed=171 + b)
    qs = synth.gaussian_unit(b, dim, seed=172 + b)
    qs[1] = rows[n // 2]          # an exact self-match inside the block
    idx = HipIndex.build_from_flat(None, rows)
    check(oracle, idx, rows, qs, k)
    idx.close()


def test_batched_filters_modes_and_edges(hip, oracle):
    n = 6000
    rows = synth.gaussian_unit(n, seed=181)
    rows[17, 5] = NAN
    rows[n - 1, 0] = INF
    qs = synth.gaussian_unit(48, seed=182)
    idx = HipIndex.build_from_flat(None, rows)
    rng = np.random.default_rng(3)
    keep = rng.integers(0, 2**32, size=(n + 31) // 32, dtype=np.uint64).astype(np.uint32)
    check(oracle, idx, rows, qs, 20, keep=keep)
    check(oracle, idx, rows, qs, 30, mode=_lib.MODE_PIPELINE, thr=0.05)
    bad = qs.copy(); bad[7, 3] = NAN
    r, s, c = idx.search_batch(bad, 10)
    assert c[7] == 0 and all(c[i] == 10 for i in range(48) if i != 7)
    r0, s0, c0 = idx.search_batch(qs, 10)
    assert np.array_equal(np.delete(r, 7, 0), np.delete(r0, 7, 0))  # neighbours of a bad query are unaffected
    idx.close()


def test_batched_vs_single_scores_close(hip, oracle):
    """MFMA (k-ordered fma chain) and gemv (lane-partial sums) round differently: scores agree to 1e-6."""
    rows = synth.gaussian_unit(8000, seed=191)
    qs = synth.gaussian_unit(64, seed=192)
    idx = HipIndex.build_from_flat(None, rows)
    rb, sb, cb = idx.search_batch(qs, 20)
    for i in (0, 31, 63):
        r1, s1, c1 = idx.search_batch(qs[i], 20)
        assert np.max(np.abs(s1[0] - sb[i])) < 2e-6
    idx.close()


@pytest.mark.parametrize("dim", [4, 100, 256, 384, 512, 1024, 1536, 2048])
def test_other_dims(hip, oracle, dim):
    rows = synth.gaussian_unit(1500, dim, seed=131)
    qs = synth.gaussian_unit(3, dim, seed=132)
    idx = HipIndex.build_from_flat(None, rows)
    check(oracle, idx, rows, qs, 10)
    idx.close()


@pytest.mark.parametrize("dim", [2052, 2560, 3000, 4096])
def test_wide_dims_single_and_batched(hip, oracle, dim):
    """The reference's presets reach 2560 and 4096 dimensions (src/embedder/models.rs:515,572) and its index traits are
    dimension-generic: rows of 9-16 KiB take the one-row-per-batch gemv variants (one query per pass), query blocks of
    >= 9 the matrix-core kernel (any dim % 32 == 0).  A corpus large enough for 64-row tasks, the keep bitset and the
    PIPELINE mode ride along; 4100 is refused at create."""
    rows = synth.gaussian_unit(9000, dim, seed=133)
    qs = synth.gaussian_unit(12, dim, seed=134)
    idx = HipIndex.build_from_flat(None, rows)
    check(oracle, idx, rows, qs[0], 20)
    check(oracle, idx, rows, qs[:3], 10)                        # three passes of one query
    check(oracle, idx, rows, qs, 20)                            # 12 queries: matrix cores when dim % 32 == 0
    keep = np.random.default_rng(135).integers(0, 2**32, size=(9000 + 31) // 32, dtype=np.uint64).astype(np.uint32)
    check(oracle, idx, rows, qs[1], 20, keep=keep)
    check(oracle, idx, rows, qs[2], 20, mode=1, thr=0.02)
    idx.close()
    small = HipIndex.build_from_flat(None, rows[:700])          # 16-row tasks
    check(oracle, small, rows[:700], qs[3], 10)
    small.close()
    with pytest.raises(Exception):
        HipIndex.build_from_flat(None, np.zeros((10, 4100), np.float32))


def test_dot_metric_unnormalised(hip, oracle):
    rng = np.random.default_rng(7)
    rows = (rng.standard_normal((4000, 768)) * rng.uniform(0.1, 30, (4000, 1))).astype(np.float32)
    q = rng.standard_normal(768).astype(np.float32)
    idx = HipIndex.build_from_flat(None, rows, DistanceMetric.DotProduct)
    r, s, c = idx.search_batch(q, 20)
    ids, sc = oracle.index_search(rows, q, 20)
    assert list(r[0]) == list(ids)
    assert np.allclose(s[0], sc, rtol=1e-5, atol=1e-4)
    assert not idx.index_scores_are_cosine()
    idx.close()


def test_extend_and_row_base(hip, oracle):
    rows = synth.gaussian_unit(2500, seed=141)
    q = synth.gaussian_unit(1, seed=142)[0]
    idx = HipIndex.build_from_flat(None, rows[:1000])
    idx.extend(None, rows[1000:1800])
    idx.extend(None, rows[1800:])
    assert len(idx) == 2500
    check(oracle, idx, rows, q, 20)
    idx.close()
    # two shards with row_base, merged on the host == whole corpus (associative comparator)
    a = HipIndex.build_from_flat(None, rows[:1300], row_base=0)
    b = HipIndex.build_from_flat(None, rows[1300:], row_base=1300)
    k = 20
    ra, sa, ca = a.search_batch(q, k)
    rb, sb, cb = b.search_batch(q, k)
    assert rb[0].min() >= 1300

    def pack(rows_, scores_):
        bits = scores_.view(np.uint32).astype(np.uint64)
        ok = np.where(bits >> 31 != 0, (~bits) & 0xFFFFFFFF, bits ^ 0x80000000)
        return (ok << np.uint64(32)) | (np.uint64(0xFFFFFFFF) - rows_.astype(np.uint64))

    lists = np.stack([pack(ra[0], sa[0]), pack(rb[0], sb[0])])
    merged = merge_keys(lists, np.array([ca[0], cb[0]], np.uint32), k)
    mr, ms = unpack_keys(merged)
    ext_ids, ext_scores = oracle.index_search(rows, q, k + MARGIN)
    assert_topk_parity(mr, ms, ext_ids, ext_scores, k)
    a.close(); b.close()


def test_device_api_with_torch(hip, oracle):
    """cqs_hip_index_search_device: buffers in HBM, launched on torch's current stream."""
    import torch
    rows = synth.gaussian_unit(8000, seed=151)
    qs = synth.gaussian_unit(4, seed=152)
    d_rows = torch.from_numpy(rows).cuda()
    d_q = torch.from_numpy(qs).cuda()
    idx = HipIndex.build_from_device(None, d_rows.data_ptr(), 8000, 768, borrow=True, keepalive=d_rows)
    k = 20
    keys = torch.zeros((4, k), dtype=torch.int64, device="cuda")
    counts = torch.zeros((4,), dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    idx.search_device(d_q.data_ptr(), 4, k, keys.data_ptr(), counts.data_ptr(), stream=st)
    torch.cuda.synchronize()
    hk = keys.cpu().numpy().view(np.uint64)
    hc = counts.cpu().numpy()
    for i in range(4):
        r, s = unpack_keys(hk[i, :hc[i]])
        ext_ids, ext_scores = oracle.index_search(rows, qs[i], k + MARGIN)
        assert_topk_parity(r, s, ext_ids, ext_scores, k)
        assert np.all(hk[i, :-1] >= hk[i, 1:])  # packed keys sorted descending
    idx.close()


def test_full_size_properties_1m(hip):
    """BASELINE configs[1] size (1M x 768 fp32) through size-independent properties: planted
    rows come back first with the right scores, results are sorted, scores equal a direct
    torch dot of the returned rows, and two half-corpus shards merge to the whole-corpus answer."""
    import torch
    n, dim, k = 1_000_000, 768, 20
    g = torch.Generator(device="cuda"); g.manual_seed(1234)
    rows = torch.randn((n, dim), generator=g, device="cuda", dtype=torch.float32)
    rows /= rows.norm(dim=1, keepdim=True)
    q = torch.randn((dim,), generator=g, device="cuda", dtype=torch.float32)
    q /= q.norm()
    planted = [0, 511_111, n - 1]
    for j, r in enumerate(planted):  # cos = 1 - j*0.05 exactly-ish
        noise = torch.randn((dim,), generator=g, device="cuda")
        noise -= (noise @ q) * q
        noise /= noise.norm()
        c = 1.0 - 0.05 * j
        rows[r] = c * q + (1 - c * c) ** 0.5 * noise
    idx = HipIndex.build_from_device(None, rows.data_ptr(), n, dim, borrow=True, keepalive=rows)
    keys = torch.zeros((1, k), dtype=torch.int64, device="cuda")
    counts = torch.zeros((1,), dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    idx.search_device(q.data_ptr(), 1, k, keys.data_ptr(), counts.data_ptr(), stream=st)
    torch.cuda.synchronize()
    r, s = unpack_keys(keys.cpu().numpy().view(np.uint64)[0])
    assert counts.item() == k and list(r[:3]) == planted
    assert np.all(np.diff(s) <= 0)
    direct = (rows[torch.from_numpy(r.astype(np.int64)).cuda()].double() @ q.double()).cpu().numpy()
    assert np.max(np.abs(direct - s)) <= 1e-5
    # exhaustive check of the threshold: nothing outside the list beats the k-th score by > 2e-6
    allsc = rows @ q
    assert int((allsc > float(s[-1]) + 2e-6).sum().item()) <= k - 1
    # two shards + host merge == whole
    half = n // 2
    a = HipIndex.build_from_device(None, rows.data_ptr(), half, dim, borrow=True, row_base=0)
    b = HipIndex.build_from_device(None, rows[half:].data_ptr(), n - half, dim, borrow=True, row_base=half)
    ka = torch.zeros((2, k), dtype=torch.int64, device="cuda")
    ca = torch.zeros((2,), dtype=torch.int32, device="cuda")
    a.search_device(q.data_ptr(), 1, k, ka[0].data_ptr(), ca[0:].data_ptr(), stream=st)
    b.search_device(q.data_ptr(), 1, k, ka[1].data_ptr(), ca[1:].data_ptr(), stream=st)
    torch.cuda.synchronize()
    merged = merge_keys(ka.cpu().numpy().view(np.uint64), ca.cpu().numpy().astype(np.uint32), k)
    assert list(merged) == list(keys.cpu().numpy().view(np.uint64)[0])
    for x in (idx, a, b):
        x.close()


def test_device_api_null_stream_orders_with_torch(hip, oracle):
    """stream = NULL must mean the HIP null stream (= torch's default stream): a torch op enqueued right
    after the search sees the results without any explicit synchronisation (regression: the search used
    to run on the index's private stream and raced with the consumer)."""
    import torch
    rows = synth.gaussian_unit(50_000, seed=201)
    qs = synth.gaussian_unit(6, seed=202)
    d_rows = torch.from_numpy(rows).cuda()
    idx = HipIndex.build_from_device(None, d_rows.data_ptr(), 50_000, 768, borrow=True, keepalive=d_rows)
    k = 20
    for i in range(6):
        d_q = torch.from_numpy(qs[i:i + 1]).cuda()
        keys = torch.zeros((1, k), dtype=torch.int64, device="cuda")
        counts = torch.zeros((1,), dtype=torch.int32, device="cuda")
        idx.search_device(d_q.data_ptr(), 1, k, keys.data_ptr(), counts.data_ptr(), stream=0)
        snapshot = keys.clone()          # default-stream consumer, no synchronize() in between
        r, s = unpack_keys(snapshot.cpu().numpy().view(np.uint64)[0])
        ext_ids, ext_scores = oracle.index_search(rows, qs[i], k + MARGIN)
        assert_topk_parity(r, s, ext_ids, ext_scores, k)
    idx.close()


class _FakeStore:
    """Just enough of `Store` for a backend: dim, chunk_count(), embedding_batches() in rowid order
    (src/store/chunks/async_helpers.rs:189-203)."""

    def __init__(self, ids, rows):
        self.ids, self.rows, self.dim = ids, rows, rows.shape[1]

    def chunk_count(self):
        return len(self.ids)

    def embedding_batches(self, batch):
        for i in range(0, len(self.ids), batch):
            yield [(self.ids[j], self.rows[j]) for j in range(i, min(len(self.ids), i + batch))]


def test_backend_try_open_and_persistence(hip, oracle, tmp_path):
    """`IndexBackend::try_open` contract (src/index.rs:271-291; CagraBackend src/cagra.rs:1676-1802):
    below threshold -> None; build from the store; persisted blob + sidecar reused on the next open;
    a corrupted / stale pair is rejected, deleted and rebuilt."""
    from cqs_amd import BackendContext, HipBackend
    n = 1200
    rows = synth.gaussian_unit(n, seed=211)
    ids = [f"src/f{i % 7}.rs:{i}:abcd" for i in range(n)]
    store = _FakeStore(ids, rows)
    be = HipBackend()
    assert be.name() == "hip" and be.priority() > 150
    assert be.try_open(BackendContext(str(tmp_path), store, hip_threshold=5000)) is None      # gate
    idx = be.try_open(BackendContext(str(tmp_path), store, hip_threshold=1000))
    assert idx is not None and len(idx) == n and (tmp_path / "index.hipflat").exists()
    q = rows[77]
    res = idx.search(q, 5)
    assert res[0].id == ids[77]
    idx.close()
    idx2 = be.try_open(BackendContext(str(tmp_path), store, hip_threshold=1000))                # persisted path
    assert [r.id for r in idx2.search(q, 5)] == [r.id for r in res]
    idx2.close()
    # flip one byte of the blob: checksum mismatch -> rejected, files deleted, rebuilt from the store
    blob = tmp_path / "index.hipflat"
    raw = bytearray(blob.read_bytes()); raw[64 + 123] ^= 0x40; blob.write_bytes(bytes(raw))
    with pytest.raises(ValueError):
        HipIndex.load(str(blob), 768, n)
    idx3 = be.try_open(BackendContext(str(tmp_path), store, hip_threshold=1000))
    assert idx3 is not None and [r.id for r in idx3.search(q, 5)] == [r.id for r in res]
    idx3.close()
    with pytest.raises(ValueError):                       # stale sidecar (chunk_count changed)
        HipIndex.load(str(blob), 768, n + 1)
    with pytest.raises(ValueError):
        HipIndex.load(str(blob), 1024, n)
    # zero / non-finite rows are skipped at build like prepare_index_data (src/hnsw/mod.rs:717-731)
    rows2 = rows.copy(); rows2[5] = 0; rows2[9, 3] = NAN
    idx4 = be.try_open(BackendContext(str(tmp_path / "b"), _FakeStore(ids, rows2), hip_threshold=1000, persist=False))
    assert len(idx4) == n - 2 and ids[5] not in idx4.id_map and ids[9] not in idx4.id_map
    idx4.close()


@pytest.mark.parametrize("n,dim", [(400_001, 64), (150_000, 128), (2_000_003, 16)])
def test_task_layouts_across_corpus_sizes(hip, oracle, n, dim):
    """The scan cuts the corpus into wave tasks differently by size: 16-row tasks (small), 64-row tasks,
    64-row tasks with a tail of 32-row tasks (>= ~393k rows), and beyond ~1.5M rows a persistent grid fed
    by the work queue.  Full parity (ids + scores) against the oracle in each regime, with the best rows
    planted at both ends so that every task tier holds part of the answer; single query, query pair,
    bitset filter and k = 500."""
    rows = synth.gaussian_unit(n, dim=dim, seed=700 + dim)
    qs = synth.gaussian_unit(2, dim=dim, seed=701 + dim)
    for j, r in enumerate((0, 1, 65, n // 2, n - 40_000, n - 20_001, n - 2, n - 1)):
        v = qs[0] + 0.02 * (j + 1) * rows[r]
        rows[r] = v / np.linalg.norm(v)
    idx = HipIndex.build_from_flat(None, rows)
    check(oracle, idx, rows, qs[0], 20)
    check(oracle, idx, rows, qs, 500)
    rng = np.random.default_rng(n)
    keep = (rng.random(n) < 0.3)
    keep[-3000:] = True          # the tail tasks stay (partly) selected
    keep[n // 3: n // 3 + 5000] = False
    bits = np.zeros((n + 31) // 32, dtype=np.uint32)
    np.bitwise_or.at(bits, np.nonzero(keep)[0] // 32, (np.uint32(1) << (np.nonzero(keep)[0] % 32).astype(np.uint32)))
    check(oracle, idx, rows, qs[0], 100, keep=bits)
    idx.close()


@pytest.mark.parametrize("b,n,dim", [(5, 1, 768), (6, 63, 100), (7, 4097, 384), (8, 70_000, 768), (6, 3000, 1024),
                                     (8, 2500, 4), (7, 1500, 1280)])
def test_five_to_eight_query_blocks(hip, oracle, b, n, dim):
    """5..7 queries ride the 8-query pass with padded (never stored) slots (dim <= 1024; beyond that 2 + 2 +
    ... passes): ragged corpus ends, partial 1-KiB chunks, the bitset filter and PIPELINE mode."""
    rows = synth.gaussian_unit(n, dim, seed=900 + n % 97)
    qs = synth.gaussian_unit(b, dim, seed=901 + b)
    idx = HipIndex.build_from_flat(None, rows)
    k = min(20, n)
    check(oracle, idx, rows, qs, k)
    rng = np.random.default_rng(n + b)
    keep = rng.random(n) < 0.5
    keep[: min(n, 100)] = True
    bits = np.zeros((n + 31) // 32, dtype=np.uint32)
    on = np.nonzero(keep)[0]
    np.bitwise_or.at(bits, on // 32, np.uint32(1) << (on % 32).astype(np.uint32))
    check(oracle, idx, rows, qs, k, keep=bits)
    check(oracle, idx, rows, qs, k, mode=1, thr=0.02)
    idx.close()


# ---- round 2: regimes bench.py times but nothing verified -------------------------
def _bits(keep):
    bits = np.zeros((len(keep) + 31) // 32, dtype=np.uint32)
    on = np.nonzero(keep)[0]
    np.bitwise_or.at(bits, on // 32, np.uint32(1) << (on % 32).astype(np.uint32))
    return bits


@pytest.mark.parametrize("b", [20, 64, 128, 256])
def test_mfma_work_queue_regime(hip, oracle, b):
    """scan_mfma_kernel continues from its work queue once a launch has more row tiles than workgroups
    (n_pad / RT > CUs: 80 000 rows = 313 tiles of 256 rows for the 32/64-query configs, 625 tiles of 128 rows for the
    128/256-query ones, on 256 CUs): dequeue, the s_task broadcast and tiles 2..3 of a workgroup.  Every query
    of the block against the oracle, with the best rows planted in the FIRST and the LAST tiles."""
    n = 80_000
    rows = synth.gaussian_unit(n, seed=2000 + b)
    qs = synth.gaussian_unit(b, seed=2001 + b)
    for j, r in enumerate((0, 255, 256, 40_000, n - 257, n - 129, n - 1)):     # first / queue-fed / last tiles
        v = qs[j % b] + 0.03 * (j + 1) * rows[r]
        rows[r] = v / np.linalg.norm(v)
    idx = HipIndex.build_from_flat(None, rows)
    check(oracle, idx, rows, qs, 20)
    idx.close()


def test_mfma_work_queue_filter_pipeline_and_second_slot(hip, oracle):
    """Queue regime with (a) a keep-bitset that empties whole tiles, (b) PIPELINE mode, (c) b = 300 = two
    launches (256 + 44 queries: the second launch uses its own work-queue slot and another tile config)."""
    n = 80_000
    rows = synth.gaussian_unit(n, seed=2100)
    qs = synth.gaussian_unit(300, seed=2101)
    idx = HipIndex.build_from_flat(None, rows)
    rng = np.random.default_rng(21)
    keep = rng.random(n) < 0.4
    keep[1000:9000] = False                   # several consecutive tiles fully filtered
    keep[-300:] = True
    check(oracle, idx, rows, qs[:40], 20, keep=_bits(keep))
    check(oracle, idx, rows, qs[:130], 30, mode=_lib.MODE_PIPELINE, thr=0.05)
    # b = 300: check a spread of queries from both launches (incl. the block edges 255 / 256 / 299)
    got_rows, got_scores, counts = idx.search_batch(qs, 20)
    for i in (0, 1, 127, 128, 254, 255, 256, 257, 280, 298, 299):
        ext_ids, ext_scores = oracle.index_search(rows, qs[i], 20 + MARGIN)
        assert_topk_parity(got_rows[i, :counts[i]], got_scores[i, :counts[i]], ext_ids, ext_scores, 20)
    idx.close()


def _property_check(torch, rows, q, r, s, k, planted=None):
    """Size-independent properties of one query's top-k (rows / q on the device; r, s host arrays)."""
    assert len(r) == k
    assert np.all(np.diff(s) <= 0), "not sorted by score"
    for i in range(k - 1):
        assert s[i] > s[i + 1] or r[i] < r[i + 1], "exact ties not ordered by row"
    if planted is not None:
        assert list(r[:len(planted)]) == list(planted), (list(r[:len(planted)]), planted)
    direct = (rows[torch.from_numpy(r.astype(np.int64)).to(rows.device)].double() @ q.double()).cpu().numpy()
    assert np.max(np.abs(direct - s)) <= 1e-5, float(np.max(np.abs(direct - s)))


def test_full_size_against_the_oracle_1m(hip, oracle):
    """the ORACLE itself at BASELINE's full sizes, beside the property tests.  configs[1]:
    1M x 768 fp32, one query, k = 20 and k = 500 (the production k, src/limits.rs:315-320); configs[2]: four queries
    of one 256-query block (matrix-core kernel, work-queue regime) - each against `oracle.index_search` over the same
    rows with the usual parity rule (identical ids wherever oracle scores are > 2e-6 apart, |score diff| <= 1e-5).
    ~0.1-0.5 s of CPU per oracle query, 3 GB of host RAM."""
    import torch
    n, dim, b = 1_000_000, 768, 256
    g = torch.Generator(device="cuda"); g.manual_seed(20260)
    d_rows = torch.empty((n, dim), device="cuda", dtype=torch.float32)
    for lo in range(0, n, 1 << 18):
        hi = min(n, lo + (1 << 18))
        x = torch.randn((hi - lo, dim), generator=g, device="cuda"); x /= x.norm(dim=1, keepdim=True); d_rows[lo:hi] = x
    d_qs = torch.randn((b, dim), generator=g, device="cuda"); d_qs /= d_qs.norm(dim=1, keepdim=True)
    rows = d_rows.cpu().numpy()
    qs = d_qs.cpu().numpy()
    idx = HipIndex.build_from_device(None, d_rows.data_ptr(), n, dim, borrow=True, keepalive=d_rows)
    st = torch.cuda.current_stream().cuda_stream
    for k in (20, 500):                                                  # configs[1]: the HBM-streaming kernel
        keys = torch.zeros((1, k), dtype=torch.int64, device="cuda"); cnt = torch.zeros((1,), dtype=torch.int32, device="cuda")
        idx.search_device(d_qs[3].data_ptr(), 1, k, keys.data_ptr(), cnt.data_ptr(), stream=st)
        torch.cuda.synchronize()
        assert cnt.item() == k
        r, s = unpack_keys(keys.cpu().numpy().view(np.uint64)[0])
        ext_ids, ext_scores = oracle.index_search(rows, qs[3], k + MARGIN)
        assert_topk_parity(r, s, ext_ids, ext_scores, k)
    k = 20                                                                # configs[2]: one 256-query block
    keys = torch.zeros((b, k), dtype=torch.int64, device="cuda"); cnt = torch.zeros((b,), dtype=torch.int32, device="cuda")
    idx.search_device(d_qs.data_ptr(), b, k, keys.data_ptr(), cnt.data_ptr(), stream=st)
    torch.cuda.synchronize()
    hk = keys.cpu().numpy().view(np.uint64)
    for qi in (0, 77, 128, 255):
        assert int(cnt[qi].item()) == k
        r, s = unpack_keys(hk[qi])
        ext_ids, ext_scores = oracle.index_search(rows, qs[qi], k + MARGIN)
        assert_topk_parity(r, s, ext_ids, ext_scores, k)
    # round 5: blocks of 9-32 queries run the K-split kernel (15 625 row tiles over 512 workgroups: its work queue, the
    # two-tiles-ahead hand-over) and blocks of 33-64 the LDS-tiled one - both against the oracle at full size, k = 20 and the
    # production k = 500 (at 500 the gemv-only select index does not apply: matrix-core producers do not write it)
    for bb, kk, picks in ((32, 20, (0, 13, 31)), (17, 500, (0, 16)), (64, 20, (5, 63))):
        keys = torch.zeros((bb, kk), dtype=torch.int64, device="cuda"); cnt = torch.zeros((bb,), dtype=torch.int32, device="cuda")
        idx.search_device(d_qs.data_ptr(), bb, kk, keys.data_ptr(), cnt.data_ptr(), stream=st)
        torch.cuda.synchronize()
        hk = keys.cpu().numpy().view(np.uint64)
        for qi in picks:
            assert int(cnt[qi].item()) == kk
            r, s = unpack_keys(hk[qi])
            ext_ids, ext_scores = oracle.index_search(rows, qs[qi], kk + MARGIN)
            assert_topk_parity(r, s, ext_ids, ext_scores, kk)
    # and through the blocking host API (what VectorIndex::search calls: host query in, host results out)
    got_rows, got_scores, counts = idx.search_batch(qs[9], 20)
    ext_ids, ext_scores = oracle.index_search(rows, qs[9], 20 + MARGIN)
    assert_topk_parity(got_rows[0, :counts[0]], got_scores[0, :counts[0]], ext_ids, ext_scores, 20)
    got_rows, got_scores, counts = idx.search_batch(qs[9], 500)                # (k >= 100: the select index on the gemv path)
    ext_ids, ext_scores = oracle.index_search(rows, qs[9], 500 + MARGIN)
    assert_topk_parity(got_rows[0, :counts[0]], got_scores[0, :counts[0]], ext_ids, ext_scores, 500)
    idx.close()


def test_full_size_properties_1m_x_256_queries(hip):
    """BASELINE configs[2] at full size (1M x 768, 256-query block = 7 813 row tiles through the matrix-core
    kernel's work queue): for every query sortedness + direct fp64 dot of the returned rows <= 1e-5; planted rows
    first for a spread of queries; exhaustive threshold count (nothing outside the list beats the k-th score by
    more than 2e-6) for all 256 queries; and the block equals 256 single-query searches (other kernel, same ids)."""
    import torch
    n, dim, k, b = 1_000_000, 768, 20, 256
    g = torch.Generator(device="cuda"); g.manual_seed(4321)
    rows = torch.empty((n, dim), device="cuda", dtype=torch.float32)
    for lo in range(0, n, 1 << 18):
        hi = min(n, lo + (1 << 18))
        x = torch.randn((hi - lo, dim), generator=g, device="cuda"); x /= x.norm(dim=1, keepdim=True); rows[lo:hi] = x
    qs = torch.randn((b, dim), generator=g, device="cuda"); qs /= qs.norm(dim=1, keepdim=True)
    planted = {}
    for qi, slots in ((0, [0, 999_999, 500_000]), (100, [127, 128, 999_872]), (255, [999_998, 1, 12_345])):
        for j, r in enumerate(slots):
            noise = torch.randn((dim,), generator=g, device="cuda")
            noise -= (noise @ qs[qi]) * qs[qi]; noise /= noise.norm()
            c = 0.95 - 0.05 * j - 0.001 * qi / 255
            rows[r] = c * qs[qi] + (1 - c * c) ** 0.5 * noise
        planted[qi] = slots
    idx = HipIndex.build_from_device(None, rows.data_ptr(), n, dim, borrow=True, keepalive=rows)
    keys = torch.zeros((b, k), dtype=torch.int64, device="cuda")
    counts = torch.zeros((b,), dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    idx.search_device(qs.data_ptr(), b, k, keys.data_ptr(), counts.data_ptr(), stream=st)
    torch.cuda.synchronize()
    assert bool((counts == k).all())
    hk = keys.cpu().numpy().view(np.uint64)
    allsc = rows @ qs.T                                            # [n, b] f32, 1 GB
    kth = []
    for qi in range(b):
        r, s = unpack_keys(hk[qi])
        _property_check(torch, rows, qs[qi], r, s, k, planted.get(qi))
        kth.append(float(s[-1]))
    beat = (allsc > (torch.tensor(kth, device="cuda") + 2e-6)[None, :]).sum(dim=0).cpu().numpy()
    assert np.all(beat <= k - 1), beat.max()
    del allsc
    # the same queries one at a time (HBM-streaming kernel): identical ids wherever scores are > 2e-6 apart
    k1 = torch.zeros((1, k), dtype=torch.int64, device="cuda"); c1 = torch.zeros((1,), dtype=torch.int32, device="cuda")
    for qi in (0, 100, 255, 17):
        idx.search_device(qs[qi].data_ptr(), 1, k, k1.data_ptr(), c1.data_ptr(), stream=st)
        torch.cuda.synchronize()
        r1, s1 = unpack_keys(k1.cpu().numpy().view(np.uint64)[0])
        rb, sb = unpack_keys(hk[qi])
        assert np.max(np.abs(s1 - sb)) <= 2e-6
        gaps = np.abs(np.diff(sb)) > 4e-6
        if np.all(gaps):
            assert list(r1) == list(rb)
        else:
            assert set(r1[:k - 1]) <= set(rb) | set(r1) and len(set(r1) & set(rb)) >= k - 2
    idx.close()


def test_full_size_properties_10m_and_8_shards(hip):
    """BASELINE configs[4] size on ONE GPU: 10M x 768 fp32 (30.7 GB) through the persistent-grid scan, k = 20 and
    k = 500; properties as above + the corpus cut into 8 row shards with row_base, per-shard top-k merged on the
    host == the whole-corpus answer, key for key (what the 8-GPU all-gather + merge computes)."""
    import torch
    n, dim = 10_000_000, 768
    free, _total = torch.cuda.mem_get_info()
    if free < 40 * (1 << 30):
        pytest.skip("needs ~36 GB of free HBM")
    g = torch.Generator(device="cuda"); g.manual_seed(777)
    rows = torch.empty((n, dim), device="cuda", dtype=torch.float32)
    for lo in range(0, n, 1 << 18):
        hi = min(n, lo + (1 << 18))
        x = torch.randn((hi - lo, dim), generator=g, device="cuda"); x /= x.norm(dim=1, keepdim=True); rows[lo:hi] = x
    q = torch.randn((dim,), generator=g, device="cuda"); q /= q.norm()
    planted = [9_999_999, 0, 1_250_000, 1_249_999, 8_750_001]      # ends of the corpus and both sides of shard cuts
    for j, r in enumerate(planted):
        noise = torch.randn((dim,), generator=g, device="cuda")
        noise -= (noise @ q) * q; noise /= noise.norm()
        c = 0.9 - 0.05 * j
        rows[r] = c * q + (1 - c * c) ** 0.5 * noise
    idx = HipIndex.build_from_device(None, rows.data_ptr(), n, dim, borrow=True, keepalive=rows)
    st = torch.cuda.current_stream().cuda_stream
    allsc = rows @ q
    whole = {}
    for k in (20, 500):
        keys = torch.zeros((1, k), dtype=torch.int64, device="cuda"); cnt = torch.zeros((1,), dtype=torch.int32, device="cuda")
        idx.search_device(q.data_ptr(), 1, k, keys.data_ptr(), cnt.data_ptr(), stream=st)
        torch.cuda.synchronize()
        assert cnt.item() == k
        hk = keys.cpu().numpy().view(np.uint64)[0]
        r, s = unpack_keys(hk)
        _property_check(torch, rows, q, r, s, k, planted)
        assert int((allsc > float(s[-1]) + 2e-6).sum().item()) <= k - 1
        whole[k] = hk
    idx.close()
    k, shards = 20, 8
    per = n // shards
    ks = torch.zeros((shards, k), dtype=torch.int64, device="cuda"); cs = torch.zeros((shards,), dtype=torch.int32, device="cuda")
    parts = []
    for sidx in range(shards):
        lo = sidx * per
        part = HipIndex.build_from_device(None, rows[lo:].data_ptr(), per, dim, borrow=True, row_base=lo)
        part.search_device(q.data_ptr(), 1, k, ks[sidx].data_ptr(), cs[sidx:].data_ptr(), stream=st)
        parts.append(part)
    torch.cuda.synchronize()
    merged = merge_keys(ks.cpu().numpy().view(np.uint64), cs.cpu().numpy().astype(np.uint32), k)
    assert list(merged) == list(whole[20])
    for p in parts:
        p.close()


def test_two_streams_share_one_handle(hip, oracle):
    """include/cqs_hip.h: searches on one handle share one scratch; the handle orders searches enqueued on
    DIFFERENT streams (event wait), so back-to-back device-API calls on two streams and a host-API call right
    behind them all return the right answers (this used to race silently)."""
    import torch
    n, k = 300_000, 20
    rows = synth.gaussian_unit(n, seed=2300)
    qs = synth.gaussian_unit(6, seed=2301)
    d_rows = torch.from_numpy(rows).cuda()
    idx = HipIndex.build_from_device(None, d_rows.data_ptr(), n, 768, borrow=True, keepalive=d_rows)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    d_q = torch.from_numpy(qs).cuda()
    keys = torch.zeros((6, k), dtype=torch.int64, device="cuda")
    counts = torch.zeros((6,), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for rep in range(3):
        for i in range(6):
            st = (s1, s2)[i & 1]
            idx.search_device(d_q[i].data_ptr(), 1, k, keys[i].data_ptr(), counts[i:].data_ptr(), stream=st.cuda_stream)
        hr, hs, hc = idx.search_batch(qs[0], k)        # host API on the handle's own stream, right behind
        torch.cuda.synchronize()
        hk = keys.cpu().numpy().view(np.uint64)
        for i in range(6):
            r, s = unpack_keys(hk[i])
            ext_ids, ext_scores = oracle.index_search(rows, qs[i], k + MARGIN)
            assert_topk_parity(r, s, ext_ids, ext_scores, k)
        ext_ids, ext_scores = oracle.index_search(rows, qs[0], k + MARGIN)
        assert_topk_parity(hr[0, :hc[0]], hs[0, :hc[0]], ext_ids, ext_scores, k)
    idx.close()


def test_debug_stamps_mode_with_filter_and_k500(hip, oracle, monkeypatch):
    """CQS_HIP_DEBUG_STAMPS=1 (diagnostic build of the handle): a filtered search used to free the stamp buffer
    (stray hipFree in the bitset-grow branch) and later searches wrote through the dangling pointer."""
    monkeypatch.setenv("CQS_HIP_DEBUG_STAMPS", "1")
    n = 20_000
    rows = synth.gaussian_unit(n, seed=2400)
    q = synth.gaussian_unit(1, seed=2401)[0]
    idx = HipIndex.build_from_flat(None, rows)
    rng = np.random.default_rng(24)
    keep = _bits(rng.random(n) < 0.5)
    check(oracle, idx, rows, q, 20, keep=keep)
    check(oracle, idx, rows, q, 500)
    check(oracle, idx, rows, q, 20, keep=keep)
    check(oracle, idx, rows, q, 20)
    idx.close()


# ---- A6: find_neighbors (src/cli/commands/search/neighbors.rs:86-132) ------------------------------------
@pytest.mark.parametrize("n,limit", [(3000, 5), (3000, 0), (3000, 1), (3000, 100), (3000, 500), (40, 100), (2, 10), (1, 10)])
def test_find_neighbors_matches_oracle(hip, oracle, n, limit):
    """Exact kNN of a stored row, itself excluded, limit clamped to [1, 100]; the oracle restates the reference's
    sequential-f32 dot + full sort (ids exact outside 2e-6 near-ties, scores to 1e-5)."""
    rows = synth.gaussian_unit(n, seed=3000 + n)
    idx = HipIndex.build_from_flat(None, rows)
    want = min(max(limit, 1), 100, n - 1)
    for t in sorted({0, n // 2, n - 1}):
        r, s = idx.neighbors_rows(t, limit)
        assert len(r) == want and t not in set(int(x) for x in r)
        if n > 1:
            # extended oracle list = the same restatement with a larger limit is capped at 100, so rank against
            # the exact index search (k + margin, target removed) and pin scores to the sequential-f32 sums
            ext_ids, ext_scores = oracle.index_search(rows, rows[t], min(n, want + MARGIN + 1), None, 0, 0.0, 2)
            sel = ext_ids != t
            assert_topk_parity(r, s, ext_ids[sel], ext_scores[sel], want)
            ref_ids, ref_scores = oracle.find_neighbors(rows, t, limit)
            assert len(ref_ids) == want
            assert np.max(np.abs(np.sort(ref_scores)[::-1] - s)) <= 1e-5
    idx.close()


def test_find_neighbors_duplicates_row_base_and_ids(hip, oracle):
    """Duplicates of the target tie with it and order by row asc (neighbors.rs:131); a sharded index addresses
    the target by its GLOBAL row; the trait-side mirror maps chunk ids; bad targets are errors."""
    base = synth.gaussian_unit(300, seed=3100)
    rows = np.concatenate([base[:10], np.tile(base[7], (150, 1)), base[10:]])      # rows 10..159 duplicate row 7
    idx = HipIndex.build_from_flat(None, rows)
    r, s = idx.neighbors_rows(100, 100)                                            # target inside the duplicate run
    ref_ids, ref_scores = oracle.find_neighbors(rows, 100, 100)
    assert list(r) == list(ref_ids) and 100 not in r
    assert list(r[:3]) == [7, 10, 11]
    r, s = idx.neighbors_rows(7, 3)
    assert list(r) == [10, 11, 12]
    idx.close()
    rows = synth.gaussian_unit(2000, seed=3101)
    ids = [f"src/m{i % 5}.rs:{i}:beef" for i in range(2000)]
    idx = HipIndex.build_from_flat(ids, rows, row_base=5000)
    res = idx.find_neighbors(ids[42], 10)
    ref_ids, ref_scores = oracle.find_neighbors(rows, 42, 10)
    assert [x.id for x in res] == [ids[int(i)] for i in ref_ids]
    assert np.allclose([x.score for x in res], ref_scores, atol=1e-5)
    with pytest.raises(KeyError):
        idx.find_neighbors("nope", 5)
    with pytest.raises(Exception):
        idx.neighbors_rows(42, 5)                    # local row id on a row_base index: not in [5000, 7000)
    r, s = idx.neighbors_rows(5042, 5)
    assert list(r) == [5000 + int(i) for i in ref_ids[:5]]
    idx.close()
