"""GPU tests of the single-process row-sharded index (`cqs_hip_index_create_sharded`, include/cqs_hip.h): the
multi-GPU path of BASELINE configs[4] behind the same handle the single-device index uses.  A dev box has ONE
GPU, so the device list names device 0 several times (split / per-shard scan with global row ids / gather /
host merge all run; the gather uses device-to-device copies because one device cannot form an RCCL clique), and
`devices=[0]` goes through RCCL itself (ncclCommInitAll + grouped ncclAllGather with one rank).  Every answer
is checked against the CPU oracle AND against the single-device index key for key."""
import numpy as np
import pytest

from cqs_amd import BackendContext, HipBackend, HipIndex, _lib, synth
from parity import assert_topk_parity
from test_scan_gpu import MARGIN, _FakeStore, _bits, check

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0, 0, 0, 0, 0, 0]])
def test_sharded_matches_oracle_and_single(hip, oracle, devices):
    n = 20_011
    rows = synth.gaussian_unit(n, seed=4000)
    qs = synth.gaussian_unit(5, seed=4001)
    for j, r in enumerate((0, 2559, 2560, 2561, n - 1)):            # around the 256-row-aligned shard cuts
        v = qs[0] + 0.05 * (j + 1) * rows[r]
        rows[r] = v / np.linalg.norm(v)
    single = HipIndex.build_from_flat(None, rows)
    sh = HipIndex.build_sharded(None, rows, devices)
    info = sh.shards()
    assert len(info) == len(devices) and sum(x[2] for x in info) == n and len(sh) == n
    assert all(x[1] % 256 == 0 for x in info if x[2])
    assert [x[3] for x in info] == [len(set(devices)) == len(devices)] * len(devices)   # RCCL only for distinct devices
    assert sh.dim() == 768 and sh.max_k() == 1024 and not sh.is_poisoned()
    for k in (1, 20, 500):
        check(oracle, sh, rows, qs, k)
        r1, s1, c1 = single.search_batch(qs, k)
        r2, s2, c2 = sh.search_batch(qs, k)
        assert np.array_equal(c1, c2) and np.max(np.abs(s1 - s2)) <= 2e-6      # task plans differ with shard size
        for i in range(len(qs)):
            if np.all(np.abs(np.diff(s1[i, :c1[i]])) > 4e-6):
                assert np.array_equal(r1[i], r2[i])
    rng = np.random.default_rng(40)
    keep = rng.random(n) < 0.3
    keep[2560:5120] = False                                           # one whole shard (8-way split) filtered out
    check(oracle, sh, rows, qs, 20, keep=_bits(keep))
    few = np.zeros(n, bool); few[[5, 2600, n - 2]] = True             # k capped at the kept rows, spread over shards
    r, s, c = sh.search_batch(qs[0], 20, keep_bitset=_bits(few))
    assert c[0] == 3 and set(int(x) for x in r[0, :3]) == {5, 2600, n - 2}
    check(oracle, sh, rows, qs, 30, mode=_lib.MODE_PIPELINE, thr=0.05)
    bad = qs.copy(); bad[2, 7] = np.nan
    r, s, c = sh.search_batch(bad, 10)
    assert list(c) == [10, 10, 0, 10, 10]
    assert sh.search(qs[0][:100], 5) == [] and sh.search(qs[0], 0) == []
    # 40 queries: the matrix-core path on every shard
    q40 = synth.gaussian_unit(40, seed=4002)
    check(oracle, sh, rows, q40, 20)
    single.close(); sh.close()


def test_sharded_neighbors_extend_persistence(hip, oracle, tmp_path):
    n = 9000
    rows = synth.gaussian_unit(n + 700, seed=4100)
    ids = [f"src/s{i % 11}.rs:{i}:cafe" for i in range(n + 700)]
    sh = HipIndex.build_sharded(ids[:n], rows[:n], [0, 0, 0])
    for t in (0, 3100, n - 1):                                        # targets on different shards
        r, s = sh.neighbors_rows(t, 10)
        ref_ids, ref_scores = oracle.find_neighbors(rows[:n], t, 10)
        assert list(r) == list(ref_ids) and np.allclose(s, ref_scores, atol=1e-5)
    assert [x.id for x in sh.find_neighbors(ids[42], 5)] == [ids[int(i)] for i in oracle.find_neighbors(rows[:n], 42, 5)[0]]
    sh.extend(ids[n:], rows[n:])                                      # appended to the last shard
    assert len(sh) == n + 700 and sum(x[2] for x in sh.shards()) == n + 700
    q = rows[n + 650]
    assert sh.search(q, 3)[0].id == ids[n + 650]
    check_ids = sh.search_batch(synth.gaussian_unit(1, seed=4101)[0], 20)
    ext_ids, ext_scores = oracle.index_search(rows, synth.gaussian_unit(1, seed=4101)[0], 20 + MARGIN)
    assert_topk_parity(check_ids[0][0, :check_ids[2][0]], check_ids[1][0, :check_ids[2][0]], ext_ids, ext_scores, 20)
    # one blob for all shards; both loaders read it
    path = str(tmp_path / "index.hipflat")
    sh.save(path)
    a = HipIndex.load(path, 768, n + 700)
    b = HipIndex.load(path, 768, n + 700, devices=[0, 0])
    for idx in (a, b):
        assert [x.id for x in idx.search(q, 5)] == [x.id for x in sh.search(q, 5)]
        idx.close()
    sh.save(path)                                                     # overwrite: .bak rollback path, no leftovers
    assert not (tmp_path / "index.hipflat.bak").exists() and not (tmp_path / "index.hipflat.tmp").exists()
    (tmp_path / "index.hipflat.bak").write_bytes(b"stale")
    with pytest.raises(Exception):
        sh.save(path)                                                 # stale .bak refuses (src/cagra.rs:1485-1493)
    sh.close()
    # the backend with a device list
    be = HipBackend()
    store = _FakeStore(ids[:n], rows[:n])
    idx = be.try_open(BackendContext(str(tmp_path / "be"), store, hip_threshold=1000, devices=[0, 0], persist=False))
    assert idx is not None and len(idx.shards()) == 2 and idx.search(rows[77], 1)[0].id == ids[77]
    idx.close()


def test_sharded_tiny_corpora_leave_shards_empty(hip, oracle):
    for n in (1, 255, 257, 700):
        rows = synth.gaussian_unit(n, seed=4200 + n)
        q = synth.gaussian_unit(1, seed=4201)[0]
        sh = HipIndex.build_sharded(None, rows, [0, 0, 0, 0])
        assert len(sh) == n
        check(oracle, sh, rows, q, min(20, n))
        sh.close()
    with pytest.raises(Exception):
        HipIndex.build_sharded(None, synth.gaussian_unit(10, seed=1), [])
    with pytest.raises(Exception):
        HipIndex.build_sharded(None, synth.gaussian_unit(10, seed=1), [99])
