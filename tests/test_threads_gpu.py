"""The header promises that calls on one handle may come from several host threads (include/cqs_hip.h, threading):
each handle serialises its entry points internally.  Two / three threads hammer one index, one embedding engine and one
BERT engine; every result must equal the single-threaded answer bit for bit."""
import threading

import numpy as np
import pytest

from cqs_amd import HipIndex, synth

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
