"""HIP BERT-family engines (SPLADE masked-LM encoder, cross-encoder reranker) against the fp32 oracle
(oracle/bert_ref.py, pinned to transformers in tests/test_bert_oracle.py) through the C ABI.  Weights are seeded and
bf16-exact, so the only differences are bf16 activations / f32 accumulation order.  The reference holds no golden
logits for either model: numerics are "parity unpinned" against the reference itself."""
import numpy as np
import pytest

from oracle import bert_ref as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from cqs_amd import _lib
    _lib.load()
    import torch
    assert torch.cuda.is_available()
    return True


def _engine(cfg, head, seed):
    from cqs_amd import _lib
    from cqs_amd.splade import HipBertEngine, bert_config
    kind = _lib.BERT_HEAD_MLM if head == "mlm" else _lib.BERT_HEAD_CLASSIFIER
    c = bert_config(kind, vocab_size=cfg.vocab_size, hidden=cfg.hidden, layers=cfg.layers, heads=cfg.heads,
                    intermediate=cfg.intermediate, max_pos=cfg.max_pos, type_vocab=cfg.type_vocab,
                    num_labels=cfg.num_labels, ln_eps=cfg.ln_eps)
    eng = HipBertEngine(c)
    w = R.seeded_weights(cfg, head, seed=seed)
    eng.set_weights(w)
    return eng, w


def _seqs(cfg, lens, seed):
    rng = np.random.default_rng(seed)
    return [rng.integers(1, cfg.vocab_size, size=n).astype(np.int32) for n in lens]


def _padded(seqs, types=None):
    L = max(len(s) for s in seqs)
    ids = np.zeros((len(seqs), L), np.int64)
    mask = np.zeros((len(seqs), L), np.int64)
    tt = np.zeros((len(seqs), L), np.int64)
    for i, s in enumerate(seqs):
        ids[i, :len(s)] = s
        mask[i, :len(s)] = 1
        if types is not None:
            tt[i, :len(s)] = types[i]
    return ids, mask, tt


def cos(a, b):
    return float(np.dot(a, b) / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))


@pytest.mark.parametrize("heads", [6, 12])          # head dim 64 (BERT-base) and 32 (MiniLM)
def test_encoder_hidden_states(hip, heads):
    cfg = R.BertConfig(vocab_size=1000, hidden=384, layers=3, heads=heads, intermediate=768, max_pos=300)
    eng, w = _engine(cfg, "mlm", seed=1)
    lens = [70, 1, 64, 65, 129, 300, 17]
    seqs = _seqs(cfg, lens, seed=2)
    types = [np.r_[np.zeros(n // 2, np.int32), np.ones(n - n // 2, np.int32)] for n in lens]
    got = eng.hidden(seqs, types)
    ids, mask, tt = _padded(seqs, types)
    ref = R.encode(cfg, w, ids, mask, tt).numpy()
    m = 0
    for i, n in enumerate(lens):
        g, r = got[m:m + n], ref[i, :n]
        err = np.abs(g - r)
        assert err.mean() / np.abs(r).mean() < 0.02 and err.max() < 0.25, (heads, i, err.mean(), err.max())
        assert cos(g.ravel(), r.ravel()) > 0.999
        m += n
    eng.close()


def test_splade_sparse_vectors(hip):
    """`encode_batch`: activations against the oracle, and the sparse vectors the threshold leaves: every strong
    weight present with the same id, weights within bf16 noise, ascending ids, nothing at or below the threshold."""
    from cqs_amd.splade import SpladeEncoder
    cfg = R.BertConfig(vocab_size=1531, hidden=384, layers=2, heads=6, intermediate=768, max_pos=128)
    eng, w = _engine(cfg, "mlm", seed=3)
    lens = [40, 7, 128, 64, 1, 0, 90]
    seqs = _seqs(cfg, lens, seed=4)
    thr = 0.3
    enc = SpladeEncoder(eng, threshold=thr, max_seq_len=128)
    got = enc.encode_batch(seqs)
    dense = eng.splade_dense(seqs)
    live = [s for s in seqs if len(s)]
    ids, mask, _ = _padded(live)
    want, want_dense = R.splade_encode_batch(cfg, w, ids, mask, thr)
    j = 0
    for i, s in enumerate(seqs):
        if len(s) == 0:
            assert got[i] == [] and np.all(dense[i] == 0)                  # empty sequence: every pooled value -inf -> ln(1) = 0
            continue
        d, wd = dense[i], want_dense[j]
        assert d.shape == (cfg.vocab_size,) and np.all(np.isfinite(d)) and np.all(d >= 0)
        assert np.max(np.abs(d - wd)) < 0.06 and cos(d, wd) > 0.999, (i, np.max(np.abs(d - wd)))
        g = dict(got[i])
        assert [k for k, _ in got[i]] == sorted(g) and all(v > thr for v in g.values())
        for tok, wt in want[j]:
            if wt > thr + 0.08:                                             # clear of the threshold: must be present
                assert tok in g and abs(g[tok] - wt) < 0.06, (i, tok, wt, g.get(tok))
        for tok, wt in g.items():
            assert wd[tok] > thr - 0.08                                      # nothing far below the threshold got in
        j += 1
    assert enc.encode(seqs[0]) == got[0]
    eng.close()


def test_splade_padding_free_and_batch_invariant(hip):
    cfg = R.BertConfig(vocab_size=1000, hidden=384, layers=2, heads=12, intermediate=768, max_pos=128)
    eng, _ = _engine(cfg, "mlm", seed=5)
    seqs = _seqs(cfg, [33, 100, 5], seed=6)
    full = eng.splade_dense(seqs)
    for i, s in enumerate(seqs):
        assert np.array_equal(eng.splade_dense([s])[0], full[i])            # packed: neighbours do not matter
    assert np.array_equal(eng.splade_dense(seqs), full)                     # deterministic
    eng.close()


def test_reranker_scores(hip):
    from cqs_amd.splade import Reranker
    cfg = R.BertConfig(vocab_size=1000, hidden=384, layers=6, heads=12, intermediate=1536, max_pos=256, num_labels=1)
    eng, w = _engine(cfg, "classifier", seed=7)
    lens = [60, 200, 12, 256, 33]
    seqs = _seqs(cfg, lens, seed=8)
    types = [np.r_[np.zeros(10, np.int32), np.ones(n - 10, np.int32)] for n in lens]    # query | passage
    got = Reranker(eng, max_length=256).scores(seqs, types)
    ids, mask, tt = _padded(seqs, types)
    want = R.rerank_scores(cfg, w, ids, mask, tt)
    assert got.shape == (5,) and np.all((got > 0) & (got < 1))
    assert np.max(np.abs(got - want)) < 0.02, (got, want)
    logits = eng.rerank_logits(seqs, types)[:, 0]
    ref_logits = R.classifier_logits(cfg, w, R.encode(cfg, w, ids, mask, tt))[:, 0]
    assert np.max(np.abs(logits - ref_logits)) < 0.05 * max(1.0, float(np.abs(ref_logits).max()))
    assert np.array_equal(np.argsort(-got), np.argsort(-want)) or np.max(np.abs(got - want)) < 0.005
    eng.close()


def test_bert_io_contract(hip):
    from cqs_amd import _lib
    from cqs_amd.splade import BertError, HipBertEngine, bert_config
    cfg = R.BertConfig(vocab_size=500, hidden=384, layers=1, heads=6, intermediate=384, max_pos=64)
    eng, _ = _engine(cfg, "mlm", seed=9)
    with pytest.raises(BertError):
        eng.splade_dense([np.array([1, 2, 500], np.int32)])                 # token id out of range
    with pytest.raises(BertError):
        eng.splade_dense([np.arange(1, 70, dtype=np.int32)])                # longer than max_position_embeddings
    with pytest.raises(BertError):
        eng.rerank_logits([np.array([1, 2], np.int32)], None)               # wrong head
    assert eng.splade_dense([np.array([1, 2, 3], np.int32)]).shape == (1, 500)      # still usable
    eng.close()
    with pytest.raises(BertError):
        HipBertEngine(bert_config(_lib.BERT_HEAD_MLM, hidden=100))          # geometry the kernels do not cover
    e2 = HipBertEngine(bert_config(_lib.BERT_HEAD_CLASSIFIER, vocab_size=500, layers=1, max_pos=64))
    with pytest.raises(BertError):
        e2.set_weights({"embeddings.word_embeddings.weight": np.zeros((500, 384), np.float32)})   # incomplete
    e2.close()


def test_full_geometry_presets(hip):
    """The two presets at their real dims (BERT-base masked-LM with the 30 522-word tied decoder, padded to the GEMM's
    tile; MiniLM-L6 classifier) on a small batch against the oracle."""
    from cqs_amd.splade import Reranker, SpladeEncoder
    cfg = R.splade_base()
    eng, w = _engine(cfg, "mlm", seed=11)
    seqs = _seqs(cfg, [90, 300, 12], seed=12)
    dense = eng.splade_dense(seqs)
    ids, mask, _ = _padded(seqs)
    _, want = R.splade_encode_batch(cfg, w, ids, mask, 0.01)
    assert dense.shape == (3, 30522)
    for i in range(3):
        assert np.max(np.abs(dense[i] - want[i])) < 0.08 and cos(dense[i], want[i]) > 0.999, (i, np.max(np.abs(dense[i] - want[i])))
    top_g, top_w = np.argsort(-dense[1])[:50], np.argsort(-want[1])[:50]
    assert len(set(top_g) & set(top_w)) >= 45                              # the heavy vocabulary entries agree
    sv = SpladeEncoder(eng, threshold=1.0).encode_batch(seqs)
    assert all(len(v) > 0 and all(wt > 1.0 for _, wt in v) for v in sv)
    eng.close()
    cfg = R.minilm_l6()
    eng, w = _engine(cfg, "classifier", seed=13)
    seqs = _seqs(cfg, [128, 512, 40, 77], seed=14)
    types = [np.r_[np.zeros(16, np.int32), np.ones(len(s) - 16, np.int32)] for s in seqs]
    got = Reranker(eng).scores(seqs, types)
    ids, mask, tt = _padded(seqs, types)
    want = R.rerank_scores(cfg, w, ids, mask, tt)
    assert np.max(np.abs(got - want)) < 0.02, (got, want)
    eng.close()
