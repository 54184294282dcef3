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
    kind = {"mlm": _lib.BERT_HEAD_MLM, "classifier": _lib.BERT_HEAD_CLASSIFIER, "none": _lib.BERT_HEAD_NONE}[head]
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


@pytest.mark.parametrize("heads", [4, 8])            # head dim 64 and 32
def test_resident_key_attention_at_the_block_edges(hip, monkeypatch, heads):
    """Sequences of up to 512 tokens take the resident-key attention kernel (one workgroup per (sequence, head) or
    per part of its queries; 128-key groups; 8 or 16 waves x 16 queries per pass): lengths around every edge it has,
    in a batch that needs four groups and in one that needs two, with the query split forced to 1, 2, 4 and chosen by
    the launcher - against the oracle and against the first-generation kernel (CQS_HIP_BERT_ATTN_RESIDENT=0, read per
    batch; same products, another order of the softmax carries, bf16 row sums: close, not bit-identical)."""
    cfg = R.BertConfig(vocab_size=500, hidden=256, layers=2, heads=heads, intermediate=512, max_pos=512)
    eng, w = _engine(cfg, "none", seed=11)
    for lens in ([1, 15, 16, 17, 127, 128, 129, 255, 256, 257, 300, 384, 385, 512, 33],
                 [256, 1, 129, 255, 16, 128, 200, 31]):
        seqs = _seqs(cfg, lens, seed=12)
        got = eng.hidden(seqs, None)
        assert np.array_equal(got, eng.hidden(seqs, None))     # deterministic
        for q in ("1", "2", "4"):
            monkeypatch.setenv("CQS_HIP_BERT_ATTN_QSPLIT", q)
            assert np.array_equal(got, eng.hidden(seqs, None)), q      # which workgroup takes a query changes nothing
        monkeypatch.delenv("CQS_HIP_BERT_ATTN_QSPLIT")
        monkeypatch.setenv("CQS_HIP_BERT_ATTN_RESIDENT", "0")
        old = eng.hidden(seqs, None)
        monkeypatch.delenv("CQS_HIP_BERT_ATTN_RESIDENT")
        assert not np.array_equal(got, old)                    # (the switch does select another kernel)
        ids, mask, tt = _padded(seqs)
        ref = R.encode(cfg, w, ids, mask, tt).numpy()
        m = 0
        for i, n in enumerate(lens):
            g, o, r = got[m:m + n], old[m:m + n], ref[i, :n]
            assert np.isfinite(g).all()
            err = np.abs(g - r)
            assert err.mean() / np.abs(r).mean() < 0.02 and err.max() < 0.25, (n, err.mean(), err.max())
            assert cos(g.ravel(), r.ravel()) > 0.999, n
            assert cos(g.ravel(), o.ravel()) > 0.9995, n
            # no further from the oracle than the first-generation kernel (plus noise)
            assert err.mean() <= np.abs(o - r).mean() * 1.25 + 1e-4, (n, err.mean(), np.abs(o - r).mean())
            m += n
    eng.close()


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_resident_key_attention_random_batches(hip, monkeypatch, seed):
    """Random ragged batches (empty sequences, lengths to 512, 1-12 sequences, head dims 64 and 32): the resident-key
    attention kernel against the first-generation one on the same engine - finite, deterministic, cosine >= 0.9995 per
    sequence, and equally close to each other whatever the query split."""
    rng = np.random.default_rng(1000 + seed)
    heads = int(rng.choice([4, 8]))
    cfg = R.BertConfig(vocab_size=400, hidden=256, layers=1, heads=heads, intermediate=256, max_pos=512)
    eng, _ = _engine(cfg, "none", seed=50 + seed)
    for _ in range(4):
        nseq = int(rng.integers(1, 13))
        top = int(rng.choice([40, 130, 256, 300, 512]))
        lens = [int(x) for x in rng.integers(0, top + 1, size=nseq)]
        if sum(lens) == 0:
            lens[0] = 3
        seqs = _seqs(cfg, lens, seed=int(rng.integers(1, 10_000)))
        got = eng.hidden(seqs, None)
        assert np.isfinite(got).all() and np.array_equal(got, eng.hidden(seqs, None))
        monkeypatch.setenv("CQS_HIP_BERT_ATTN_QSPLIT", str(int(rng.choice([1, 2, 4]))))
        assert np.array_equal(got, eng.hidden(seqs, None))
        monkeypatch.delenv("CQS_HIP_BERT_ATTN_QSPLIT")
        monkeypatch.setenv("CQS_HIP_BERT_ATTN_RESIDENT", "0")
        old = eng.hidden(seqs, None)
        monkeypatch.delenv("CQS_HIP_BERT_ATTN_RESIDENT")
        m = 0
        for n in lens:
            if n:
                assert cos(got[m:m + n].ravel(), old[m:m + n].ravel()) > 0.9995, (heads, lens, n)
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
    # `encode` = a batch of one: 40 tokens alone take the search-time GEMM kernels (another f32 association than the
    # batch's kernels): the same vector up to bf16 noise - ids equal away from the threshold, weights within 0.01
    one, many = dict(enc.encode(seqs[0])), dict(got[0])
    for tok in set(one) | set(many):
        if tok in one and tok in many:
            assert abs(one[tok] - many[tok]) < 0.01, (tok, one[tok], many[tok])
        else:
            assert abs((one.get(tok) or many.get(tok)) - thr) < 0.02, (tok, one.get(tok), many.get(tok))
    import os
    os.environ["CQS_HIP_GEMM_SMALL_ROWS"] = "0"
    try:
        assert enc.encode(seqs[0]) == got[0]                                    # the same kernels: the same bits
    finally:
        del os.environ["CQS_HIP_GEMM_SMALL_ROWS"]
    eng.close()


def test_splade_ids_at_the_default_threshold(hip):
    """the pooled weight is the max over F32 logits (the decoder GEMM's ROWMAX epilogue reduces the f32
    accumulators, not a bf16 copy of them).  At the reference's default threshold 0.01 (src/splade/mod.rs) the id sets
    of HIP and oracle may differ only where the oracle's weight lies within the forward's bf16-operand error of the
    threshold; the count of such flips is printed."""
    from cqs_amd.splade import SpladeEncoder
    cfg = R.BertConfig(vocab_size=1531, hidden=384, layers=2, heads=6, intermediate=768, max_pos=128)
    eng, w = _engine(cfg, "mlm", seed=13)
    seqs = _seqs(cfg, [64, 31, 128, 9], seed=14)
    thr = 0.01
    got = SpladeEncoder(eng, threshold=thr, max_seq_len=128).encode_batch(seqs)
    ids, mask, _ = _padded(seqs)
    want, want_dense = R.splade_encode_batch(cfg, w, ids, mask, thr)
    flips = total = 0
    for i in range(len(seqs)):
        g, o = {t for t, _ in got[i]}, {t for t, _ in want[i]}
        total += len(o)
        for tok in g ^ o:
            flips += 1
            assert abs(want_dense[i][tok] - thr) < 0.06, (i, tok, want_dense[i][tok])     # the forward's own tolerance
    print(f"ids differing from the oracle at threshold {thr}: {flips} of {total}")
    assert flips <= max(8, total // 10)
    eng.close()


def test_splade_padding_free_and_batch_invariant(hip, monkeypatch):
    """Packed tokens: a sequence's activations do not depend on its neighbours in the batch - bit for bit as long as the
    projections take the same kernels.  Batches of up to 64 tokens take the search-time GEMM kernels (K split over a
    workgroup's waves: another f32 association), so a short sequence ALONE is compared bit for bit with that path
    switched off (CQS_HIP_GEMM_SMALL_ROWS=0, read per call) and to cosine 0.9999 / 2 % of the largest activation with it."""
    cfg = R.BertConfig(vocab_size=1000, hidden=384, layers=2, heads=12, intermediate=768, max_pos=128)
    eng, _ = _engine(cfg, "mlm", seed=5)
    seqs = _seqs(cfg, [33, 100, 5], seed=6)
    full = eng.splade_dense(seqs)
    assert np.array_equal(eng.splade_dense(seqs), full)                     # deterministic
    assert np.array_equal(eng.splade_dense([seqs[1]])[0], full[1])          # 100 tokens alone: the same kernels
    for i in (0, 2):
        alone = eng.splade_dense([seqs[i]])[0]
        assert np.array_equal(eng.splade_dense([seqs[i]])[0], alone)
        assert cos(alone, full[i]) > 0.9999 and np.max(np.abs(alone - full[i])) < 0.02 * np.abs(full[i]).max() + 1e-3
    monkeypatch.setenv("CQS_HIP_GEMM_SMALL_ROWS", "0")
    for i, s in enumerate(seqs):
        assert np.array_equal(eng.splade_dense([s])[0], full[i])            # packed: neighbours do not matter
    monkeypatch.delenv("CQS_HIP_GEMM_SMALL_ROWS")
    eng.close()


@pytest.mark.parametrize("hidden,heads,inter", [(768, 12, 3072), (384, 12, 1536), (1024, 16, 4096)])
def test_small_batches_through_the_search_time_gemms(hip, monkeypatch, hidden, heads, inter):
    """A batch of up to 64 tokens (a SPLADE query, one short passage) runs its projections through the search-time GEMM
    kernels (`launch_gemm_small_rows`: bias + erf-GELU epilogue; staged operands for K in {384, 768, 3072}, the gather form
    for other K): token counts around the row-tile edges, BERT-base / MiniLM / BERT-large widths, against the oracle and
    against the one-wave-per-tile kernels (CQS_HIP_GEMM_SMALL_ROWS=0)."""
    cfg = R.BertConfig(vocab_size=700, hidden=hidden, layers=2, heads=heads, intermediate=inter, max_pos=128)
    eng, w = _engine(cfg, "none", seed=21)
    for lens in ([1], [8], [16], [17], [33], [48], [49], [64], [20, 30, 14], [3, 2]):
        seqs = _seqs(cfg, lens, seed=22 + sum(lens))
        got = eng.hidden(seqs, None)
        monkeypatch.setenv("CQS_HIP_GEMM_SMALL_ROWS", "0")
        old = eng.hidden(seqs, None)
        monkeypatch.delenv("CQS_HIP_GEMM_SMALL_ROWS")
        ids, mask, tt = _padded(seqs)
        ref = R.encode(cfg, w, ids, mask, tt).numpy()
        m = 0
        for i, n in enumerate(lens):
            g, o, r = got[m:m + n], old[m:m + n], ref[i, :n]
            assert np.isfinite(g).all()
            err = np.abs(g - r)
            assert err.mean() / np.abs(r).mean() < 0.02 and err.max() < 0.25, (lens, i, err.mean(), err.max())
            assert cos(g.ravel(), r.ravel()) > 0.999 and cos(g.ravel(), o.ravel()) > 0.9995, (lens, i)
            m += n
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


def _bert_cfg(cfg, head):
    from cqs_amd import _lib
    from cqs_amd.splade import bert_config
    kind = {"mlm": _lib.BERT_HEAD_MLM, "classifier": _lib.BERT_HEAD_CLASSIFIER, "none": _lib.BERT_HEAD_NONE}[head]
    return bert_config(kind, vocab_size=cfg.vocab_size, hidden=cfg.hidden, layers=cfg.layers, heads=cfg.heads,
                       intermediate=cfg.intermediate, max_pos=cfg.max_pos, type_vocab=cfg.type_vocab,
                       num_labels=cfg.num_labels, ln_eps=cfg.ln_eps)


def test_load_dir_safetensors_checkpoint(hip, tmp_path):
    """`cqs_hip_bert_load_dir` on a Hugging Face checkpoint: `bert.`-prefixed names, the tied decoder present twice,
    an int64 buffer that is not a weight, one tensor stored as f16."""
    import torch
    from safetensors.torch import save_file
    from cqs_amd.splade import HipBertEngine
    cfg = R.BertConfig(vocab_size=700, hidden=384, layers=2, heads=6, intermediate=768, max_pos=96)
    eng, w = _engine(cfg, "mlm", seed=21)
    body = {k: v for k, v in R.hf_state_dict(cfg, w, "mlm").items()}
    body = {k: v.clone() for k, v in body.items()}
    body["bert.embeddings.position_ids"] = torch.arange(cfg.max_pos, dtype=torch.int64)[None]
    body["bert.encoder.layer.1.output.dense.weight"] = body["bert.encoder.layer.1.output.dense.weight"].to(torch.float16)
    save_file(body, str(tmp_path / "model.safetensors"), metadata={"format": "pt"})
    eng2 = HipBertEngine.load_dir(str(tmp_path), _bert_cfg(cfg, "mlm"))
    seqs = _seqs(cfg, [50, 9, 96], seed=22)
    a, b = eng.splade_dense(seqs), eng2.splade_dense(seqs)
    assert np.max(np.abs(a - b)) < 0.03 and all(cos(a[i], b[i]) > 0.9999 for i in range(3))   # one weight went through f16
    eng.close(); eng2.close()
    # the same checkpoint opened as an EMBEDDER (encoder only): head tensors in the file are skipped
    eng3 = HipBertEngine.load_dir(str(tmp_path), _bert_cfg(cfg, "none"))
    ids, mask, tt = _padded(seqs)
    w16 = dict(w)
    want = R.pooled_embedding(cfg, w16, ids, mask, tt, "mean")
    got = eng3.embed(seqs, None, "mean")
    assert all(cos(got[i], want[i]) > 0.999 for i in range(3))
    eng3.close()


@pytest.mark.parametrize("flat", [False, True])
def test_load_dir_onnx_export(hip, tmp_path, flat):
    """`cqs_hip_bert_load_dir` on the bundle layout the reference uses (`{dir}/onnx/model.onnx`, src/reranker.rs:548-556):
    embeddings / LayerNorm / biases as named initialisers, `Linear` weights as anonymous TRANSPOSED MatMul operands found
    through the consuming node's module path, the pooler / classifier as named Gemm weights, big tensors in the external
    data sidecar.  Written by the byte-level encoder of tests/onnx_bytes.py (no `onnx` package in the image)."""
    import onnx_bytes as ob
    from cqs_amd.splade import HipBertEngine
    cfg = R.BertConfig(vocab_size=600, hidden=384, layers=2, heads=12, intermediate=768, max_pos=80, num_labels=1)
    eng, w = _engine(cfg, "classifier", seed=23)
    d = tmp_path if flat else tmp_path / "onnx"
    d.mkdir(parents=True, exist_ok=True)
    side = bytearray(b"\0" * 8)
    nodes, inits, n = [], [], [0]

    def ext(arr):
        raw = ob.encode_values(arr, ob.FLOAT)
        off = len(side)
        side.extend(raw)
        side.extend(b"\0" * ((-len(side)) % 16))
        return ("model.onnx_data", off, len(raw))

    for k, v in w.items():
        is_linear = k.endswith(".weight") and v.ndim == 2 and k.startswith("encoder.")
        if is_linear:
            n[0] += 1
            iname = f"onnx::MatMul_{2000 + n[0]}"
            path = "/bert/" + k[:-len(".weight")].replace(".", "/").replace("layer/", "layer.") + "/MatMul"
            t = np.ascontiguousarray(v.T)
            big = v.size > 200_000
            nodes.append(ob.node("MatMul", path, [f"h{n[0]}", iname], [f"o{n[0]}"]))
            inits.append(ob.tensor(iname, t, ob.FLOAT, "external" if big else "raw", external=ext(t) if big else None))
        elif k.startswith(("pooler.", "classifier.")):
            inits.append(ob.tensor(("bert." if k.startswith("pooler.") else "") + k, v))      # Gemm: [out, in], by name
        else:
            big = v.size > 200_000
            inits.append(ob.tensor("bert." + k, v, ob.FLOAT, "external" if big else "raw", external=ext(v) if big else None))
    inits.append(ob.tensor("onnx::Reshape_5", np.array([0, -1, 12, 32]), ob.INT64))
    (d / "model.onnx").write_bytes(ob.model(nodes, inits))
    (d / "model.onnx_data").write_bytes(bytes(side))
    eng2 = HipBertEngine.load_dir(str(tmp_path), _bert_cfg(cfg, "classifier"))
    seqs = _seqs(cfg, [40, 80, 5], seed=24)
    types = [np.r_[np.zeros(2, np.int32), np.ones(len(s) - 2, np.int32)] for s in seqs]
    assert np.array_equal(eng.rerank_logits(seqs, types), eng2.rerank_logits(seqs, types))
    eng.close(); eng2.close()
    (d / "model.onnx").write_bytes(b"\x00\x01garbage")
    from cqs_amd.splade import BertError
    with pytest.raises(BertError):
        HipBertEngine.load_dir(str(tmp_path), _bert_cfg(cfg, "classifier"))


def test_splade_device_side_threshold_filter(hip):
    """`cqs_hip_splade_encode_sparse` = the dense activations + the host filter, entry for entry; a row with more
    survivors than the cap reports its true count (and its first `cap` entries), an empty sequence reports 0."""
    from cqs_amd.splade import SpladeEncoder
    cfg = R.BertConfig(vocab_size=2500, hidden=384, layers=2, heads=6, intermediate=768, max_pos=128)
    eng, _ = _engine(cfg, "mlm", seed=31)
    seqs = _seqs(cfg, [40, 0, 128, 3, 77], seed=32)
    dense = eng.splade_dense(seqs)
    for thr in (0.05, 0.5, 1.2):
        ids, wts, cnt = eng.splade_sparse(seqs, thr, cap=4096)
        for b in range(len(seqs)):
            keep = np.nonzero(dense[b] > np.float32(thr))[0]
            assert cnt[b] == len(keep), (thr, b, cnt[b], len(keep))
            assert np.array_equal(ids[b, :cnt[b]], keep.astype(np.uint32)) and np.array_equal(wts[b, :cnt[b]], dense[b][keep])
    ids, wts, cnt = eng.splade_sparse(seqs, 0.05, cap=64)                   # cap smaller than the survivors
    keep0 = np.nonzero(dense[0] > np.float32(0.05))[0]
    assert cnt[0] == len(keep0) > 64 and np.array_equal(ids[0], keep0[:64].astype(np.uint32)) and cnt[1] == 0
    enc = SpladeEncoder(eng, threshold=0.05, sparse_cap=64)                  # the mirror falls back to the dense form for such rows
    got = enc.encode_batch_arrays(seqs)
    for b in range(len(seqs)):
        keep = np.nonzero(dense[b] > np.float32(0.05))[0]
        assert np.array_equal(got[b][0], keep.astype(np.uint32)) and np.array_equal(got[b][1], dense[b][keep])
    eng.close()


@pytest.mark.parametrize("hidden,heads,inter", [(384, 6, 768), (768, 12, 3072), (1024, 16, 1024)])
def test_bert_embedder_presets_pooling(hip, hidden, heads, inter):
    """The BERT-family EMBEDDER presets (e5-base / v9-200k: BERT-base; bge-large: BERT-large, hidden 1024): encoder +
    `mean_pool` / `cls_pool` (src/embedder/pooling.rs:87-128) on the device against the oracle; an empty sequence pools
    to zeros; results do not depend on the batch."""
    cfg = R.BertConfig(vocab_size=900, hidden=hidden, layers=2, heads=heads, intermediate=inter, max_pos=160)
    eng, w = _engine(cfg, "none", seed=41)
    lens = [50, 0, 160, 1, 97]
    seqs = _seqs(cfg, lens, seed=42)
    types = [np.zeros(n, np.int32) for n in lens]
    live = [i for i, n in enumerate(lens) if n]
    ids, mask, tt = _padded([seqs[i] for i in live], [types[i] for i in live])
    for pooling in ("mean", "cls"):
        got = eng.embed(seqs, types, pooling=pooling)
        want = R.pooled_embedding(cfg, w, ids, mask, tt, pooling)
        assert got.shape == (len(lens), hidden) and np.all(got[1] == 0)
        for j, i in enumerate(live):
            assert cos(got[i], want[j]) > 0.999 and np.max(np.abs(got[i] - want[j])) < 0.05 * np.abs(want[j]).max() + 0.02, (pooling, i)
        assert np.array_equal(eng.embed([seqs[2]], None, pooling=pooling)[0], got[2])       # alone == in the batch (type ids default 0)
    with pytest.raises(Exception):
        eng.splade_dense(seqs)                                                               # wrong head
    eng.close()


def test_bge_large_full_geometry(hip):
    cfg = R.bge_large()
    eng, w = _engine(cfg, "none", seed=43)
    seqs = _seqs(cfg, [120, 512, 30], seed=44)
    got = eng.embed(seqs, None, pooling="mean")
    ids, mask, tt = _padded(seqs)
    want = R.pooled_embedding(cfg, w, ids, mask, tt, "mean")
    assert got.shape == (3, 1024)
    for i in range(3):
        assert cos(got[i], want[i]) > 0.999, (i, cos(got[i], want[i]))
    eng.close()


def test_tickets_match_blocking_calls_and_release_slots(hip):
    """submit / collect of the BERT-family engines (the reference calls `SpladeEncoder::encode_batch` from
    its index pipeline, src/splade/mod.rs:774-1075): three tickets in flight on two execution contexts, collected out of
    order, equal the blocking call bit for bit; a 4th submit is refused; an abandoned or failed ticket gives its slot
    back; the pipelined encoder returns per-batch results in order."""
    from cqs_amd.splade import BertError, SpladeEncoder
    cfg = R.BertConfig(vocab_size=1531, hidden=384, layers=2, heads=6, intermediate=768, max_pos=128)
    eng, w = _engine(cfg, "mlm", seed=21)
    batches = [_seqs(cfg, lens, seed=30 + i) for i, lens in enumerate(([40, 7, 128], [64, 1, 0, 90], [5], [100, 100]))]
    thr = 0.3
    want = [eng.splade_sparse(b, thr, 512) for b in batches]
    hs = [eng.submit_sparse(b, thr, 512) for b in batches[:3]]
    with pytest.raises(BertError):
        eng.submit_sparse(batches[3], thr, 512)                      # every ticket slot in flight
    # ... but the BLOCKING calls have a slot of their own: a search-time query beside a pipelined index run
    ids3, wts3, cnt3 = eng.splade_sparse(batches[3], thr, 512)
    assert np.array_equal(cnt3, want[3][2])
    dense3 = eng.splade_dense(batches[3])
    assert dense3.shape[0] == len(batches[3]) and np.all(np.isfinite(dense3))
    for j in (2, 0, 1):
        ids, wts, cnt = eng.collect_sparse(hs[j])
        assert np.array_equal(cnt, want[j][2])
        for b in range(len(cnt)):
            assert np.array_equal(ids[b, :cnt[b]], want[j][0][b, :cnt[b]]) and np.array_equal(wts[b, :cnt[b]], want[j][1][b, :cnt[b]])
    with pytest.raises(BertError):
        eng.collect_sparse(hs[0])                                     # the ticket is gone
    h = eng.submit_sparse(batches[3], thr, 512)
    eng.abandon_sparse(h)
    bad = [np.array([1, 2, cfg.vocab_size + 3], np.int32)]
    for _ in range(4):                                                # failed submits must not strand slots
        with pytest.raises(BertError):
            eng.submit_sparse(bad, thr, 512)
    enc = SpladeEncoder(eng, threshold=thr, max_seq_len=128, sparse_cap=512)
    piped = enc.encode_batches_arrays(batches * 2)
    for i, res in enumerate(piped):
        ref = enc.encode_batch_arrays(batches[i % 4])
        assert len(res) == len(ref)
        for (a_ids, a_w), (b_ids, b_w) in zip(res, ref):
            assert np.array_equal(a_ids, b_ids) and np.array_equal(a_w, b_w)
    with pytest.raises(BertError):
        enc.encode_batches_arrays([batches[0], bad, batches[1]])      # the batch in flight is abandoned, then the error
    assert len(enc.encode_batches_arrays([batches[0]])[0]) == 3      # ... and the engine still has all its slots
    eng.close()
    # pooled embeddings through tickets
    cfg2 = R.BertConfig(vocab_size=997, hidden=384, layers=2, heads=6, intermediate=768, max_pos=128)
    e2, _ = _engine(cfg2, "none", seed=22)
    bs = [_seqs(cfg2, lens, seed=40 + i) for i, lens in enumerate(([33, 100], [5, 6, 7], [128]))]
    want = [e2.embed(b, None, "mean") for b in bs]
    hs = [e2.embed_submit(b, None, "mean") for b in bs]
    assert np.array_equal(e2.embed(bs[1], None, "mean"), want[1])    # blocking call with three tickets in flight
    assert e2.hidden(bs[2]).shape[1] == cfg2.hidden
    for j in (1, 2, 0):
        assert np.array_equal(e2.embed_collect(hs[j]), want[j])
    e2.embed_abandon(e2.embed_submit(bs[0], None, "cls"))
    assert np.array_equal(e2.embed(bs[2], None, "mean"), want[2])
    e2.close()
