"""Pins the CPU restatement of the two BERT-family forwards (oracle/bert_ref.py: SPLADE masked-LM, cross-encoder
reranker) against the BERT definition shipped with `transformers` in this image (third-party library:
it fixes the operator semantics; the reference holds no golden logits for either model), and the Rust-side pooling /
threshold rules against the reference's own known-answer tests (src/splade/mod.rs:1716-1790).  CPU only."""
import math

import numpy as np
import pytest

from oracle import bert_ref as R

SMALL = R.BertConfig(vocab_size=211, hidden=64, layers=3, heads=4, intermediate=128, max_pos=48, num_labels=1)


def _batch(cfg, lens, seed, with_types=False):
    rng = np.random.default_rng(seed)
    L = max(lens)
    ids = np.zeros((len(lens), L), np.int64)
    mask = np.zeros((len(lens), L), np.int64)
    tt = np.zeros((len(lens), L), np.int64)
    for i, n in enumerate(lens):
        ids[i, :n] = rng.integers(1, cfg.vocab_size, size=n)
        mask[i, :n] = 1
        if with_types:
            tt[i, n // 2:n] = 1
    return ids, mask, tt


def test_mlm_matches_transformers():
    import torch
    from transformers import BertForMaskedLM
    w = R.seeded_weights(SMALL, "mlm", seed=3, bf16_exact=False)
    model = BertForMaskedLM(R.hf_config(SMALL)).eval()
    missing, unexpected = model.load_state_dict(R.hf_state_dict(SMALL, w, "mlm"), strict=False)
    assert not unexpected and all("position_ids" in m for m in missing), (missing, unexpected)
    ids, mask, _ = _batch(SMALL, [17, 5, 40], seed=4)
    with torch.no_grad():
        theirs = model(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask)).logits.numpy()
    ours = R.mlm_logits(SMALL, w, R.encode(SMALL, w, ids, mask))
    live = mask.astype(bool)
    assert np.max(np.abs(ours[live] - theirs[live])) < 2e-4


def test_classifier_matches_transformers():
    import torch
    from transformers import BertForSequenceClassification
    w = R.seeded_weights(SMALL, "classifier", seed=5, bf16_exact=False)
    model = BertForSequenceClassification(R.hf_config(SMALL)).eval()
    missing, unexpected = model.load_state_dict(R.hf_state_dict(SMALL, w, "classifier"), strict=False)
    assert not unexpected and all("position_ids" in m for m in missing), (missing, unexpected)
    ids, mask, tt = _batch(SMALL, [30, 12, 44, 3], seed=6, with_types=True)
    with torch.no_grad():
        theirs = model(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask),
                       token_type_ids=torch.from_numpy(tt)).logits.numpy()
    ours = R.classifier_logits(SMALL, w, R.encode(SMALL, w, ids, mask, tt))
    assert np.max(np.abs(ours - theirs)) < 2e-4
    s = R.rerank_scores(SMALL, w, ids, mask, tt)
    assert np.allclose(s, 1 / (1 + np.exp(-theirs[:, 0])), atol=1e-6) and np.all((s > 0) & (s < 1))


def test_activation_threshold_kats():
    """src/splade/mod.rs:1743-1790: NaN logit -> dropped; +Inf -> kept with +Inf weight; -Inf -> ln(1) = 0, dropped."""
    assert R.activate_threshold(float("nan"), 0.01) is None
    assert R.activate_threshold(float("inf"), 0.01) == float("inf")
    assert R.activate_threshold(float("-inf"), 0.01) is None
    assert R.activate_threshold(0.0, 0.01) is None
    assert abs(R.activate_threshold(1.0, 0.01) - math.log(2.0)) < 1e-7
    assert R.activate_threshold(-3.0, 0.0) is None                       # ln(1) = 0 is not > 0


def test_pooling_rules():
    """src/splade/mod.rs:1026-1062: padded positions cannot win, strict `>` from -inf, ascending ids, weights > threshold."""
    logits = np.array([[0.5, -1.0, np.nan, 2.0],
                       [0.1, -2.0, 1.0, np.nan],
                       [9.0, 9.0, 9.0, 9.0]], np.float32)      # row 2 is padding
    pooled = R.splade_pool(logits, real_len=2)
    assert pooled[0] == np.float32(0.5) and pooled[1] == np.float32(-1.0) and pooled[2] == np.float32(1.0) and pooled[3] == np.float32(2.0)
    sv = R.sparse_vector(pooled, threshold=0.01)
    assert [i for i, _ in sv] == [0, 2, 3] and all(wt > 0.01 for _, wt in sv)
    assert abs(sv[0][1] - math.log(1.5)) < 1e-6
    assert R.sparse_vector(R.splade_pool(logits, real_len=0), 0.01) == []      # empty sequence: all -inf -> nothing


def test_geometry_presets():
    b = R.splade_base()
    assert (b.hidden, b.layers, b.heads, b.head_dim, b.intermediate, b.vocab_size) == (768, 12, 12, 64, 3072, 30522)
    m = R.minilm_l6()
    assert (m.hidden, m.layers, m.heads, m.head_dim, m.intermediate) == (384, 6, 12, 32, 1536)   # src/reranker.rs:7,35
    n = sum(int(np.prod(s)) for _, s, _ in R.tensor_specs(m, "classifier"))
    assert 21e6 < n < 24e6                                                # "22M params"


def test_padding_invariance_of_the_sparse_vector():
    w = R.seeded_weights(SMALL, "mlm", seed=7)
    ids, mask, _ = _batch(SMALL, [9], seed=8)
    a, da = R.splade_encode_batch(SMALL, w, ids, mask, 0.05)
    ids2 = np.pad(ids, ((0, 0), (0, 7)))
    mask2 = np.pad(mask, ((0, 0), (0, 7)))
    b, db = R.splade_encode_batch(SMALL, w, ids2, mask2, 0.05)
    assert [i for i, _ in a[0]] == [i for i, _ in b[0]] and np.allclose(da, db, atol=1e-5)
