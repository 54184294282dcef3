"""Pins the CPU restatement of the EmbeddingGemma forward (oracle/gemma3_ref.py) against the
Gemma3 definition shipped with `transformers` in this image (third-party library that fixes the
operator semantics; the reference itself holds no golden vector for this path - SURVEY §8c).
CPU only."""
import numpy as np
import pytest

from oracle import gemma3_ref as G


def _hf_hidden(cfg, w, ids, mask):
    import torch
    from transformers import Gemma3TextModel
    torch.manual_seed(0)
    model = Gemma3TextModel(G.hf_config(cfg)).eval()
    missing, unexpected = model.load_state_dict(G.hf_state_dict(cfg, w), strict=False)
    assert not unexpected and all("inv_freq" in m or "embed_scale" in m for m in missing), (missing, unexpected)
    with torch.no_grad():
        out = model(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask))
    return out.last_hidden_state.numpy()


@pytest.mark.parametrize("L,window", [(12, 512), (40, 16), (70, 32)])
def test_matches_transformers_gemma3(L, window):
    cfg = G.GemmaConfig(vocab_size=300, hidden=64, layers=7, heads=4, kv_heads=2, head_dim=16, intermediate=96,
                        sliding_window=window, dense_hidden=128, max_seq=128, query_pre_attn_scalar=16.0)
    w = G.seeded_weights(cfg, seed=11, bf16_exact=False)
    rng = np.random.default_rng(L)
    ids = rng.integers(1, cfg.vocab_size, size=(3, L)).astype(np.int64)
    mask = np.ones((3, L), np.int64)
    mask[1, L // 2:] = 0      # right padding (pad_2d_i64_from_encodings, src/embedder/pooling.rs:40-57)
    ids[1, L // 2:] = 0
    mask[2, L - 3:] = 0
    ids[2, L - 3:] = 0
    ours = G.forward(cfg, w, ids, mask, return_hidden=True)
    theirs = _hf_hidden(cfg, w, ids, mask)
    live = mask.astype(bool)
    err = np.max(np.abs(ours[live] - theirs[live]))
    assert err < 2e-4, f"hidden states differ from transformers Gemma3TextModel: {err}"


def test_layer_types_and_window():
    cfg = G.GemmaConfig()
    assert [i for i in range(24) if cfg.is_full(i)] == [5, 11, 17, 23]
    assert cfg.window == 257            # 512 // 2 + 1 (configuration_gemma3.py:105-106)
    assert cfg.query_pre_attn_scalar ** -0.5 == 1 / 16
    n_params = sum(int(np.prod(s)) for _, s, _ in G.tensor_specs(cfg))
    assert 300e6 < n_params < 315e6     # "308 M params" (src/embedder/models.rs:430-448)


def test_pooling_head_and_padding_invariance():
    """Masked mean pool + 2 dense: padded tail must not change the embedding."""
    cfg = G.GemmaConfig(vocab_size=200, hidden=32, layers=2, heads=2, kv_heads=1, head_dim=16, intermediate=48,
                        sliding_window=8, dense_hidden=64, max_seq=64, query_pre_attn_scalar=16.0)
    w = G.seeded_weights(cfg, seed=3)
    ids = np.array([[5, 9, 17, 3, 44, 7]], np.int64)
    a = G.forward(cfg, w, ids, np.ones_like(ids))
    ids_p = np.concatenate([ids, np.zeros((1, 5), np.int64)], axis=1)
    mask_p = np.concatenate([np.ones_like(ids), np.zeros((1, 5), np.int64)], axis=1)
    b = G.forward(cfg, w, ids_p, mask_p)
    assert a.shape == (1, 32) and np.max(np.abs(a - b)) < 1e-5
    assert np.all(np.isfinite(a))


def test_seeded_weights_are_bf16_exact_and_deterministic():
    cfg = G.GemmaConfig(vocab_size=64, hidden=32, layers=1, heads=2, kv_heads=1, head_dim=16, intermediate=48,
                        dense_hidden=64)
    a = G.seeded_weights(cfg, seed=5)
    b = G.seeded_weights(cfg, seed=5)
    for k in a:
        assert np.array_equal(a[k], b[k])
        assert np.array_equal(G.round_bf16(a[k]), a[k])
