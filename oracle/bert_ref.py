"""CPU restatement of the BERT-family forwards behind cqs's two auxiliary ONNX models — TEST INFRASTRUCTURE ONLY.

SURVEY.md §8(f)4: the SPLADE sparse encoder (src/splade/mod.rs) and the cross-encoder reranker (src/reranker.rs)
reuse `create_session`; both are BERT encoders with a small head:

  * SPLADE (`naver/splade-cocondenser-ensembledistil`: BERT-base masked-LM, src/splade/mod.rs:120-150): the ONNX graph
    returns raw MLM `logits` [B, L, vocab]; the Rust side pools them (src/splade/mod.rs:980-1070):
        pooled[v]  = max over the sequence's REAL tokens of logits[s, v]      (strict `>` from -inf: NaN never wins)
        activated  = ln(1 + max(pooled, 0))
        keep (v, activated) iff activated > threshold, ascending v            (default threshold: mod.rs:400-410)
  * reranker (`cross-encoder/ms-marco-MiniLM-L-6-v2`: BERT, 6 layers, hidden 384, src/reranker.rs:7,35): inputs
    input_ids / attention_mask / token_type_ids, output logits [B, n]; score_i = sigmoid(logits[i, 0])
    (src/reranker.rs:474-520).

Both model artefacts are third-party and absent from /root/reference, and the reference holds no golden logits for
either: numerics of these forwards are "parity unpinned" against the reference.  What IS pinned: the operator
semantics against `transformers` `BertForMaskedLM` / `BertForSequenceClassification` on seeded weights
(tests/test_bert_oracle.py), and the pooling / threshold rules against the reference's own known-answer tests
(src/splade/mod.rs:1716-1790: NaN dropped, +Inf kept, -Inf dropped).

Architecture restated (transformers/models/bert/modeling_bert.py):
  embeddings = LayerNorm(word[id] + position[pos] + token_type[tt])                      BertEmbeddings
  per layer:  a = SelfAttention(x)  (q, k, v = x W^T + b; softmax(q k^T / sqrt(d) + mask) v; 12 heads)
              x = LayerNorm(x + a Wo^T + bo)                                              BertSelfOutput
              x = LayerNorm(x + gelu_erf(x W1^T + b1) W2^T + b2)                          BertIntermediate / BertOutput
  MLM head:   logits = LayerNorm(gelu_erf(x Wt^T + bt)) E^T + b   (decoder tied to word embeddings)   BertLMPredictionHead
  classifier: logits = tanh(x[:, 0] Wp^T + bp) Wc^T + bc                                  BertPooler + classifier
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np


@dataclass
class BertConfig:
    vocab_size: int = 30522
    hidden: int = 768
    layers: int = 12
    heads: int = 12
    intermediate: int = 3072
    max_pos: int = 512
    type_vocab: int = 2
    ln_eps: float = 1e-12
    num_labels: int = 1

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads


def splade_base() -> BertConfig:
    """naver/splade-cocondenser-ensembledistil = bert-base-uncased geometry."""
    return BertConfig()


def minilm_l6() -> BertConfig:
    """cross-encoder/ms-marco-MiniLM-L-6-v2."""
    return BertConfig(hidden=384, layers=6, heads=12, intermediate=1536, num_labels=1)


def e5_base() -> BertConfig:
    """intfloat/e5-base-v2 and the v9-200k fine-tune: BERT-base (src/embedder/models.rs:346-372)."""
    return BertConfig()


def bge_large() -> BertConfig:
    """BAAI/bge-large-en-v1.5 and bge-large-ft: BERT-large (src/embedder/models.rs:374-405)."""
    return BertConfig(hidden=1024, layers=24, heads=16, intermediate=4096)


def tensor_specs(cfg: BertConfig, head: str) -> List[Tuple[str, tuple, str]]:
    """(name, shape, kind): HF names without the leading `bert.`; head = "mlm" | "classifier" | "none"."""
    H, I = cfg.hidden, cfg.intermediate
    s = [("embeddings.word_embeddings.weight", (cfg.vocab_size, H), "embed"),
         ("embeddings.position_embeddings.weight", (cfg.max_pos, H), "embed"),
         ("embeddings.token_type_embeddings.weight", (cfg.type_vocab, H), "embed"),
         ("embeddings.LayerNorm.weight", (H,), "gamma"), ("embeddings.LayerNorm.bias", (H,), "beta")]
    for i in range(cfg.layers):
        p = f"encoder.layer.{i}."
        for n in ("query", "key", "value"):
            s += [(p + f"attention.self.{n}.weight", (H, H), "linear"), (p + f"attention.self.{n}.bias", (H,), "beta")]
        s += [(p + "attention.output.dense.weight", (H, H), "linear"), (p + "attention.output.dense.bias", (H,), "beta"),
              (p + "attention.output.LayerNorm.weight", (H,), "gamma"), (p + "attention.output.LayerNorm.bias", (H,), "beta"),
              (p + "intermediate.dense.weight", (I, H), "linear"), (p + "intermediate.dense.bias", (I,), "beta"),
              (p + "output.dense.weight", (H, I), "linear"), (p + "output.dense.bias", (H,), "beta"),
              (p + "output.LayerNorm.weight", (H,), "gamma"), (p + "output.LayerNorm.bias", (H,), "beta")]
    if head == "mlm":
        s += [("cls.predictions.transform.dense.weight", (H, H), "linear"), ("cls.predictions.transform.dense.bias", (H,), "beta"),
              ("cls.predictions.transform.LayerNorm.weight", (H,), "gamma"), ("cls.predictions.transform.LayerNorm.bias", (H,), "beta"),
              ("cls.predictions.bias", (cfg.vocab_size,), "beta")]
    elif head == "classifier":
        s += [("pooler.dense.weight", (H, H), "linear"), ("pooler.dense.bias", (H,), "beta"),
              ("classifier.weight", (cfg.num_labels, H), "linear"), ("classifier.bias", (cfg.num_labels,), "beta")]
    elif head != "none":
        raise ValueError(head)
    return s


def bf16_round(a: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even to bf16, returned as f32 (so the GPU's bf16 weights are exactly these values)."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) & 0xFFFF0000).view(np.float32)


def seeded_weights(cfg: BertConfig, head: str, seed: int = 0, bf16_exact: bool = True) -> Dict[str, np.ndarray]:
    rng = np.random.default_rng(seed)
    w = {}
    for name, shape, kind in tensor_specs(cfg, head):
        if kind == "embed":
            t = rng.standard_normal(shape, dtype=np.float32) * 0.05
        elif kind == "linear":
            t = rng.standard_normal(shape, dtype=np.float32) / math.sqrt(shape[1])
        elif kind == "gamma":
            t = 1.0 + rng.standard_normal(shape, dtype=np.float32) * 0.1
        else:
            t = rng.standard_normal(shape, dtype=np.float32) * 0.05
        if bf16_exact and kind in ("embed", "linear"):
            t = bf16_round(t)
        w[name] = t.astype(np.float32)
    return w


def _t(x):
    import torch
    return torch.from_numpy(np.ascontiguousarray(x))


def encode(cfg: BertConfig, w: Dict[str, np.ndarray], ids: np.ndarray, mask: np.ndarray,
           type_ids: np.ndarray | None = None):
    """Final hidden states [B, L, H] (torch f32 tensor); ids / mask / type_ids int64 [B, L], right-padded."""
    import torch
    import torch.nn.functional as F
    B, L = ids.shape
    H, nh, d = cfg.hidden, cfg.heads, cfg.head_dim
    if type_ids is None:
        type_ids = np.zeros_like(ids)
    W = {k: _t(v) for k, v in w.items()}
    pos = torch.arange(L)
    x = W["embeddings.word_embeddings.weight"][_t(ids)] + W["embeddings.position_embeddings.weight"][pos][None] + \
        W["embeddings.token_type_embeddings.weight"][_t(type_ids)]
    x = F.layer_norm(x, (H,), W["embeddings.LayerNorm.weight"], W["embeddings.LayerNorm.bias"], cfg.ln_eps)
    neg = torch.zeros((B, 1, 1, L))
    neg[_t(mask)[:, None, None, :] == 0] = float("-inf")
    for i in range(cfg.layers):
        p = f"encoder.layer.{i}."
        def lin(name, t):
            return t @ W[p + name + ".weight"].T + W[p + name + ".bias"]
        q = lin("attention.self.query", x).view(B, L, nh, d).transpose(1, 2)
        k = lin("attention.self.key", x).view(B, L, nh, d).transpose(1, 2)
        v = lin("attention.self.value", x).view(B, L, nh, d).transpose(1, 2)
        s = q @ k.transpose(-1, -2) / math.sqrt(d) + neg
        a = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, L, H)
        x = F.layer_norm(x + lin("attention.output.dense", a), (H,), W[p + "attention.output.LayerNorm.weight"],
                         W[p + "attention.output.LayerNorm.bias"], cfg.ln_eps)
        h = F.gelu(lin("intermediate.dense", x))                       # erf GELU (hidden_act = "gelu")
        x = F.layer_norm(x + lin("output.dense", h), (H,), W[p + "output.LayerNorm.weight"],
                         W[p + "output.LayerNorm.bias"], cfg.ln_eps)
    return x


def mlm_logits(cfg: BertConfig, w: Dict[str, np.ndarray], hidden) -> np.ndarray:
    """[B, L, vocab] f32: BertLMPredictionHead with the decoder tied to the word embeddings."""
    import torch.nn.functional as F
    t = hidden @ _t(w["cls.predictions.transform.dense.weight"]).T + _t(w["cls.predictions.transform.dense.bias"])
    t = F.layer_norm(F.gelu(t), (cfg.hidden,), _t(w["cls.predictions.transform.LayerNorm.weight"]),
                     _t(w["cls.predictions.transform.LayerNorm.bias"]), cfg.ln_eps)
    return (t @ _t(w["embeddings.word_embeddings.weight"]).T + _t(w["cls.predictions.bias"])).numpy()


def classifier_logits(cfg: BertConfig, w: Dict[str, np.ndarray], hidden) -> np.ndarray:
    """[B, num_labels] f32: BertPooler (first token, dense + tanh) + classifier."""
    import torch
    pooled = torch.tanh(hidden[:, 0] @ _t(w["pooler.dense.weight"]).T + _t(w["pooler.dense.bias"]))
    return (pooled @ _t(w["classifier.weight"]).T + _t(w["classifier.bias"])).numpy()


# ---- the Rust side of the SPLADE path (src/splade/mod.rs:1015-1062) ----------------------------------------------
def splade_pool(logits: np.ndarray, real_len: int) -> np.ndarray:
    """pooled[v] = max over s < real_len of logits[s, v], folded with a strict `>` from -inf (mod.rs:1033-1043):
    a NaN logit never becomes the maximum; an all-NaN / empty column stays -inf."""
    L, V = logits.shape
    pooled = np.full(V, -np.inf, dtype=np.float32)
    for s in range(min(real_len, L)):
        row = logits[s]
        take = row > pooled                       # False for NaN
        pooled = np.where(take, row, pooled)
    return pooled


def activate(val: np.ndarray) -> np.ndarray:
    """ln(1 + max(val, 0)) with Rust's f32::max (a NaN operand yields the other one: NaN -> 0)."""
    v = np.asarray(val, dtype=np.float32)
    clamped = np.where(np.isnan(v), np.float32(0.0), np.maximum(v, np.float32(0.0)))
    with np.errstate(over="ignore"):
        return np.log(np.float32(1.0) + clamped).astype(np.float32)


def activate_threshold(val: float, threshold: float):
    """The reference's test helper (src/splade/mod.rs:1733-1741): Some(activated) iff activated > threshold."""
    a = float(activate(np.array([val], np.float32))[0])
    return a if a > threshold else None


def sparse_vector(pooled: np.ndarray, threshold: float) -> List[Tuple[int, float]]:
    act = activate(pooled)
    keep = np.nonzero(act > np.float32(threshold))[0]          # ascending id, NaN > t is False
    return [(int(i), float(act[i])) for i in keep]


def splade_encode_batch(cfg: BertConfig, w, ids: np.ndarray, mask: np.ndarray, threshold: float):
    """`SpladeEncoder::encode_batch` on token ids: (sparse vectors, dense activations [B, vocab])."""
    hidden = encode(cfg, w, ids, mask)
    logits = mlm_logits(cfg, w, hidden)
    lens = mask.sum(axis=1)
    dense = np.stack([activate(splade_pool(logits[b], int(lens[b]))) for b in range(len(ids))])
    return [sparse_vector(splade_pool(logits[b], int(lens[b])), threshold) for b in range(len(ids))], dense


def pooled_embedding(cfg: BertConfig, w, ids: np.ndarray, mask: np.ndarray, type_ids: np.ndarray | None = None,
                     pooling: str = "mean") -> np.ndarray:
    """The BERT-family embedder presets: last_hidden_state -> `mean_pool` (src/embedder/pooling.rs:87-121: masked sum /
    count, zero rows for an empty mask) or `cls_pool` (:123-128).  Not normalised."""
    h = encode(cfg, w, ids, mask, type_ids).numpy()
    if pooling == "cls":
        return h[:, 0].astype(np.float32)
    m = mask.astype(np.float32)[:, :, None]
    cnt = m.sum(axis=1)
    summed = (h * m).sum(axis=1)
    return np.where(cnt > 0, summed / np.maximum(cnt, 1.0), 0.0).astype(np.float32)


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-np.asarray(x, dtype=np.float64)))


def rerank_scores(cfg: BertConfig, w, ids: np.ndarray, mask: np.ndarray, type_ids: np.ndarray) -> np.ndarray:
    """`compute_scores_opt` on token ids: sigmoid(logits[:, 0]) (src/reranker.rs:516-518)."""
    hidden = encode(cfg, w, ids, mask, type_ids)
    return sigmoid(classifier_logits(cfg, w, hidden)[:, 0]).astype(np.float32)


# ---- transformers bridge (tests only) ---------------------------------------------------------------------------
def hf_config(cfg: BertConfig):
    from transformers import BertConfig as HC
    return HC(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden, num_hidden_layers=cfg.layers,
              num_attention_heads=cfg.heads, intermediate_size=cfg.intermediate, max_position_embeddings=cfg.max_pos,
              type_vocab_size=cfg.type_vocab, layer_norm_eps=cfg.ln_eps, hidden_act="gelu", num_labels=cfg.num_labels,
              hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)


def hf_state_dict(cfg: BertConfig, w: Dict[str, np.ndarray], head: str):
    import torch
    sd = {}
    for k, v in w.items():
        name = k if k.startswith(("cls.", "classifier.")) else "bert." + k
        sd[name] = torch.from_numpy(v.copy())
    if head == "mlm":
        sd["cls.predictions.decoder.weight"] = sd["bert.embeddings.word_embeddings.weight"]
        sd["cls.predictions.decoder.bias"] = sd["cls.predictions.bias"]
    return sd
