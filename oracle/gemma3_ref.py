"""CPU restatement of the EmbeddingGemma-300m forward — TEST INFRASTRUCTURE ONLY.

What the reference runs for this path is a third-party artefact absent from /root/reference:
the ONNX export `onnx-community/embeddinggemma-300m-ONNX` executed by ONNX Runtime
(src/embedder/core.rs:1097 `session.run`; model preset src/embedder/models.rs:455-470).  The
reference holds NO golden embedding vector — its real-model tests assert only
dim, unit norm, determinism and finiteness (tests/embedding_test.rs:39-239).  Numerics of
this forward are therefore "parity unpinned" against the reference; what IS pinned:

  * the operator semantics, against the Gemma3 definition shipped in this image
    (transformers/models/gemma3/modeling_gemma3.py — third-party library, not the reference):
    tests/test_gemma3_oracle.py compares this file with `Gemma3TextModel` on seeded weights;
  * the I/O contract of `Embedder::embed_batch` (src/embedder/core.rs:994-1273): int64
    `input_ids` / `attention_mask` [B, L] right-padded with pad_id 0, output
    `sentence_embedding` [B, 768] f32, L2-normalised by the caller (core.rs:1196-1203).

Architecture restated (Gemma3 text encoder, bidirectional; sentence-transformers head):
  embed_tokens * sqrt(hidden)                                  modeling_gemma3.py:106-117
  per layer: x += post_attn_norm(attn(input_norm(x)))          :386-430
             x += post_ffw_norm(mlp(pre_ffw_norm(x)))
  RMSNorm: x * rsqrt(mean(x^2) + eps) * (1 + w), in fp32       :136-150
  attention: q/k/v proj, per-head q_norm/k_norm, RoPE (theta per layer type), GQA,
             scores * query_pre_attn_scalar^-0.5, mask, softmax, o_proj        :308-383
  masks: full layers attend every non-padded key; sliding layers additionally need
         |q - k| < sliding_window//2 + 1 (bidirectional)       :471-483, configuration_gemma3.py:105-106
  layer i is full attention iff (i + 1) % 6 == 0               configuration_gemma3.py:109-113
  MLP: down(gelu_tanh(gate(x)) * up(x))                        :120-133
  final norm; masked mean pool; Dense 768->3072 (no bias); Dense 3072->768 (no bias)
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List

import numpy as np


@dataclass
class GemmaConfig:
    vocab_size: int = 262144
    hidden: int = 768
    layers: int = 24
    heads: int = 3
    kv_heads: int = 1
    head_dim: int = 256
    intermediate: int = 1152
    sliding_window: int = 512          # config value; the bidirectional mask uses window//2 + 1
    sliding_pattern: int = 6           # every 6th layer is full attention
    rms_eps: float = 1e-6
    rope_theta_global: float = 1_000_000.0
    rope_theta_local: float = 10_000.0
    query_pre_attn_scalar: float = 256.0
    dense_hidden: int = 3072
    max_seq: int = 2048

    def is_full(self, layer: int) -> bool:
        return (layer + 1) % self.sliding_pattern == 0

    @property
    def window(self) -> int:
        return self.sliding_window // 2 + 1


def tensor_specs(cfg: GemmaConfig) -> List[tuple]:
    """(name, shape, kind) of every weight, in a fixed order (HF Gemma3TextModel names + the head)."""
    H, D, I = cfg.hidden, cfg.head_dim, cfg.intermediate
    specs = [("embed_tokens.weight", (cfg.vocab_size, H), "embed")]
    for i in range(cfg.layers):
        p = f"layers.{i}."
        specs += [
            (p + "input_layernorm.weight", (H,), "norm"),
            (p + "self_attn.q_proj.weight", (cfg.heads * D, H), "linear"),
            (p + "self_attn.k_proj.weight", (cfg.kv_heads * D, H), "linear"),
            (p + "self_attn.v_proj.weight", (cfg.kv_heads * D, H), "linear"),
            (p + "self_attn.o_proj.weight", (H, cfg.heads * D), "linear"),
            (p + "self_attn.q_norm.weight", (D,), "norm"),
            (p + "self_attn.k_norm.weight", (D,), "norm"),
            (p + "post_attention_layernorm.weight", (H,), "norm"),
            (p + "pre_feedforward_layernorm.weight", (H,), "norm"),
            (p + "mlp.gate_proj.weight", (I, H), "linear"),
            (p + "mlp.up_proj.weight", (I, H), "linear"),
            (p + "mlp.down_proj.weight", (H, I), "linear"),
            (p + "post_feedforward_layernorm.weight", (H,), "norm"),
        ]
    specs += [("norm.weight", (H,), "norm"),
              ("dense1.weight", (cfg.dense_hidden, H), "linear"),
              ("dense2.weight", (H, cfg.dense_hidden), "linear")]
    return specs


def seeded_weights(cfg: GemmaConfig, seed: int = 1234, bf16_exact: bool = True) -> Dict[str, np.ndarray]:
    """Deterministic synthetic weights (numpy Philox, one stream per tensor index).

    linear ~ N(0, 1/fan_in), embed ~ N(0, 1) * 0.05, norm ~ N(0, 0.1) (Gemma norms use 1 + w).
    With bf16_exact the values are rounded to bf16-representable f32 so that a bf16 engine and
    the fp32 oracle start from identical weights.
    """
    out = {}
    for t, (name, shape, kind) in enumerate(tensor_specs(cfg)):
        rng = np.random.Generator(np.random.Philox(key=seed + 7919 * t))
        w = rng.standard_normal(shape, dtype=np.float32)
        if kind == "linear":
            w *= np.float32(1.0 / math.sqrt(shape[1]))
        elif kind == "embed":
            w *= np.float32(0.05)
        else:
            w *= np.float32(0.1)
        if bf16_exact:
            w = round_bf16(w)
        out[name] = w
    return out


def round_bf16(x: np.ndarray) -> np.ndarray:
    """f32 -> nearest-even bf16 -> f32 (finite inputs)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)
    return r.view(np.float32)


def forward(cfg: GemmaConfig, w: Dict[str, np.ndarray], input_ids: np.ndarray, attention_mask: np.ndarray,
            return_hidden: bool = False):
    """fp32 forward on CPU (torch).  input_ids / attention_mask: int64 [B, L].  -> f32 [B, hidden]
    (`sentence_embedding` before L2 normalisation)."""
    import torch
    torch.set_grad_enabled(False)
    ids = torch.from_numpy(np.ascontiguousarray(input_ids, dtype=np.int64))
    mask = torch.from_numpy(np.ascontiguousarray(attention_mask, dtype=np.int64))
    B, L = ids.shape
    W = {k: torch.from_numpy(v.astype(np.float32)) for k, v in w.items()}
    H, D, nh, nkv = cfg.hidden, cfg.head_dim, cfg.heads, cfg.kv_heads

    def rms(x, wt):
        xf = x.float()
        return xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + cfg.rms_eps) * (1.0 + wt)

    x = W["embed_tokens.weight"][ids] * torch.tensor(math.sqrt(H), dtype=torch.float32)
    pos = torch.arange(L, dtype=torch.float32)
    key_ok = mask.bool()[:, None, None, :]                                   # [B,1,1,L] non-padded keys
    dist = (torch.arange(L)[:, None] - torch.arange(L)[None, :]).abs()       # |q - k|
    neg = torch.finfo(torch.float32).min

    def rope_tables(theta):
        inv = 1.0 / (theta ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
        fr = pos[:, None] * inv[None, :]
        emb = torch.cat([fr, fr], dim=-1)
        return emb.cos(), emb.sin()

    tables = {True: rope_tables(cfg.rope_theta_global), False: rope_tables(cfg.rope_theta_local)}

    def rot(t):
        a, b = t[..., : D // 2], t[..., D // 2:]
        return torch.cat([-b, a], dim=-1)

    for i in range(cfg.layers):
        p = f"layers.{i}."
        full = cfg.is_full(i)
        h = rms(x, W[p + "input_layernorm.weight"])
        q = (h @ W[p + "self_attn.q_proj.weight"].T).view(B, L, nh, D).transpose(1, 2)
        k = (h @ W[p + "self_attn.k_proj.weight"].T).view(B, L, nkv, D).transpose(1, 2)
        v = (h @ W[p + "self_attn.v_proj.weight"].T).view(B, L, nkv, D).transpose(1, 2)
        q = rms(q, W[p + "self_attn.q_norm.weight"])
        k = rms(k, W[p + "self_attn.k_norm.weight"])
        cos, sin = tables[full]
        q = q * cos + rot(q) * sin
        k = k * cos + rot(k) * sin
        k = k.repeat_interleave(nh // nkv, dim=1)
        v = v.repeat_interleave(nh // nkv, dim=1)
        s = (q @ k.transpose(2, 3)) * (cfg.query_pre_attn_scalar ** -0.5)
        allow = key_ok if full else (key_ok & (dist < cfg.window)[None, None])
        s = s.masked_fill(~allow, neg)
        a = torch.softmax(s, dim=-1) @ v
        a = a.transpose(1, 2).reshape(B, L, nh * D) @ W[p + "self_attn.o_proj.weight"].T
        x = x + rms(a, W[p + "post_attention_layernorm.weight"])
        h = rms(x, W[p + "pre_feedforward_layernorm.weight"])
        g = torch.nn.functional.gelu(h @ W[p + "mlp.gate_proj.weight"].T, approximate="tanh")
        m = (g * (h @ W[p + "mlp.up_proj.weight"].T)) @ W[p + "mlp.down_proj.weight"].T
        x = x + rms(m, W[p + "post_feedforward_layernorm.weight"])
    x = rms(x, W["norm.weight"])
    if return_hidden:
        return x.numpy()
    mf = mask.float()[:, :, None]
    cnt = mf.sum(1).clamp(min=1e-9)                  # sentence-transformers Pooling (mean, masked)
    pooled = (x * mf).sum(1) / cnt
    y = pooled @ W["dense1.weight"].T
    y = y @ W["dense2.weight"].T
    return y.numpy()


def hf_state_dict(cfg: GemmaConfig, w: Dict[str, np.ndarray]):
    """The same weights under HF `Gemma3TextModel` parameter names (for the cross-check test)."""
    import torch
    return {k: torch.from_numpy(v) for k, v in w.items() if not k.startswith("dense")}


def hf_config(cfg: GemmaConfig):
    from transformers import Gemma3TextConfig
    return Gemma3TextConfig(
        vocab_size=cfg.vocab_size, hidden_size=cfg.hidden, intermediate_size=cfg.intermediate,
        num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads, num_key_value_heads=cfg.kv_heads,
        head_dim=cfg.head_dim, sliding_window=cfg.sliding_window, rms_norm_eps=cfg.rms_eps,
        query_pre_attn_scalar=int(cfg.query_pre_attn_scalar), max_position_embeddings=cfg.max_seq,
        use_bidirectional_attention=True, attn_implementation="eager", pad_token_id=0,
        rope_parameters={"full_attention": {"rope_type": "default", "rope_theta": cfg.rope_theta_global},
                         "sliding_attention": {"rope_type": "default", "rope_theta": cfg.rope_theta_local}},
    )
