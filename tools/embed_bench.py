#!/usr/bin/env python3
"""Embedding-forward throughput on one GPU: full EmbeddingGemma-300m geometry, seeded weights."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=24)
    ap.add_argument("--vocab", type=int, default=262144)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--len", type=int, default=512)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--lognormal", action="store_true")
    a = ap.parse_args()
    from cqs_amd.embedder import HipEmbedEngine, default_config
    cfg = default_config(); cfg.layers = a.layers; cfg.vocab_size = a.vocab
    eng = HipEmbedEngine(cfg)
    rng = np.random.default_rng(0)
    t0 = time.time()
    H, D, I = 768, 256, 1152
    def lin(n, k): return (rng.standard_normal((n, k), dtype=np.float32) / np.sqrt(k)).astype(np.float32)
    eng.set_tensor("embed_tokens.weight", rng.standard_normal((a.vocab, H), dtype=np.float32) * 0.05)
    for l in range(a.layers):
        p = f"layers.{l}."
        for n in ("input_layernorm", "post_attention_layernorm", "pre_feedforward_layernorm", "post_feedforward_layernorm"):
            eng.set_tensor(p + n + ".weight", rng.standard_normal(H, dtype=np.float32) * 0.1)
        eng.set_tensor(p + "self_attn.q_norm.weight", rng.standard_normal(D, dtype=np.float32) * 0.1)
        eng.set_tensor(p + "self_attn.k_norm.weight", rng.standard_normal(D, dtype=np.float32) * 0.1)
        eng.set_tensor(p + "self_attn.q_proj.weight", lin(3 * D, H)); eng.set_tensor(p + "self_attn.k_proj.weight", lin(D, H))
        eng.set_tensor(p + "self_attn.v_proj.weight", lin(D, H)); eng.set_tensor(p + "self_attn.o_proj.weight", lin(H, 3 * D))
        eng.set_tensor(p + "mlp.gate_proj.weight", lin(I, H)); eng.set_tensor(p + "mlp.up_proj.weight", lin(I, H))
        eng.set_tensor(p + "mlp.down_proj.weight", lin(H, I))
    eng.set_tensor("norm.weight", rng.standard_normal(H, dtype=np.float32) * 0.1)
    eng.set_tensor("dense1.weight", lin(3072, H)); eng.set_tensor("dense2.weight", lin(H, 3072))
    eng.set_weights({})
    print(f"weights ready in {time.time()-t0:.1f}s", file=sys.stderr)
    if a.lognormal:
        lens = np.clip(np.exp(rng.normal(np.log(300), 0.6, size=a.batch)).astype(int), 8, 2048)
    else:
        lens = np.full(a.batch, a.len)
    L = int(lens.max())
    ids = np.zeros((a.batch, L), np.int64); mask = np.zeros((a.batch, L), np.int64)
    for i, n in enumerate(lens):
        ids[i, :n] = rng.integers(1, a.vocab, size=n); mask[i, :n] = 1
    eng.run(ids, mask)
    ms = []
    t0 = time.time()
    for _ in range(a.iters):
        out = eng.run(ids, mask); ms.append(eng.last_ms())
    wall = (time.time() - t0) / a.iters
    toks = int(lens.sum())
    dev = float(np.median(ms)) / 1e3
    flops = toks * 2 * (a.layers * (768 * 1280 + 768 * 768 + 768 * 2304 + 1152 * 768))  # GEMMs only
    print(f"batch={a.batch} tokens={toks} device_ms={dev*1e3:.2f} wall_ms={wall*1e3:.2f} chunks/s={a.batch/dev:.0f} "
          f"tokens/s={toks/dev:.0f} gemm_TF/s={flops/dev/1e12:.1f} finite={bool(np.all(np.isfinite(out)))}")

if __name__ == "__main__":
    main()
