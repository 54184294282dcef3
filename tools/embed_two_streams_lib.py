#!/usr/bin/env python3
"""Shared by the embed throughput experiments: seeded full-geometry engine.
Experiment notes: one engine with 32 x 512 batches vs TWO engines (own stream, own scratch, own weight copy) with
16 x 512 (and 32 x 512) batches each, driven from two host threads: does desynchronised execution of two kernel chains
fill the bubbles of the lock-step single chain?"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cqs_amd.embedder import HipEmbedEngine, default_config

def make_engine(seed=0):
    cfg = default_config()
    eng = HipEmbedEngine(cfg)
    rng = np.random.default_rng(seed)
    H, D, I, V = 768, 256, 1152, cfg.vocab_size
    lin = lambda n, k: (rng.standard_normal((n, k), dtype=np.float32) / np.sqrt(k)).astype(np.float32)
    eng.set_tensor("embed_tokens.weight", rng.standard_normal((V, H), dtype=np.float32) * 0.05)
    for l in range(cfg.layers):
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
    return eng, cfg

