#!/usr/bin/env python3
"""Experiment: one engine with 32 x 512 batches vs TWO engines (own stream, own scratch, own weight copy) with
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

def run(engines, B, L, iters):
    rng = np.random.default_rng(1)
    ids = rng.integers(1, 262144, size=(B, L)).astype(np.int64); mask = np.ones((B, L), np.int64)
    for e in engines: e.run(ids, mask)
    def work(e):
        pend = []
        for _ in range(iters):
            pend.append(e.submit(ids, mask))
            if len(pend) == 3: e.collect(pend.pop(0), B)
        for t in pend: e.collect(t, B)
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(e,)) for e in engines]
    [t.start() for t in th]; [t.join() for t in th]
    dt = time.perf_counter() - t0
    return len(engines) * iters * B / dt

e1, cfg = make_engine(0)
print("one engine  32x512: %.0f chunks/s" % run([e1], 32, 512, 20), flush=True)
print("one engine  64x512: %.0f chunks/s" % run([e1], 64, 512, 10), flush=True)
e2, _ = make_engine(1)
print("two engines 16x512: %.0f chunks/s" % run([e1, e2], 16, 512, 40), flush=True)
print("two engines 32x512: %.0f chunks/s" % run([e1, e2], 32, 512, 20), flush=True)
print("one engine  32x512: %.0f chunks/s" % run([e1], 32, 512, 20), flush=True)
