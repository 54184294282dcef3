"""Seeded synthetic SPLADE corpora + queries shared by the CPU and GPU sparse-index tests."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "splade_index_kats.json")


def kats():
    return json.load(open(GOLDEN))


def kat_query(q):
    return [(int(t), float(w)) for t, w in q]      # float("nan") / float("inf") parse the JSON strings


def corpus(rng, n, vocab, nnz_lo, nnz_hi, dup_frac=0.0, zipf=1.1, special=False):
    """-> (doc_off u64, tokens u32, weights f32).  Token frequencies ~ Zipf (a few tokens in most documents, like SPLADE's
    expansion terms); `dup_frac` of the documents repeat one of their tokens; `special` sprinkles -0.0, subnormals, negative
    and huge weights (the f32 classes the reference's own proptest strategy names, index.rs:1717-1731)."""
    lens = rng.integers(nnz_lo, nnz_hi + 1, size=n)
    off = np.zeros(n + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    P = int(off[-1])
    p = 1.0 / np.arange(1, vocab + 1) ** zipf
    p /= p.sum()
    tok = np.zeros(P, dtype=np.uint32)
    for i in range(n):
        a, b = int(off[i]), int(off[i + 1])
        if b > a:
            k = min(b - a, vocab)
            t = rng.choice(vocab, size=k, replace=False, p=p)
            if k < b - a:
                t = np.concatenate([t, rng.integers(0, vocab, size=b - a - k)])
            tok[a:b] = np.sort(t)                     # the encoder emits ascending ids (src/splade/mod.rs:1049-1062)
            if dup_frac and b - a >= 2 and rng.random() < dup_frac:
                tok[a + 1] = tok[a]
    w = (rng.random(P, dtype=np.float32) * 2.5 + 0.01).astype(np.float32)
    if special and P:
        m = rng.random(P)
        w[m < 0.02] = np.float32(-0.0)
        w[(m >= 0.02) & (m < 0.04)] = np.float32(1e-42)          # subnormal
        w[(m >= 0.04) & (m < 0.08)] *= np.float32(-1.0)
        w[(m >= 0.08) & (m < 0.09)] = np.float32(3e37)
        w[(m >= 0.09) & (m < 0.10)] = np.float32(0.0)
    return off, tok, w


def query(rng, vocab, terms, zipf=1.1, shuffle=True, dups=0, absent=0):
    p = 1.0 / np.arange(1, vocab + 1) ** zipf
    p /= p.sum()
    t = rng.choice(vocab, size=min(terms, vocab), replace=False, p=p).astype(np.uint32)
    if dups:
        t = np.concatenate([t, t[:dups]])
    if absent:
        t = np.concatenate([t, np.arange(vocab + 7, vocab + 7 + absent, dtype=np.uint32)])
    if shuffle:
        rng.shuffle(t)
    else:
        t = np.sort(t)
    w = (rng.random(t.size, dtype=np.float32) * 2.0 + 0.05).astype(np.float32)
    return t, w
