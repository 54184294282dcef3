#!/usr/bin/env python3
"""Ticketed embedding throughput (3 tickets in flight) against the batch's token count, for the engine as configured by the
environment (CQS_HIP_EMBED_CONTEXTS=1 / 2): where does alternating two execution contexts pay?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
sys.argv = [sys.argv[0]]
import bench


def main():
    from cqs_amd.embedder import HipEmbedEngine, default_config
    cfg = default_config()
    eng = HipEmbedEngine(cfg)
    for name, t in bench.seeded_embed_weights(np, cfg).items():
        eng.set_tensor(name, t)
    eng.set_weights({})
    rng = np.random.default_rng(1)
    out = []
    for B in (16, 24, 28, 32, 36, 40, 48, 64, 96, 128):
        ids = rng.integers(1, cfg.vocab_size, size=(B, 512)).astype(np.int64)
        mask = np.ones((B, 512), np.int64)
        eng.run(ids, mask); eng.run(ids, mask)
        steps = max(6, 24 * 32 // B)
        t0 = time.perf_counter()
        for _ in range(max(3, steps // 2)):
            eng.run(ids, mask)
        sync = B * max(3, steps // 2) / (time.perf_counter() - t0)
        t0 = time.perf_counter()
        pend = []
        for _ in range(steps):
            pend.append(eng.submit(ids, mask))
            if len(pend) == 3:
                eng.collect(pend.pop(0), B)
        for t in pend:
            eng.collect(t, B)
        tick = B * steps / (time.perf_counter() - t0)
        out.append((B * 512, tick, sync))
    print(os.environ.get("CQS_HIP_EMBED_CONTEXTS", "2"), " ".join("%d:%.0f/%.0f" % o for o in out))


if __name__ == "__main__":
    main()
