"""SPLADE encoder, BERT-family embedder presets and the reranker at their real geometry."""
import time


def aux_models_leg(a, np):
    """the two BERT-family auxiliary models at their real geometry (seeded weights, synthetic token
    ids): SPLADE encode (BERT-base masked-LM + pooling -> sparse vectors) and reranker scoring (MiniLM-L6
    cross-encoder).  Each is checked against the fp32 CPU oracle on a small batch outside the timed region; the CPU
    baseline is that oracle (torch CPU) on a bounded sample."""
    from oracle import bert_ref as R
    from cqs_amd import _lib
    from cqs_amd.splade import HipBertEngine, Reranker, SpladeEncoder, bert_config
    rng = np.random.default_rng(0xC950009)
    out = {}

    def padded(seqs, types=None):
        L = max(len(s) for s in seqs)
        ids = np.zeros((len(seqs), L), np.int64); mask = np.zeros((len(seqs), L), np.int64); tt = np.zeros((len(seqs), L), np.int64)
        for i, s in enumerate(seqs):
            ids[i, :len(s)] = s; mask[i, :len(s)] = 1
            if types is not None:
                tt[i, :len(s)] = types[i]
        return ids, mask, tt

    # SPLADE
    cfg = R.splade_base()
    w = R.seeded_weights(cfg, "mlm", seed=1)
    eng = HipBertEngine(bert_config(_lib.BERT_HEAD_MLM))
    eng.set_weights(w)
    small = [rng.integers(1, cfg.vocab_size, size=n).astype(np.int32) for n in (48, 200, 7)]
    got = eng.splade_dense(small)
    t0 = time.perf_counter()
    _, want = R.splade_encode_batch(cfg, w, *padded(small)[:2], 0.01)
    cpu_s = time.perf_counter() - t0
    err = float(np.max(np.abs(got - want)))
    assert err < 0.08, "splade activations differ from the fp32 oracle: %g" % err
    B, L = 64, 256
    seqs = [rng.integers(1, cfg.vocab_size, size=L).astype(np.int32) for _ in range(B)]
    # seeded weights make half the vocabulary "active"; a trained SPLADE keeps 100-300 entries per document
    # (src/splade/mod.rs:44): put the threshold where ~200 survive so the host-side filter does realistic work
    warm = eng.splade_dense(seqs)
    thr = float(np.sort(warm[0])[-200])
    enc = SpladeEncoder(eng, threshold=thr)
    t_end = time.perf_counter() + 0.5                                  # (let the oracle's CPU threads stop spinning)
    while time.perf_counter() < t_end:
        enc.encode_batch_arrays(seqs)
    steps = max(4, a.embed_steps)
    t0 = time.perf_counter()
    for _ in range(steps):
        sv = enc.encode_batch_arrays(seqs)
    dt_sync = (time.perf_counter() - t0) / steps                       # one blocking encode_batch at a time
    enc.encode_batches_arrays([seqs] * 3)
    t0 = time.perf_counter()
    piped = enc.encode_batches_arrays([seqs] * (2 * steps))             # the index pipeline's form: 3 tickets in flight
    dt = (time.perf_counter() - t0) / (2 * steps)
    assert all(np.array_equal(piped[0][b][0], sv[b][0]) and np.array_equal(piped[-1][b][1], sv[b][1]) for b in range(B)), "tickets != blocking call"
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.splade_dense(seqs)
    dt_dense = (time.perf_counter() - t0) / steps
    flops = 2.0 * B * L * (cfg.layers * (4 * cfg.hidden * cfg.hidden + 2 * cfg.hidden * cfg.intermediate) + cfg.hidden * cfg.hidden
                           + cfg.hidden * cfg.vocab_size) + 4.0 * B * cfg.layers * L * L * cfg.hidden
    out["splade"] = {"model": "BERT-base masked-LM geometry (12 x [768 | 12 x 64 | 3072], vocab 30522), seeded weights",
                     "batch": B, "tokens_per_doc": L, "docs_per_sec": round(B / dt, 1), "tokens_per_sec": round(B * L / dt, 1),
                     "ms_per_batch": round(dt * 1e3, 3), "sync_api": {"docs_per_sec": round(B / dt_sync, 1), "ms_per_batch": round(dt_sync * 1e3, 3)},
                     "ms_per_batch_device_side": round(dt_dense * 1e3, 3),
                     "tflops": round(flops / dt_dense / 1e12, 1), "nnz_per_doc": round(float(np.mean([len(v[0]) for v in sv])), 1),
                     "checked": {"max_abs_err_vs_fp32_oracle": round(err, 4)},
                     "cpu_baseline": {"docs_per_sec": round(3 / cpu_s, 2), "kind": "port", "sample": "oracle/bert_ref (torch CPU fp32), 3 docs / 255 tokens"},
                     "threshold": round(thr, 4),
                     "note": "host API: token ids in, sparse vectors out (threshold filter of src/splade/mod.rs:1049-1062 on the device), 3 tickets in flight "
                             "(cqs_hip_splade_submit_sparse / _collect_sparse; sync_api = one blocking cqs_hip_splade_encode_sparse per batch); "
                             "threshold set where ~200 entries per document survive (seeded weights are not sparse)"}
    eng.close()

    # BERT-family embedder presets of the `Embedder` seam (bge-large = the reference's strongest: src/embedder/models.rs:374-405)
    for name, cfg, B, L in (("bge_large", R.bge_large(), 32, 512), ("e5_base", R.e5_base(), 32, 512)):
        w = R.seeded_weights(cfg, "none", seed=3)
        eng = HipBertEngine(bert_config(_lib.BERT_HEAD_NONE, hidden=cfg.hidden, layers=cfg.layers, heads=cfg.heads,
                                        intermediate=cfg.intermediate))
        eng.set_weights(w)
        small = [rng.integers(1, cfg.vocab_size, size=n).astype(np.int32) for n in (40, 130)]
        got = eng.embed(small, None, "mean")
        t0 = time.perf_counter()
        want = R.pooled_embedding(cfg, w, *padded(small)[:2])
        cpu_s = time.perf_counter() - t0
        cs = min(float(np.dot(got[i], want[i]) / (np.linalg.norm(got[i]) * np.linalg.norm(want[i]))) for i in range(2))
        assert cs > 0.999, "%s embeddings differ from the fp32 oracle: cos %g" % (name, cs)
        seqs = [rng.integers(1, cfg.vocab_size, size=L).astype(np.int32) for _ in range(B)]
        t_end = time.perf_counter() + 0.3
        while time.perf_counter() < t_end:
            eng.embed(seqs, None, "mean")
        steps = max(4, a.embed_steps)
        t0 = time.perf_counter()
        for _ in range(steps):
            ref_out = eng.embed(seqs, None, "mean")
        dt_sync = (time.perf_counter() - t0) / steps
        pend, last = [], None
        t0 = time.perf_counter()
        for _ in range(2 * steps):                                        # tickets: 3 in flight
            pend.append(eng.embed_submit(seqs, None, "mean"))
            if len(pend) == 3:
                last = eng.embed_collect(pend.pop(0))
        for h in pend:
            last = eng.embed_collect(h)
        dt = (time.perf_counter() - t0) / (2 * steps)
        assert np.array_equal(last, ref_out), "tickets != blocking call"
        flops = 2.0 * B * L * cfg.layers * (4 * cfg.hidden * cfg.hidden + 2 * cfg.hidden * cfg.intermediate) + 4.0 * B * cfg.layers * L * L * cfg.hidden
        out["embedder_" + name] = {"model": "%s geometry (%d x [%d | %d x 64 | %d]), seeded weights, mean pooling" % (
                                       name.replace("_", "-"), cfg.layers, cfg.hidden, cfg.heads, cfg.intermediate),
                                   "batch": B, "tokens_per_chunk": L, "chunks_per_sec": round(B / dt, 1), "tokens_per_sec": round(B * L / dt, 1),
                                   "ms_per_batch": round(dt * 1e3, 3), "tflops": round(flops / dt / 1e12, 1),
                                   "sync_api": {"chunks_per_sec": round(B / dt_sync, 1), "ms_per_batch": round(dt_sync * 1e3, 3)},
                                   "note": "3 tickets in flight (cqs_hip_bert_embed_submit / _collect); sync_api = one blocking cqs_hip_bert_embed per batch",
                                   "checked": {"min_cosine_vs_fp32_oracle": round(cs, 6)},
                                   "cpu_baseline": {"chunks_per_sec": round(2 / cpu_s, 2), "kind": "port", "sample": "oracle/bert_ref (torch CPU fp32), 2 chunks / 170 tokens"}}
        eng.close()
        del w

    # reranker
    cfg = R.minilm_l6()
    w = R.seeded_weights(cfg, "classifier", seed=2)
    eng = HipBertEngine(bert_config(_lib.BERT_HEAD_CLASSIFIER))
    eng.set_weights(w)
    rr = Reranker(eng)
    small = [rng.integers(1, cfg.vocab_size, size=n).astype(np.int32) for n in (64, 300, 20, 128)]
    st = [np.r_[np.zeros(12, np.int32), np.ones(len(s) - 12, np.int32)] for s in small]
    got = rr.scores(small, st)
    t0 = time.perf_counter()
    want = R.rerank_scores(cfg, w, *padded(small, st))
    cpu_s = time.perf_counter() - t0
    err = float(np.max(np.abs(got - want)))
    assert err < 0.02, "reranker scores differ from the fp32 oracle: %g" % err
    B, L = 32, 512                                                     # the reference's batch (src/reranker.rs:83)
    seqs = [rng.integers(1, cfg.vocab_size, size=L).astype(np.int32) for _ in range(B)]
    tts = [np.r_[np.zeros(16, np.int32), np.ones(L - 16, np.int32)] for _ in range(B)]
    t_end = time.perf_counter() + 0.5                                  # (let the oracle's CPU threads stop spinning)
    while time.perf_counter() < t_end:
        rr.scores(seqs, tts)
    steps = max(20, 4 * a.embed_steps)
    t0 = time.perf_counter()
    for _ in range(steps):
        rr.scores(seqs, tts)
    dt = (time.perf_counter() - t0) / steps
    out["reranker"] = {"model": "MiniLM-L6-H384 cross-encoder geometry (6 x [384 | 12 x 32 | 1536]), seeded weights",
                       "batch": B, "tokens_per_pair": L, "pairs_per_sec": round(B / dt, 1), "ms_per_batch": round(dt * 1e3, 3),
                       "checked": {"max_abs_err_vs_fp32_oracle": round(err, 4)},
                       "cpu_baseline": {"pairs_per_sec": round(4 / cpu_s, 2), "kind": "port", "sample": "oracle/bert_ref (torch CPU fp32), 4 pairs / 512 tokens"}}
    eng.close()
    return out
