"""BASELINE configs[3]: chunks -> embed pipeline -> index -> k-NN, with R@K against the CPU oracle on a sub-sample."""
import time


def e2e_leg(a, torch, np, dev, eng, cfg, weights):
    """BASELINE configs[3]: synthetic code chunks (log-normal token lengths) -> GPU EmbeddingGemma forward through the
    index pipeline (length-sorted batches, cqs_amd.pipeline) -> L2 normalise -> HIP index -> 256 queries k-NN.
    R@5 / R@20: the CPU-oracle pipeline (fp32 forward + oracle scan) on a sub-sample at the real geometry."""
    try:
        from cqs_amd.pipeline import EmbedPipeline
    except Exception as e:  # pipeline module not built yet
        return {"skipped": f"pipeline unavailable: {e}"}
    from cqs_amd import HipIndex
    rng = np.random.default_rng(0xC950008)
    n = a.e2e_chunks
    lens = np.clip(np.exp(rng.normal(np.log(300.0), 0.6, size=n)).astype(int), 8, cfg.max_seq)
    V = cfg.vocab_size
    chunks = [rng.integers(1, V, size=int(L)).astype(np.int64) for L in lens]
    pipe = EmbedPipeline(eng)
    pipe.embed_token_lists(chunks[:256])             # warm-up (scratch sizes)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    emb = pipe.embed_token_lists(chunks)             # [n, 768] f32, L2-normalised, input order
    t_embed = time.perf_counter() - t0
    t1 = time.perf_counter()
    idx = HipIndex.build_from_flat(None, emb[:4096])
    for lo in range(4096, n, 32768):
        idx.extend(None, emb[lo:lo + 32768])
    nq, k = 256, 20
    qrows = rng.choice(n, size=nq, replace=False)
    noise = rng.standard_normal((nq, emb.shape[1])).astype(np.float32) * np.float32(0.02)
    queries = emb[qrows] + noise
    queries /= np.linalg.norm(queries, axis=1, keepdims=True)
    got_rows, got_scores, counts = idx.search_batch(queries.astype(np.float32), k)
    t_index = time.perf_counter() - t1
    total = time.perf_counter() - t0
    assert bool(np.all(counts == k)) and np.all(np.isfinite(got_scores))
    hit1 = float(np.mean(got_rows[:, 0] == qrows))   # a query is its chunk's embedding + 2 % noise
    out = {"workload": f"configs[3]: {n} synthetic chunks (log-normal lengths, median ~300 tokens) -> HIP embed pipeline -> "
                       f"extend -> {nq} queries top-{k}",
           "chunks_per_sec_e2e": round(n / total, 1), "embed_chunks_per_sec": round(n / t_embed, 1),
           "embed_tokens_per_sec": round(float(lens.sum()) / t_embed, 1), "embed_s": round(t_embed, 2),
           "index_and_query_s": round(t_index, 3), "self_hit_at_1": round(hit1, 4), "pipeline": pipe.stats()}
    idx.close()
    # recall against the CPU-oracle pipeline on a sub-sample (the fp32 CPU forward is ~1e4 x slower)
    if a.cpu_seconds > 0:
        from oracle import gemma3_ref as G
        from oracle import oracle
        gc = G.GemmaConfig(vocab_size=cfg.vocab_size, hidden=cfg.hidden, layers=cfg.layers, heads=cfg.heads,
                           kv_heads=cfg.kv_heads, head_dim=cfg.head_dim, intermediate=cfg.intermediate,
                           sliding_window=cfg.sliding_window, sliding_pattern=cfg.sliding_pattern,
                           dense_hidden=cfg.dense_hidden, max_seq=cfg.max_seq)
        budget = max(20.0, 6 * a.cpu_seconds)
        order = np.argsort(lens)[: max(64, n // 4)]           # short chunks first: most chunks per CPU-second
        sub, t_cpu0, ref = [], time.perf_counter(), []
        for lo in range(0, len(order), 16):
            sel = order[lo:lo + 16]
            L = int(max(lens[sel]))
            ids = np.zeros((len(sel), L), np.int64)
            mask = np.zeros((len(sel), L), np.int64)
            for i, c in enumerate(sel):
                ids[i, :lens[c]] = chunks[c]
                mask[i, :lens[c]] = 1
            ref.append(G.forward(gc, weights, ids, mask))
            sub.extend(int(c) for c in sel)
            if time.perf_counter() - t_cpu0 > budget and len(sub) >= 64:
                break
        ref = np.concatenate(ref)
        ref /= np.linalg.norm(ref, axis=1, keepdims=True)
        ref = ref.astype(np.float32)
        sub = np.array(sub)
        hip_sub = np.ascontiguousarray(emb[sub])
        cs = np.sum(hip_sub * ref, axis=1)
        nqs = min(64, len(sub) // 2)
        sidx = HipIndex.build_from_flat(None, hip_sub)
        r5 = r20 = 0.0
        kk = min(20, len(sub) - 1)
        gr, _, gc_ = sidx.search_batch(hip_sub[:nqs], kk)
        for qi in range(nqs):
            ref_ids, _ = oracle.index_search(ref, ref[qi], kk)
            got = [int(x) for x in gr[qi, :int(gc_[qi])]]
            r5 += len(set(ref_ids[:5]) & set(got[:5])) / 5.0
            r20 += len(set(ref_ids[:kk]) & set(got[:kk])) / float(kk)
        sidx.close()
        out["recall_vs_cpu_oracle"] = {"sample_chunks": int(len(sub)), "queries": int(nqs), "R@5": round(r5 / nqs, 4),
                                       "R@20": round(r20 / nqs, 4), "min_cosine_hip_vs_fp32": round(float(cs.min()), 6),
                                       "cpu_forward_s": round(time.perf_counter() - t_cpu0, 1),
                                       "note": "seeded weights: parity unpinned w.r.t. the real checkpoint (no weights offline)"}
    return out
