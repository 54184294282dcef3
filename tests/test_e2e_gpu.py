"""End-to-end slice of BASELINE configs[3] in miniature: chunks -> HIP embedding forward -> L2 normalise ->
HIP exact index -> top-k, against the same pipeline run entirely on the CPU oracle (fp32 forward + oracle
brute-force scan).  Recall of the oracle's neighbours among the HIP pipeline's (R@5 / R@20).  The two
pipelines differ only by the bf16 matrix-core operands of the forward, so near-ties may swap; the bound
is written in the test.  Seeded weights (no checkpoint in the image): "parity unpinned" w.r.t. the real
model, as everywhere on the embedding path."""
import numpy as np
import pytest

from cqs_amd import HipIndex
from cqs_amd.embedder import normalize_l2
from oracle import gemma3_ref as G
from test_embed_gpu import SMALL, batch, make

pytestmark = pytest.mark.gpu


def test_embed_then_search_recall(hip, oracle):
    n_docs, n_q, k = 1500, 40, 20
    eng, w = make(SMALL, seed=5)
    rng = np.random.default_rng(11)
    lens = [int(x) for x in rng.integers(4, 48, size=n_docs + n_q)]
    ids, mask = batch(SMALL, lens, seed=12)
    hip_emb, ref_emb = [], []
    for lo in range(0, len(lens), 64):                       # embed_batch_size-style batches
        i, m = ids[lo:lo + 64], mask[lo:lo + 64]
        hip_emb.append(eng.run(i, m))
        ref_emb.append(G.forward(SMALL, w, i, m))
    hip_emb = np.stack([normalize_l2(v) for v in np.concatenate(hip_emb)]).astype(np.float32)
    ref_emb = np.stack([normalize_l2(v) for v in np.concatenate(ref_emb)]).astype(np.float32)
    eng.close()
    cs = np.sum(hip_emb * ref_emb, axis=1)
    assert cs.min() > 0.999, cs.min()

    idx = HipIndex.build_from_flat(None, np.ascontiguousarray(hip_emb[:n_docs]))
    got_rows, _, counts = idx.search_batch(np.ascontiguousarray(hip_emb[n_docs:]), k)
    idx.close()
    r5 = r20 = 0.0
    for qi in range(n_q):
        ref_ids, _ = oracle.index_search(np.ascontiguousarray(ref_emb[:n_docs]), ref_emb[n_docs + qi], k, None, 0, 0.0)
        got = [int(x) for x in got_rows[qi, :int(counts[qi])]]
        r5 += len(set(ref_ids[:5]) & set(got[:5])) / 5.0
        r20 += len(set(ref_ids[:k]) & set(got[:k])) / float(k)
    r5, r20 = r5 / n_q, r20 / n_q
    print(f"R@5 = {r5:.3f}  R@20 = {r20:.3f}")
    assert r5 >= 0.9 and r20 >= 0.9, (r5, r20)
