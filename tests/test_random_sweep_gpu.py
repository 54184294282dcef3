"""Seeded random sweeps: the shapes nobody thought of.  Each case draws a corpus size, dimension, metric, k, query
block, scoring mode, threshold and (half the time) a keep-bitset, and checks the HIP path against the CPU oracle through
the same parity rule as tests/test_scan_gpu.py; the BERT and Gemma forwards get ragged batches of random lengths.
Deterministic (fixed seeds): a failure names its case index."""
import numpy as np
import pytest

from cqs_amd import DistanceMetric, HipIndex, synth
from parity import assert_topk_parity

pytestmark = pytest.mark.gpu
MARGIN = 64


def _case(rng):
    n = int(rng.choice([1, 2, 31, 64, 65, 255, 256, 257, 1000, 4097, 20_000, 70_001]) if rng.random() < 0.6
            else rng.integers(1, 90_000))
    dim = int(rng.choice([4, 8, 60, 64, 96, 128, 252, 256, 384, 768, 772, 1024, 1536, 2048]))
    if n * dim > 40_000_000:
        n = 40_000_000 // dim
    metric = DistanceMetric.Cosine if rng.random() < 0.6 else DistanceMetric.DotProduct
    k = int(rng.choice([1, 2, 5, 20, 100, 500, 1024]))
    b = int(rng.choice([1, 1, 2, 3, 7, 8, 9, 33, 70, 130]))
    mode = int(rng.random() < 0.35)
    thr = float(rng.choice([0.0, 0.02, 0.2])) if mode else 0.0
    keep_density = float(rng.choice([0.0, 0.01, 0.5, 0.97])) if rng.random() < 0.5 else None
    return n, dim, metric, k, b, mode, thr, keep_density


@pytest.mark.parametrize("chunk", range(6))
def test_scan_random_configs(hip, oracle, chunk):
    rng = np.random.default_rng(0x5EED0 + chunk)
    for j in range(7):
        n, dim, metric, k, b, mode, thr, dens = _case(rng)
        seed = int(rng.integers(1, 1 << 30))
        rows = synth.gaussian_unit(n, dim=dim, seed=seed) if metric == DistanceMetric.Cosine else \
            (np.random.default_rng(seed).standard_normal((n, dim)).astype(np.float32) * 0.3)
        if n > 8 and rng.random() < 0.3:                          # duplicates + a NaN row + an Inf row
            rows[n // 2] = rows[0]
            rows[n // 3] = np.nan
            rows[n // 4, 0] = np.inf
        q = synth.gaussian_unit(b, dim=dim, seed=seed + 1)
        keep = None
        if dens is not None:
            bits = np.random.default_rng(seed + 2).random(n) < dens
            keep = np.packbits(bits, bitorder="little")
            keep = np.concatenate([keep, np.zeros((-len(keep)) % 4, np.uint8)]).view(np.uint32)
        idx = HipIndex.build_from_flat(None, rows, metric=metric)
        got_rows, got_scores, counts = idx.search_batch(q, k, keep_bitset=keep, mode=mode, threshold=thr)
        for i in range(b):
            ext_ids, ext_scores = oracle.index_search(rows, q[i], k + MARGIN, keep, mode, thr)
            ref_ids, _ = oracle.index_search(rows, q[i], k, keep, mode, thr)
            c = int(counts[i])
            try:
                assert_topk_parity(got_rows[i, :c], got_scores[i, :c], ext_ids, ext_scores, len(ref_ids))
            except AssertionError as e:
                raise AssertionError(f"chunk {chunk} case {j} query {i}: n={n} dim={dim} {metric} k={k} b={b} mode={mode} "
                                     f"thr={thr} keep={dens}: {e}") from e
        idx.close()


def test_gemma_forward_random_ragged_batches(hip):
    from oracle import gemma3_ref as G
    from test_embed_gpu import SMALL, batch, cos, make
    eng, w = make(SMALL, seed=71)
    rng = np.random.default_rng(72)
    for j in range(6):
        B = int(rng.integers(1, 20))
        lens = [int(x) for x in rng.choice([1, 2, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, SMALL.max_seq], size=B)]
        ids, mask = batch(SMALL, lens, seed=73 + j)
        got = eng.run(ids, mask)
        ref = G.forward(SMALL, w, ids, mask)
        for i in range(B):
            assert cos(got[i], ref[i]) > 0.999, (j, i, lens[i], cos(got[i], ref[i]))
    eng.close()


def test_bert_forward_random_ragged_batches(hip):
    from oracle import bert_ref as R
    from test_bert_gpu import _engine, _padded, cos
    cfg = R.BertConfig(vocab_size=900, hidden=384, layers=2, heads=6, intermediate=768, max_pos=200)
    eng, w = _engine(cfg, "mlm", seed=81)
    rng = np.random.default_rng(82)
    for j in range(5):
        B = int(rng.integers(1, 14))
        lens = [int(x) for x in rng.choice([1, 2, 15, 16, 17, 63, 64, 65, 127, 128, 129, 191, 192, 200], size=B)]
        seqs = [rng.integers(1, cfg.vocab_size, size=n).astype(np.int32) for n in lens]
        dense = eng.splade_dense(seqs)
        ids, mask, _ = _padded(seqs)
        _, want = R.splade_encode_batch(cfg, w, ids, mask, 0.01)
        for i in range(B):
            assert np.max(np.abs(dense[i] - want[i])) < 0.06 and cos(dense[i], want[i]) > 0.999, (j, i, lens[i])
    eng.close()
