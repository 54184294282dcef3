"""The search-time forward (cqs_amd/csrc/query_kernels.hip): ONE sequence of <= 128 tokens runs 4 launches per layer +
2 for the head, replayed from a hipGraph - what `Embedder::embed_query` (src/embedder/core.rs:768-856) costs on every
search.  Checked: (1) against the fp32 oracle (oracle/gemma3_ref.py) at lengths {1, 8, 33, 64, 65, 96, 128}, tiny and full
geometry - the reference holds no golden vector for the forward, so numerics stay "parity unpinned" as everywhere on
the embedding path; (2) against the batch path of the same engine build (CQS_HIP_QUERY_PATH=0): cosine >= 0.9999 -
NOT bit-identical, by design: the query kernels split K over a workgroup's four waves, so f32 sums associate
differently (the batch path's own alone-vs-batched bit-identity is tested in test_embed_gpu.py for rows that take the
same path); (3) graph replay == eager launches bit for bit, across changing lengths on one captured graph."""
import numpy as np
import pytest

from oracle import gemma3_ref as G
from test_embed_gpu import SMALL, batch, cos, make

pytestmark = pytest.mark.gpu

LENS = [1, 8, 33, 64, 65, 96, 128]


def _one(cfg, n, seed):
    ids, mask = batch(cfg, [n], seed=seed)
    return ids, mask


def test_query_path_matches_the_oracle_and_the_batch_path(hip, monkeypatch):
    eng_q, w = make(SMALL, seed=31)                      # query path on (default)
    monkeypatch.setenv("CQS_HIP_QUERY_PATH", "0")
    eng_b, _ = make(SMALL, seed=31)                      # the same weights through the batch chain only
    monkeypatch.delenv("CQS_HIP_QUERY_PATH")
    # every row-block edge (blocks of 8 / 16 rows); 65-128: the keys in two halves of 48 / 64 (online softmax across them),
    # incl. the edges of the halves (80 / 81: the last block of 16 queries partly filled; 97: first length with 64-key halves)
    for n in LENS + [2, 4, 5, 9, 15, 16, 17, 24, 25, 31, 32, 40, 48, 49, 56, 63, 66, 80, 81, 95, 97, 112, 113, 127]:
        ids, mask = _one(SMALL, n, seed=100 + n)
        got = eng_q.run(ids, mask)[0]
        ref = G.forward(SMALL, w, ids, mask)[0]
        bat = eng_b.run(ids, mask)[0]
        assert np.all(np.isfinite(got))
        assert cos(got, ref) > 0.999, (n, cos(got, ref))          # bf16 matrix-core operands vs an fp32 oracle
        assert cos(got, bat) > 0.9999, (n, cos(got, bat))         # same operands, different f32 association
        assert np.max(np.abs(got - bat)) < 3e-2 * np.abs(bat).max(), (n, float(np.max(np.abs(got - bat))))
    # 129 tokens and 2-row batches stay on the batch chain: both engines agree bit for bit there
    ids, mask = _one(SMALL, 129, seed=9)
    assert np.array_equal(eng_q.run(ids, mask), eng_b.run(ids, mask))
    ids, mask = batch(SMALL, [8, 33], seed=10)
    assert np.array_equal(eng_q.run(ids, mask), eng_b.run(ids, mask))
    eng_q.close(); eng_b.close()


def test_graph_replay_equals_eager_across_lengths(hip, monkeypatch):
    """One graph per (query length, execution context, variant): the first query of a length runs the chain eagerly (its
    kernels' launch attributes are set outside any capture), the second is captured, later ones replay.  The eager
    engine (CQS_HIP_QUERY_GRAPH=0) = the reference answer, bit for bit.  The counters must show that graphs really were
    captured and replayed and that no capture failed - an engine that silently fell back to eager launches would pass the
    comparison vacuously."""
    eng_g, w = make(SMALL, seed=33)
    monkeypatch.setenv("CQS_HIP_QUERY_GRAPH", "0")
    eng_e, _ = make(SMALL, seed=33)
    monkeypatch.delenv("CQS_HIP_QUERY_GRAPH")
    order = [8, 64, 1, 33, 5, 17, 64, 2, 40, 16, 1, 50, 8, 64, 1, 8, 64, 1, 100, 128, 70, 100, 128, 100]
    for j, n in enumerate(order):
        ids, mask = _one(SMALL, n, seed=200 + j)
        a, b = eng_g.run(ids, mask), eng_e.run(ids, mask)
        assert np.array_equal(a, b), (j, n, float(np.max(np.abs(a - b))))
    st = eng_g.query_graph_stats()
    assert st["failed"] == 0, (st, eng_g.last_error())
    assert st["captured"] >= 3 and st["replays"] >= 5, st         # 8 / 64 / 1 came three times each: eager, capture + replay, replay
    assert st["eager"] == len(set(order)), st                      # one eager chain per distinct length, no more
    se = eng_e.query_graph_stats()
    assert se["captured"] == 0 and se["replays"] == 0 and se["eager"] == len(order), se
    # a long row in between (batch chain on the same stream / scratch), then the graph again
    ids, mask = _one(SMALL, 200, seed=7)
    assert np.array_equal(eng_g.run(ids, mask), eng_e.run(ids, mask))
    ids, mask = _one(SMALL, 12, seed=8)
    assert np.array_equal(eng_g.run(ids, mask), eng_e.run(ids, mask))
    # three tickets in flight, all on the query path (two contexts, each with its own graph + scratch)
    qs = [_one(SMALL, n, seed=300 + n) for n in (9, 30, 64)]
    want = [eng_e.run(i, m) for i, m in qs]
    tickets = [eng_g.submit(i, m) for i, m in qs]
    for t, wv in zip(reversed(tickets), reversed(want)):
        assert np.array_equal(eng_g.collect(t, 1), wv)
    eng_g.close(); eng_e.close()


def test_warm_builds_every_graph_and_the_first_query_replays(hip):
    """`cqs_hip_embedder_warm` (`Embedder::warm`, src/embedder/core.rs:933-957): after it, the FIRST query of any length
    is a graph replay - no eager chain, no capture on the query's clock - and answers what a cold engine answers."""
    eng_w, w = make(SMALL, seed=39)
    eng_c, _ = make(SMALL, seed=39)
    eng_w.warm(128)
    st0 = eng_w.query_graph_stats()
    assert st0["failed"] == 0, (st0, eng_w.last_error())
    assert st0["captured"] >= 2 * 128, st0                           # every length, at least the blocking call's variant on both contexts
    for j, n in enumerate([37, 5, 64, 1, 22, 128, 77]):
        ids, mask = _one(SMALL, n, seed=700 + j)
        a = eng_w.run(ids, mask)
        st = eng_w.query_graph_stats()
        assert st["eager"] == st0["eager"] and st["captured"] == st0["captured"], (n, st0, st)   # nothing built on the query's clock
        assert st["replays"] == st0["replays"] + j + 1
        assert np.array_equal(a, eng_c.run(ids, mask)), n
    t = eng_w.submit(*_one(SMALL, 9, seed=720))
    with pytest.raises(Exception):
        eng_w.warm(8)                                                # a ticket is in flight
    eng_w.collect(t, 1)
    eng_w.close(); eng_c.close()


def test_direct_host_buffers_equal_the_copy_calls(hip, monkeypatch):
    """A query on an idle context reads its token ids from, and writes its vector to, the context's pinned host buffers
    (no H2D / D2H copy calls); CQS_HIP_QUERY_DIRECT=0 keeps the copies.  Same kernels, same arithmetic: bit-identical -
    also when a later query is shorter than the one before it (stale ids past T in the pinned buffer are never read),
    and for tickets queued behind one another (the second on a context takes the copy variant)."""
    eng_d, w = make(SMALL, seed=37)
    monkeypatch.setenv("CQS_HIP_QUERY_DIRECT", "0")
    eng_c, _ = make(SMALL, seed=37)
    monkeypatch.delenv("CQS_HIP_QUERY_DIRECT")
    for j, n in enumerate([64, 3, 17, 1, 48, 8, 8, 33]):
        ids, mask = _one(SMALL, n, seed=500 + j)
        a, b = eng_d.run(ids, mask), eng_c.run(ids, mask)
        assert np.array_equal(a, b), (j, n)
    qs = [_one(SMALL, n, seed=600 + n) for n in (12, 40, 7)]
    want = [eng_c.run(i, m) for i, m in qs]
    tickets = [eng_d.submit(i, m) for i, m in qs]             # contexts 0, 1, 0: the third is queued behind the first
    for t, wv in zip(tickets, want):
        assert np.array_equal(eng_d.collect(t, 1), wv)
    eng_d.close(); eng_c.close()


def test_query_path_full_geometry(hip):
    """EmbeddingGemma's real per-layer geometry (768 | 3 x 256 q, 1 kv | 1152 | Dense 3072), 4 layers incl. one
    full-attention layer, the 262 144-row vocabulary left out (a 4 096-row table): vs the fp32 oracle."""
    cfg = G.GemmaConfig(vocab_size=4096, hidden=768, layers=4, heads=3, kv_heads=1, head_dim=256, intermediate=1152,
                        dense_hidden=3072, sliding_window=512, sliding_pattern=2, max_seq=2048)
    eng, w = make(cfg, seed=35)
    for n in LENS:
        ids, mask = _one(cfg, n, seed=400 + n)
        got = eng.run(ids, mask)[0]
        ref = G.forward(cfg, w, ids, mask)[0]
        assert cos(got, ref) > 0.999, (n, cos(got, ref))
    eng.close()


def test_unfused_attention_chain_in_a_child_process(hip):
    """CQS_HIP_QUERY_FUSE_ATTN=0 (read once per process): attention and o_proj as two launches - the chain that queries of
    49-64 tokens and models with other head counts take - against the fused default, every length class."""
    import os, subprocess, sys, json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, json; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "from test_embed_gpu import SMALL, batch, make\n"
        "eng, w = make(SMALL, seed=31)\n"
        "out = {}\n"
        "for n in (1, 5, 8, 9, 16, 17, 33, 48, 50, 64):\n"
        "    ids, mask = batch(SMALL, [n], seed=100 + n)\n"
        "    out[str(n)] = eng.run(ids, mask)[0].tolist()\n"
        "print('RESULT' + json.dumps(out))\n" % (root, os.path.join(root, "tests")))
    res = {}
    for fuse in ("1", "0"):
        env = dict(os.environ, CQS_HIP_QUERY_FUSE_ATTN=fuse)
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")][0]
        res[fuse] = json.loads(line[6:])
    for n, v in res["1"].items():
        a, b = np.array(v, np.float32), np.array(res["0"][n], np.float32)
        assert cos(a, b) > 0.99999 and np.max(np.abs(a - b)) < 2e-2 * np.abs(b).max(), (n, cos(a, b))
