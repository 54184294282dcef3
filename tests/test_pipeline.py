"""Index-pipeline glue (cqs_amd/pipeline.py; reference: src/cli/pipeline/embedding.rs:160-421,
src/embedder/models.rs:789-817).  CPU tests: the batch planner and the stage contract with a stand-in engine.
GPU tests: submit / collect against the synchronous call and the fp32 oracle."""
import numpy as np
import pytest

from cqs_amd.embedder import EmbedderError
from cqs_amd.pipeline import EmbeddedBatch, EmbedPipeline, GpuEmbedStage, PreparedEmbedding, plan_batches


def test_plan_batches_covers_everything_once_within_budget():
    rng = np.random.default_rng(1)
    lens = np.clip(np.exp(rng.normal(np.log(300), 0.6, size=5000)).astype(int), 0, 2048)
    lens[:5] = [0, 2048, 2048, 1, 0]
    batches = plan_batches(lens, 16384, 256)
    allidx = np.concatenate(batches)
    assert sorted(allidx.tolist()) == list(range(5000))
    prev = 1 << 30
    for b in batches:
        assert 1 <= len(b) <= 256
        assert lens[b].sum() <= 16384 or len(b) == 1
        assert np.all(np.diff(lens[b]) <= 0) and lens[b[0]] <= prev      # longest first, globally sorted
        prev = lens[b[-1]]
    assert len(plan_batches([5000], 16, 4)) == 1 and plan_batches([], 16, 4) == []
    assert [len(b) for b in plan_batches([1] * 10, 1000, 4)] == [4, 4, 2]      # sequence cap


class _FakeEngine:
    """Stands in for HipEmbedEngine on CPU: embedding = [sum, count, first, last] of the tokens."""

    def __init__(self, fail_after=None):
        self.pending = {}
        self.ticket = 0
        self.fail_after = fail_after
        self.submits = 0

    def dim(self):
        return 4

    def max_seq(self):
        return 8

    def submit_ragged(self, tokens, lens):
        self.submits += 1
        if self.fail_after is not None and self.submits > self.fail_after:
            raise EmbedderError("InferenceFailed: injected device failure")
        assert len(self.pending) < 3, "more than 3 tickets in flight"
        rows, off = [], 0
        for L in lens:
            t = tokens[off:off + L]
            off += L
            rows.append([float(t.sum()), float(L), float(t[0]) if L else 0.0, float(t[-1]) if L else 0.0])
        self.ticket += 1
        self.pending[self.ticket] = np.asarray(rows, np.float32).reshape(len(lens), 4)
        return self.ticket

    def collect(self, ticket, batch):
        return self.pending.pop(ticket)

    def abandon(self, ticket):                      # cqs_hip_embed_collect(ticket, NULL): the slot comes back
        self.pending.pop(ticket)


def test_pipeline_keeps_input_order_truncates_and_pipelines():
    eng = _FakeEngine()
    pipe = EmbedPipeline(eng, token_budget=16, max_seqs=3)
    rng = np.random.default_rng(2)
    chunks = [rng.integers(1, 100, size=int(L)) for L in rng.integers(0, 12, size=40)]
    out = pipe.embed_token_lists(chunks, normalize=False)
    for c, row in zip(chunks, out):
        t = c[:8]                                          # truncated to max_seq
        assert row[1] == len(t) and row[0] == t.sum()
    assert pipe.stats()["chunks"] == 40 and not eng.pending
    unit = pipe.embed_token_lists(chunks[:5])              # L2-normalised rows (zero rows stay zero)
    for c, row in zip(chunks[:5], unit):
        assert abs(np.linalg.norm(row) - (1.0 if len(c) else 0.0)) < 1e-6


def test_gpu_stage_failure_contract():
    """gpu_embed_stage :404-421 + flush_to_cpu :160-223: a failing GPU batch is counted, its cached pairs still
    reach the writer, its uncached chunks are requeued for the CPU stage, and the stage keeps going."""
    sent, failed = [], []
    eng = _FakeEngine(fail_after=1)
    stage = GpuEmbedStage(EmbedPipeline(eng, token_budget=64, max_seqs=8), sent.append, failed.append)
    mk = lambda n, base: [np.arange(1, 4) + base + i for i in range(n)]
    batches = [
        PreparedEmbedding(cached=[("c0", "e0")], to_embed=["a", "b"], tokens=mk(2, 0)),       # ok
        PreparedEmbedding(cached=[("c1", "e1")], to_embed=["x", "y", "z"], tokens=mk(3, 10)),  # GPU fails
        PreparedEmbedding(cached=[("c2", "e2"), ("c3", "e3")]),                               # all cached
    ]
    stage.run(batches)
    assert stage.gpu_failures == 3 and failed == [["x", "y", "z"]]
    assert [b.cached_count for b in sent] == [1, 1, 2]
    assert [len(b.chunk_embeddings) for b in sent] == [3, 1, 2]
    assert sent[0].chunk_embeddings[1][0] == "a" and isinstance(sent[0], EmbeddedBatch)
    assert stage.embedded_count == 3 + 1 + 2


def test_failed_submit_gives_every_slot_back():
    """a submit that fails while earlier tickets are in flight must not strand their slots (a slot is
    released only by collecting its ticket; the fake engine keeps the library's 3-slot accounting)."""
    eng = _FakeEngine(fail_after=2)                                   # batch 3 of the call fails, 2 tickets in flight
    pipe = EmbedPipeline(eng, token_budget=4, max_seqs=1)
    chunks = [np.arange(1, 4) + i for i in range(6)]
    with pytest.raises(EmbedderError):
        pipe.embed_token_lists(chunks)
    assert not eng.pending, "tickets stranded after a failed submit"
    eng.fail_after = None
    out = pipe.embed_token_lists(chunks, normalize=False)             # the same engine keeps working
    assert [r[0] for r in out] == [float(c.sum()) for c in chunks] and not eng.pending


@pytest.mark.gpu
def test_failed_batch_does_not_strand_submission_slots(hip):
    """The same on the real engine: an out-of-range token id in batch 2 of 3 fails that submit (not a device error:
    the engine stays healthy); the tickets in flight are abandoned and the next call on the engine succeeds."""
    from oracle import gemma3_ref as G
    from test_embed_gpu import SMALL, batch, cos, make
    eng, w = make(SMALL, seed=23)
    lens = [40, 30, 20, 10, 5, 3]
    ids, mask = batch(SMALL, lens, seed=24)
    chunks = [ids[i, :lens[i]].copy() for i in range(len(lens))]
    ref = G.forward(SMALL, w, ids, mask)
    pipe = EmbedPipeline(eng, token_budget=40, max_seqs=2)            # batches: [40] [30] [20, 10] [5, 3]
    poisoned = [c.copy() for c in chunks]
    poisoned[1][3] = SMALL.vocab_size + 7                             # lands in batch 2
    for _ in range(4):                                                # > 3 slots' worth of failures
        with pytest.raises(EmbedderError):
            pipe.embed_token_lists(poisoned)
    emb = pipe.embed_token_lists(chunks)
    for i in range(len(lens)):
        assert cos(emb[i], ref[i]) > 0.999
    t = eng.submit(ids[:1], mask[:1])                                 # abandon = collect without a buffer
    eng.abandon(t)
    with pytest.raises(EmbedderError):
        eng.collect(t, 1)                                             # the ticket is gone
    eng.close()


@pytest.mark.gpu
def test_submit_collect_matches_sync_and_oracle(hip):
    from oracle import gemma3_ref as G
    from test_embed_gpu import SMALL, batch, cos, make
    eng, w = make(SMALL, seed=9)
    lens = [70, 1, 33, 64, 100, 5, 0, 17]
    ids, mask = batch(SMALL, lens, seed=10)
    sync = eng.run(ids, mask)
    # three tickets in flight, collected out of order; a 4th submit is refused until one is collected
    t1 = eng.submit(ids[:3], mask[:3])
    toks = np.concatenate([ids[i, :lens[i]] for i in range(3, 6)]).astype(np.int32)
    t2 = eng.submit_ragged(toks, np.array(lens[3:6], np.uint32))
    t3 = eng.submit(ids[6:], mask[6:])
    with pytest.raises(EmbedderError):
        eng.submit(ids[:1], mask[:1])
    r3 = eng.collect(t3, 2)
    r1 = eng.collect(t1, 3)
    r2 = eng.collect(t2, 3)
    got = np.concatenate([r1, r2, r3])
    ref = G.forward(SMALL, w, ids, mask)
    for i, L in enumerate(lens):
        if L == 0:
            assert np.all(got[i] == 0) and np.all(sync[i] == 0)
            continue
        assert cos(got[i], sync[i]) > 0.99999 and np.max(np.abs(got[i] - sync[i])) < 2e-2 * np.abs(sync[i]).max()
        assert cos(got[i], ref[i]) > 0.999
    with pytest.raises(EmbedderError):
        eng.collect(12345, 1)                              # unknown ticket
    # the pipeline over the same engine: input order, unit rows
    pipe = EmbedPipeline(eng, token_budget=128, max_seqs=4)
    chunks = [ids[i, :lens[i]] for i in range(len(lens))]
    emb = pipe.embed_token_lists(chunks)
    for i, L in enumerate(lens):
        if L:
            assert cos(emb[i], ref[i]) > 0.999 and abs(np.linalg.norm(emb[i]) - 1.0) < 1e-5
    assert pipe.stats()["chunks"] == len(lens) and pipe.stats()["batches"] >= 3
    bad = [np.array([1, 2, SMALL.vocab_size + 5])]
    with pytest.raises(EmbedderError):
        pipe.embed_token_lists(bad)                        # token id out of range -> InferenceFailed
    assert cos(pipe.embed_token_lists(chunks[:1])[0], ref[0]) > 0.999      # the engine is still usable
    eng.close()


@pytest.mark.gpu
def test_tickets_in_flight_are_bit_identical_to_blocking_calls(hip):
    """Consecutive tickets run on the engine's two execution contexts (own stream + scratch, shared weights) and
    their kernels interleave on the device: every in-flight result must equal, bit for bit, the blocking call on the
    same batch (a scratch or table shared by mistake between the contexts shows up here)."""
    from test_embed_gpu import SMALL, batch, make
    eng, _ = make(SMALL, seed=19)
    rng = np.random.default_rng(20)
    batches = []
    for j in range(6):
        lens = list(rng.integers(1, SMALL.max_seq, size=12))
        batches.append(batch(SMALL, lens, seed=21 + j))
    want = [eng.run(ids, mask).copy() for ids, mask in batches]
    for rep in range(3):
        pend, got = [], {}
        for j, (ids, mask) in enumerate(batches):
            pend.append((j, eng.submit(ids, mask)))
            if len(pend) == 3:
                jj, t = pend.pop(0)
                got[jj] = eng.collect(t, len(batches[jj][0]))
        for jj, t in reversed(pend):                       # the tail out of order
            got[jj] = eng.collect(t, len(batches[jj][0]))
        for j in range(len(batches)):
            assert np.array_equal(got[j], want[j]), (rep, j, float(np.max(np.abs(got[j] - want[j]))))
    eng.close()
