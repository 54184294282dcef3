"""N>1 path on CPU: world_size-2 gloo process group, per-shard candidates produced by the
oracle (test infrastructure stands in for the HIP local scan), ONE all-gather, host k-way merge
through the C ABI helper.  The merged answer must equal the whole-corpus answer exactly."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cqs_amd import synth
from cqs_amd.sharded import ShardedSearch, shard_bounds


def _pack(scores: np.ndarray, rows: np.ndarray) -> np.ndarray:
    bits = scores.astype(np.float32).view(np.uint32).astype(np.uint64)
    ok = np.where(bits >> np.uint64(31) != 0, (~bits) & np.uint64(0xFFFFFFFF), bits ^ np.uint64(0x80000000))
    return (ok << np.uint64(32)) | (np.uint64(0xFFFFFFFF) - rows.astype(np.uint64))


def _worker(rank, world, port, n, k, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle
    from cqs_amd import unpack_keys
    rows = synth.gaussian_unit(n, 64, seed=5)
    rows[n // 2 + 3] = rows[7]          # a bit-identical tie across the shard boundary
    qs = synth.gaussian_unit(3, 64, seed=6)
    qs[2] = rows[7]
    lo, hi = shard_bounds(n, world, rank)

    def local_search(queries, kk):
        q = queries.numpy()
        keys = np.zeros((q.shape[0], kk), np.uint64)
        counts = np.zeros((q.shape[0],), np.int32)
        for i in range(q.shape[0]):
            ids, sc = oracle.index_search(rows[lo:hi], q[i], kk)
            keys[i, :len(ids)] = _pack(sc, ids + lo)   # global row ids = row_base + local
            counts[i] = len(ids)
        return torch.from_numpy(keys.view(np.int64)), torch.from_numpy(counts)

    shard = ShardedSearch(local_search, k)
    merged = shard.search(torch.from_numpy(qs), k)
    ok = True
    for i in range(3):
        r, s = unpack_keys(merged[i])
        ids, sc = oracle.index_search(rows, qs[i], k)
        ok &= list(r) == list(ids) and np.array_equal(s, sc)
    np.save(os.path.join(out_dir, f"ok{rank}.npy"), np.array([ok]))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,n,k", [(2, 1001, 20), (2, 37, 50)])
def test_sharded_search_gloo(tmp_path, oracle, world, n, k):
    from cqs_amd import _lib
    _lib.load()  # the host merge is part of the C ABI (no device call)
    mp.spawn(_worker, args=(world, _free_port(), n, k, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert np.load(tmp_path / f"ok{r}.npy")[0], f"rank {r}: merged != whole-corpus answer"


def test_shard_bounds_cover():
    for n in (0, 1, 7, 8, 1000, 1001):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


def test_merge_host_many_equals_kway_merge():
    """The vectorised host merge (k largest keys of the concatenated lists) == cqs_hip_merge_keys per query."""
    from cqs_amd.sharded import ShardedSearch
    rng = np.random.default_rng(5)
    steps, world, b, k = 7, 4, 3, 20
    g = np.zeros((steps, world, b, k), np.uint64)
    for s in range(steps):
        for q in range(b):
            keys = np.unique(rng.integers(1, 1 << 62, size=4 * world * k, dtype=np.uint64))[:world * k]
            rng.shuffle(keys)
            for w in range(world):
                c = int(rng.integers(0, k + 1))
                g[s, w, q, :c] = np.sort(keys[w * k:w * k + c])[::-1]
    many = ShardedSearch.merge_host_many(g.view(np.int64), k)
    for s in range(steps):
        one = ShardedSearch.merge_host(g[s].view(np.int64), k)
        for q in range(b):
            got = many[s, q]
            assert np.array_equal(got[got != 0], one[q])
