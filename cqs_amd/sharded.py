"""Row-sharded exact search over several GPUs: one process per GPU, `torch.distributed`
(backend "nccl" = RCCL over xGMI) for the one real exchange step of the path.

No reference counterpart (cqs is single-GPU): the corpus is partitioned
contiguously by row (rowid order) into `world_size` shards, every rank scans its own shard
with the same query block and emits its local top-k as packed keys carrying GLOBAL row ids
(`row_base` = shard offset), one all-gather moves the `k x 8 B` candidates of every rank
(k=20: 160 B per query per GPU - latency-, not bandwidth-bound, so ONE collective per
query block; zero-padded key lists carry their own length), and the final k-way merge runs on the host with the same comparator
(score desc, row asc).  The comparator is a total order on distinct keys, so the merged
list equals the single-GPU answer exactly.

torch is plumbing here (device memory, streams, the process group); the scan itself is
libcqs_hip.so.  `local_search` is injectable so the collective + merge logic can be covered
on CPU with the gloo backend (tests/test_sharded_cpu.py).
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np

from .index import merge_keys


def shard_bounds(n: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous, near-equal row ranges in rowid order."""
    per, rem = divmod(n, world_size)
    lo = rank * per + min(rank, rem)
    return lo, lo + per + (1 if rank < rem else 0)


class ShardedSearch:
    """Collective top-k over row shards.

    local_search(queries_tensor[b,dim], k) -> (keys int64[b,k] tensor, counts int32[b] tensor)
    on this rank's device, keys packed as in include/cqs_hip.h and sorted descending.
    """

    def __init__(self, local_search: Callable, k_max: int, group=None, all_gather: Optional[Callable] = None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        inited = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if inited else 1   # no process group = one shard
        self.rank = dist.get_rank(group) if inited else 0
        self.local_search = local_search
        self.k_max = k_max
        self._all_gather = all_gather   # optional override: f(out[world, ...], inp) (bench rehearsal mode)

    def gather_candidates(self, queries, k: int):
        """Local scan + ONE all-gather of the packed keys -> int64 tensor [world, b, k] on device.

        No separate count travels: unused key slots are zero (include/cqs_hip.h: out_keys is zero padded;
        a valid key is never zero because every finite score has a non-zero ordered image), so the count
        of a list is its number of non-zero keys."""
        import torch
        keys, _counts = self.local_search(queries, k)
        b = keys.shape[0]
        if self.world == 1:
            return keys.view(1, b, k)
        out = torch.empty((self.world, b, k), dtype=torch.int64, device=keys.device)
        if self._all_gather is not None:
            self._all_gather(out, keys)
        else:
            try:
                self.dist.all_gather_into_tensor(out.view(-1), keys.contiguous().view(-1), group=self.group)
            except (RuntimeError, NotImplementedError):  # backends without the fused form (older gloo)
                self.dist.all_gather(list(out.unbind(0)), keys.contiguous(), group=self.group)
        return out

    @staticmethod
    def merge_host(gathered: np.ndarray, k: int):
        """gathered: [world, b, k] int64 (host), zero padded.  -> list of per-query merged key arrays (uint64)."""
        world, b, _ = gathered.shape
        res = []
        for q in range(b):
            lists = np.ascontiguousarray(gathered[:, q, :k]).view(np.uint64)
            counts = np.count_nonzero(lists, axis=1).astype(np.uint32)
            res.append(merge_keys(lists, counts, k))
        return res

    @staticmethod
    def merge_host_many(gathered: np.ndarray, k: int) -> np.ndarray:
        """Vectorised host merge of many query blocks at once: gathered [..., world, b, k] int64 (zero padded)
        -> uint64 [..., b, k], zero padded.  The keys of one query are distinct and totally ordered, so the
        k-way merge of the per-shard lists equals the k largest keys of their concatenation."""
        g = np.ascontiguousarray(np.moveaxis(gathered, -3, -2)).view(np.uint64)   # [..., b, world, k]
        flat = g.reshape(g.shape[:-2] + (g.shape[-2] * g.shape[-1],))
        return np.ascontiguousarray(np.sort(flat, axis=-1)[..., ::-1][..., :k])

    def search(self, queries, k: int):
        """Every rank gets the merged global top-k of every query (host arrays)."""
        g = self.gather_candidates(queries, k)
        return self.merge_host(g.cpu().numpy(), k)


def hip_local_search(index, device: Optional[int] = None):
    """`local_search` backed by `cqs_hip_index_search_device` on torch's current stream."""
    import torch

    def fn(queries, k: int):
        b = queries.shape[0]
        keys = torch.empty((b, k), dtype=torch.int64, device=queries.device)
        counts = torch.empty((b,), dtype=torch.int32, device=queries.device)
        st = torch.cuda.current_stream(queries.device).cuda_stream
        index.search_device(queries.data_ptr(), b, k, keys.data_ptr(), counts.data_ptr(), stream=st)
        return keys, counts

    return fn
