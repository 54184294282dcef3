"""Parity checker: HIP result vs the CPU oracle, with the reference's tolerance rules.

Bar (BASELINE.json north_star, SURVEY.md §8a caveat 4): identical top-k id sequence
wherever adjacent oracle scores differ by more than TIE_EPS, set-equality inside such
near-tie groups, |score - oracle score| <= SCORE_TOL (1e-5, fp32 cosine).
"""
import numpy as np

SCORE_TOL = 1e-5
TIE_EPS = 2e-6


def assert_topk_parity(gpu_rows, gpu_scores, ext_ids, ext_scores, k_expected):
    """gpu_*: the HIP top-k (already cut to its count).  ext_*: the oracle's top-(k+margin)
    list in reference order.  k_expected: how many results the reference returns."""
    gpu_rows = np.asarray(gpu_rows).astype(np.int64)
    gpu_scores = np.asarray(gpu_scores, dtype=np.float32)
    ext_ids = np.asarray(ext_ids).astype(np.int64)
    ext_scores = np.asarray(ext_scores, dtype=np.float32)
    assert len(gpu_rows) == k_expected, f"count {len(gpu_rows)} != reference {k_expected}"
    if k_expected == 0:
        return
    # own ordering: score desc, row asc on exact ties
    for i in range(len(gpu_rows) - 1):
        assert gpu_scores[i] > gpu_scores[i + 1] or (
            gpu_scores[i] == gpu_scores[i + 1] and gpu_rows[i] < gpu_rows[i + 1]), f"order violated at {i}"
    score_of = {int(r): float(s) for r, s in zip(ext_ids, ext_scores)}
    for r, s in zip(gpu_rows, gpu_scores):
        assert int(r) in score_of, f"row {r} is not among the oracle's top-{len(ext_ids)}"
        assert abs(score_of[int(r)] - float(s)) <= SCORE_TOL, f"score of row {r}: {s} vs oracle {score_of[int(r)]}"
    # near-tie groups over the extended oracle list
    start = 0
    n = len(ext_ids)
    while start < k_expected:
        end = start + 1
        while end < n and abs(float(ext_scores[end - 1]) - float(ext_scores[end])) <= TIE_EPS:
            end += 1
        grp = set(int(x) for x in ext_ids[start:end])
        got = set(int(x) for x in gpu_rows[start:min(end, k_expected)])
        if end <= k_expected:
            assert got == grp, f"positions [{start},{end}) differ: {sorted(got)} vs oracle {sorted(grp)}"
        else:  # group straddles the k boundary (or runs past the margin)
            assert got <= grp, f"positions [{start},{k_expected}) not within the oracle's tie group"
        start = end
