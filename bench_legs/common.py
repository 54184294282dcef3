"""Shared helpers of the bench legs (bench.py): synthetic unit rows, the size-independent answer check, host facts."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MFMA_F32_PEAK_TF = 157.3  # dense f32-input MFMA peak (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E vendor peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 achievable)


def make_unit_rows(torch, n, dim, seed, device):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    rows = torch.empty((n, dim), dtype=torch.float32, device=device)
    step = 1 << 18
    for lo in range(0, n, step):  # chunked so the generator scratch stays small
        hi = min(n, lo + step)
        x = torch.randn((hi - lo, dim), generator=g, device=device, dtype=torch.float32)
        x /= x.norm(dim=1, keepdim=True)
        rows[lo:hi] = x
    return rows


def check_topk(torch, np, rows, q, keys_u64, count, k, row_base=0, exhaustive=True, what=""):
    """Size-independent properties of one query's answer (outside any timed region): full count, sorted by
    (score desc, row asc), scores equal a direct fp64 dot of the returned rows to 1e-5, and - exhaustively -
    no more than k-1 rows of the corpus beat the k-th score by more than 2e-6."""
    from cqs_amd import unpack_keys
    r, s = unpack_keys(np.ascontiguousarray(keys_u64))
    assert int(count) == k and len(r) == k, f"{what}: count {count} != {k}"
    assert np.all(np.diff(s) <= 0), f"{what}: not sorted"
    assert all(s[i] > s[i + 1] or r[i] < r[i + 1] for i in range(k - 1)), f"{what}: ties not ordered by row"
    local = torch.from_numpy((r.astype(np.int64) - row_base)).to(rows.device)
    direct = (rows[local].double() @ q.double()).cpu().numpy()
    err = float(np.max(np.abs(direct - s)))
    assert err <= 1e-5, f"{what}: scores differ from a direct fp64 dot by {err}"
    if exhaustive:
        beat = int(((rows @ q) > float(s[-1]) + 2e-6).sum().item())
        assert beat <= k - 1, f"{what}: {beat} rows beat the k-th score"
    return r, s


def file_sha256(path):
    import hashlib
    try:
        return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
    except OSError:
        return None


def physical_cores():
    """One logical CPU per physical core among the CPUs this process may run on (sysfs thread_siblings_list)."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        allowed = list(range(os.cpu_count() or 1))
    seen, out = set(), []
    for c in allowed:
        try:
            sib = open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c).read().strip()
        except OSError:
            sib = str(c)
        if sib not in seen:
            seen.add(sib)
            out.append(c)
    return out
