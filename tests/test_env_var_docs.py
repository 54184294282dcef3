"""Every CQS_* environment variable that product code reads is documented in README.md's table - the reference's own
policy (tests/env_var_docs.rs: "the README drifts behind when new env vars are introduced"), same token-boundary rule
(a short name is not satisfied by being the prefix of a longer documented one)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
READ = re.compile(r"""(?:getenv\(\s*"|environ(?:\.get\(|\[)\s*["']|env::var\(\s*")(CQS_[A-Z][A-Z0-9_]*[A-Z0-9])""")


def _read_vars():
    out = {}
    roots = [os.path.join(ROOT, "cqs_amd"), os.path.join(ROOT, "rust_shim"), os.path.join(ROOT, "bench.py"),
             os.path.join(ROOT, "__graft_entry__.py")]
    for r in roots:
        files = [r] if os.path.isfile(r) else [os.path.join(d, f) for d, _, fs in os.walk(r) for f in fs
                                                if f.endswith((".py", ".hip", ".h", ".cpp", ".rs"))]
        for f in files:
            for m in READ.finditer(open(f, errors="replace").read()):
                out.setdefault(m.group(1), os.path.relpath(f, ROOT))
    return out


def _documented(readme, var):
    return re.search(r"(?<![A-Za-z0-9_])" + re.escape(var) + r"(?![A-Za-z0-9_])", readme) is not None


def test_every_env_var_read_is_in_the_readme_table():
    readme = open(os.path.join(ROOT, "README.md")).read()
    found = _read_vars()
    assert len(found) >= 13 and "CQS_HIP_DEVICES" in found, found
    missing = {v: f for v, f in found.items() if not _documented(readme, v)}
    assert not missing, f"read but not documented in README.md: {missing}"


def test_token_match_is_not_a_prefix_match():
    assert _documented("| `CQS_HIP_GEMM_TILE` |", "CQS_HIP_GEMM_TILE")
    assert not _documented("| `CQS_HIP_GEMM_TILE_X` |", "CQS_HIP_GEMM_TILE")
