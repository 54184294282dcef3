"""bench.py's leg isolation, on the CPU: `Legs.run` turns an exception into {"error": ...}, honours the budget, and no
leg module indexes the headline's `--steps` / `--warmup` arrays (round 4: the hybrid sub-leg ran 40 queries out of K = 20)."""
import ast
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_an_exception_becomes_an_error_entry():
    import bench
    legs = bench.Legs(budget_s=1e9)

    def boom(x):
        raise IndexError("index %d is out of bounds" % x)

    assert legs.run("fine", lambda: {"ok": 1}) == {"ok": 1}
    got = legs.run("hybrid", boom, 20)
    assert got == {"error": "IndexError: index 20 is out of bounds"} and legs.errors == {"hybrid": got["error"]}
    assert set(legs.seconds) == {"fine", "hybrid"}
    assert legs.run("exit", lambda: sys.exit(2))["error"].startswith("SystemExit")     # even a SystemExit stays inside


def test_the_budget_skips_instead_of_running():
    import bench
    legs = bench.Legs(budget_s=-1.0)
    ran = []
    assert "skipped" in legs.run("late", lambda: ran.append(1))
    assert legs.run("cpu_baseline", lambda: ran.append(2) or {"v": 1}, budgeted=False) == {"v": 1}
    assert ran == [2] and not legs.errors


def test_every_leg_is_called_under_the_guard():
    """Structural: in bench.main() every function imported from bench_legs.* is called through legs.run (or, for the legs
    that carry their own watchdog thread / must stay collective across ranks, is named in the allow-list below)."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    main = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main"][0]
    imported = set()
    for n in ast.walk(main):
        if isinstance(n, ast.ImportFrom) and n.module and n.module.startswith("bench_legs."):
            imported |= {a.name for a in n.names}
    assert {"concurrent_clients_leg", "other_configs", "abi_sharded_leg", "sparse_index_leg", "embed_leg", "aux_models_leg", "e2e_leg",
            "cpu_baseline", "embed_cpu_baseline"} <= imported
    own_watchdog = {"abi_after_group_leg", "strong_n1_leg"}
    direct = set()
    for n in ast.walk(main):
        if isinstance(n, ast.Call) and isinstance(n.func, ast.Name) and n.func.id in imported:
            direct.add((n.func.id, n.lineno))
    guarded_lines = set()
    for n in ast.walk(main):
        if isinstance(n, ast.Call) and isinstance(n.func, ast.Attribute) and n.func.attr == "run" and getattr(n.func.value, "id", "") == "legs":
            for sub in ast.walk(n):
                guarded_lines.add(getattr(sub, "lineno", -1))
    bad = [(f, l) for f, l in direct if l not in guarded_lines and f not in own_watchdog
           and not (f == "embed_leg" and "no isolation here" in src.splitlines()[l - 2])]
    assert not bad, bad


def test_no_leg_takes_the_headline_queries_by_count():
    for path in glob.glob(os.path.join(ROOT, "bench_legs", "*.py")):
        text = open(path).read()
        assert "dense_queries" not in text and "queries[W" not in text, path
