#!/usr/bin/env python3
"""Per-kernel durations out of a rocprofv3 rocpd database (the default --kernel-trace output of ROCm 7.2):
prof_db.py DIR [DIR ...] -> kernel, calls, avg / min / max us, share."""
import glob, re, sqlite3, sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"_ZN3cqs(?:12_GLOBAL__N_1)?(\d+)([A-Za-z_0-9]+)", name)
    if m:
        n = int(m.group(1)); base = m.group(2)[:n]
        t = re.search(r"I((?:L[ijb]\d+E)+)E", name)
        return "cqs::" + base + ("<" + ",".join(re.findall(r"L[ijb](\d+)E", t.group(1))) + ">" if t else "")
    return name.split("(")[0][:80]


for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*_results.db", recursive=True):
        con = sqlite3.connect(f)
        tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
        kd = [t for t in tabs if "kernel_dispatch" in t][0]
        ks = [t for t in tabs if "kernel_symbol" in t][0]
        rows = con.execute(f"select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), sum(d.end-d.start) "
                           f"from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 6 desc").fetchall()
        tot = sum(r[5] for r in rows) or 1
        print(f"# {f}")
        print(f"{'kernel':60s} {'calls':>7s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
        for name, c, a, mn, mx, sm in rows:
            print(f"{short(name):60s} {c:7d} {a/1e3:9.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100*sm/tot:6.2f}")
