#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite result (--kernel-trace [--stats]) into a per-kernel CSV-like table:
name, calls, total_ms, avg_us, min_us, max_us, pct.  usage: rocpd_stats.py results.db > profiles/xxx.txt"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select {name_col}, start, end from kernels").fetchall()
if len(sys.argv) > 2 and sys.argv[2] == "--launches":   # per-launch listing of one kernel family
    pat = sys.argv[3]
    gcols = [c for c in cols if c in ("grid_size_x", "grid_x", "workgroup_size_x", "grid_size")]
    q = cur.execute(f"select {name_col}, start, end" + "".join(", " + c for c in gcols) + " from kernels order by start").fetchall()
    print("# launches of kernels matching", pat, "cols: idx dur_us", gcols)
    for i, r in enumerate(q):
        if pat in r[0]:
            print(i, f"{(r[2]-r[1])/1e3:.1f}", *r[3:], re.sub(r"\(.*", "", r[0])[-40:])
    sys.exit(0)
agg = {}
for nm, s, e in rows:
    nm = re.sub(r"\(.*", "", nm)
    a = agg.setdefault(nm, [0, 0.0, 1e30, 0.0]); d = (e - s) / 1e3
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
tot = sum(a[1] for a in agg.values()) or 1.0
print(f"{'kernel':60s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
for nm, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{nm[:60]:60s} {a[0]:7d} {a[1]/1e3:10.3f} {a[1]/a[0]:10.2f} {a[2]:10.2f} {a[3]:10.2f} {100*a[1]/tot:6.2f}")
