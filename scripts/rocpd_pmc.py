#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc results (rocpd SQLite) per kernel family: sum and per-launch mean of every counter.
usage: rocpd_pmc.py results.db [results2.db ...]"""
import sqlite3, sys, re, collections
for path in sys.argv[1:]:
    db = sqlite3.connect(path); cur = db.cursor()
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for name, cname, val, dur in cur.execute("select name, counter_name, counter_value, duration from pmc_events"):
        nm = re.sub(r"\(.*", "", name)
        a = agg[(nm, cname)]; a[0] += 1; a[1] += val; a[2] += dur
    print(f"# {path}")
    print(f"{'kernel':50s} {'counter':34s} {'launches':>8s} {'sum':>16s} {'mean/launch':>16s} {'kernel_ms':>10s}")
    for (nm, cname), a in sorted(agg.items(), key=lambda kv: -kv[1][2]):
        print(f"{nm[-50:]:50s} {cname:34s} {a[0]:8d} {a[1]:16.4e} {a[1]/a[0]:16.4e} {a[2]/1e6:10.2f}")
