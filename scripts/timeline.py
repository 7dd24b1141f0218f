#!/usr/bin/env python3
"""Timeline of one factorisation from a rocprofv3 --kernel-trace rocpd database: per kernel launch (start, duration, queue),
the union of the Schur-kernel busy time, and what ran while no Schur kernel was running.
usage: timeline.py results.db [which_factorisation=1] > profiles/xxx_timeline.txt"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
which = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
qcol = next((c for c in cols if c in ("queue_id", "queue", "stream_id", "stream")), None)
rows = cur.execute(f"select {name_col}, start, end" + (f", {qcol}" if qcol else ", 0") + " from kernels order by start").fetchall()
rows = [(re.sub(r"\(.*", "", n).replace("void sluamd::", "").replace("sluamd::", ""), s, e, q) for n, s, e, q in rows]
# factorisations are delimited by k_scatter_values (device-side re-distribution) launches
marks = [i for i, r in enumerate(rows) if r[0].startswith(("k_scatter_values", "kz_scatter_values"))]
if len(marks) > which:
    lo, hi = marks[which], marks[which + 1] if which + 1 < len(marks) else len(rows)
else:
    lo, hi = 0, len(rows)
seg = [r for r in rows[lo:hi] if not r[0].startswith(("k_fwd", "k_bwd", "k_solve", "k_sweep", "k_full_inv", "k_rfs", "kz_fwd", "kz_bwd", "kz_solve", "__amd"))]
t0 = seg[0][1]
print("# columns in kernels table:", cols)
print("# launches of one factorisation: idx start_ms dur_us queue kernel")
for i, (n, s, e, q) in enumerate(seg):
    print(i, f"{(s - t0) / 1e6:9.3f}", f"{(e - s) / 1e3:9.1f}", q, n)
sch = sorted((s, e) for n, s, e, q in seg if n.startswith("k_schur"))
busy, cur_s, cur_e, gaps = 0, None, None, []
for s, e in sch:
    if cur_e is None: cur_s, cur_e = s, e
    elif s <= cur_e: cur_e = max(cur_e, e)
    else:
        busy += cur_e - cur_s; gaps.append((cur_e, s)); cur_s, cur_e = s, e
busy += cur_e - cur_s
span = seg[-1][2] - t0
print(f"# span {span / 1e6:.2f} ms, Schur-busy union {busy / 1e6:.2f} ms, no-Schur time {(span - busy) / 1e6:.2f} ms")
print("# gaps without a running Schur kernel (> 100 us): start_ms len_us kernels running inside")
for a, b in [(t0, sch[0][0])] + gaps:
    if b - a > 100e3:
        inside = [f"{n}:{(min(e, b) - max(s, a)) / 1e3:.0f}" for n, s, e, q in seg if not n.startswith("k_schur") and s < b and e > a]
        print(f"{(a - t0) / 1e6:9.3f} {(b - a) / 1e3:9.1f}  " + " ".join(inside))
