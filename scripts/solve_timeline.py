#!/usr/bin/env python3
"""Development aid: the kernels of ONE pdgstrs3d from a rocprofv3 --kernel-trace rocpd database: busy time, gaps between
consecutive kernels, per-kernel totals.  usage: solve_timeline.py results.db [which_solve=1]"""
import re, sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
which = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
rows = [(re.sub(r"\(.*", "", n).replace("void sluamd::", "").replace("sluamd::", ""), s, e) for n, s, e in rows]
issolve = lambda n: n.startswith(("k_solve_diag", "k_fwd_update", "k_bwd_update", "k_sweep", "kz_solve_diag", "kz_fwd", "kz_bwd", "k_zero_nodes"))
# a solve = maximal run of solve kernels
runs, curr = [], []
for r in rows:
    if issolve(r[0]): curr.append(r)
    elif curr and not r[0].startswith("__amd"): runs.append(curr); curr = []
if curr: runs.append(curr)
runs = [r for r in runs if len(r) > 50]
print("# solves found:", len(runs), [len(r) for r in runs])
seg = runs[min(which, len(runs) - 1)]
t0 = seg[0][1]; span = seg[-1][2] - t0
busy = sum(e - s for _, s, e in seg)
gaps = [seg[i + 1][1] - seg[i][2] for i in range(len(seg) - 1)]
print(f"# span {span / 1e3:.1f} us, kernel busy {busy / 1e3:.1f} us, gaps {sum(gaps) / 1e3:.1f} us (avg {sum(gaps) / len(gaps) / 1e3:.2f} us over {len(gaps)})")
tot = collections.defaultdict(lambda: [0, 0.0])
for n, s, e in seg: tot[n][0] += 1; tot[n][1] += (e - s) / 1e3
for n, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]): print(f"{n:40s} {c:5d} {t:9.1f} us  avg {t / c:7.2f}")
print("# idx start_us dur_us gap_before_us kernel")
for i, (n, s, e) in enumerate(seg):
    print(i, f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(s - seg[i - 1][2]) / 1e3 if i else 0:7.2f}", n)
