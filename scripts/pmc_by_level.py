#!/usr/bin/env python3
"""HBM traffic of the Schur kernel per DAG level: joins two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; rocpd SQLite) of a
SERIAL-schedule bench run (SLUAMD_NO_LOOKAHEAD=1: one k_schur launch per level and tile-size group, in level order) with the library's own per-launch
table (SLUAMD_PROFILE_DUMP: "SCHUR level l pass p big b tiles t ...") and the algorithmic destination bytes of every level (scripts/level_flops.py).
usage: pmc_by_level.py fetch.db write.db schur_dump.txt level_flops.json [records_bytes_big records_bytes_small] > profiles/rNN_pmc_by_level.txt"""
import json, re, sqlite3, sys


def launches(path, counter):
    """k_schur dispatches in dispatch order with the counter value (KB) and duration, the mmode-1 record-building launches of the first factorisation excluded"""
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(pmc_events)")]
    order = "dispatch_id" if "dispatch_id" in cols else ("start" if "start" in cols else "rowid")
    out = []
    for name, cname, val, dur in cur.execute(f"select name, counter_name, counter_value, duration from pmc_events order by {order}"):
        if cname != counter:
            continue
        if "k_scatter_values" in name:
            out.append(("#", 0.0, 0.0))
        elif "k_schur" in name and not re.search(r"k_schur<[^>]*, 1, (true|false)>", name):
            out.append((name, val, dur))
    return out


def last_factorisation(seq):
    idx = [i for i, e in enumerate(seq) if e[0] == "#"]
    return [e for e in seq[idx[-1] + 1:]] if idx else seq


fetch = last_factorisation(launches(sys.argv[1], "FETCH_SIZE"))
write = last_factorisation(launches(sys.argv[2], "WRITE_SIZE"))
dump = [tuple(int(x) for x in re.findall(r"level (\d+) pass (\d+) big (\d+) tiles (\d+)", ln)[0]) for ln in open(sys.argv[3]) if ln.startswith("SCHUR level")]
lf = {int(k): v for k, v in json.load(open(sys.argv[4])).items()}
n = len(dump)
# the dump is printed by the PROFILED factorisation (the last one of a bench run): as many launches as the last factorisation of the trace
assert len(fetch) == len(write), (len(fetch), len(write))
if len(fetch) != n:
    print(f"# warning: {len(fetch)} k_schur launches in the trace's last factorisation, {n} in the library's table: aligned from the end")
    m = min(n, len(fetch)); fetch, write, dump = fetch[-m:], write[-m:], dump[-m:]
per = {}
for (lvl, ps, big, tiles), f, w in zip(dump, fetch, write):
    p = per.setdefault(lvl, dict(tiles=0, fetch=0.0, write=0.0, us=0.0, big=0, small=0))
    p["tiles"] += tiles; p["fetch"] += f[1] * 1024.0 * 2.0; p["write"] += w[1] * 1024.0; p["us"] += f[2] / 1e3   # FETCH_SIZE x 2 on gfx950 (MI355X_MICROARCH.md)
    p["big" if big else "small"] += tiles
rec_big = float(sys.argv[5]) if len(sys.argv) > 5 else 4.0 * (64 + 128 + 3 * 128)
rec_small = float(sys.argv[6]) if len(sys.argv) > 6 else 4.0 * (64 + 64 + 3 * 64)
print("# HBM traffic of k_schur per DAG level, serial schedule; fetch = FETCH_SIZE x 2 x 1024 B, write = WRITE_SIZE x 1024 B; alg = 16 B per updated element;")
print("# records = per-tile record bytes streamed (2.3 / 1.3 kB per 128 / 64 tile); operands = what is left of the fetch after destination reads (alg / 2) and records")
print(f"{'level':>5s} {'nodes':>6s} {'tiles':>8s} {'ms':>8s} {'fetch GB':>9s} {'write GB':>9s} {'alg GB':>8s} {'traffic/alg':>11s} {'records GB':>10s} {'operands GB':>11s} {'TB/s':>6s}")
T = dict(fetch=0.0, write=0.0, alg=0.0, rec=0.0, us=0.0)
classes = {}
for lvl in sorted(per):
    p = per[lvl]; a = lf.get(lvl, {}).get("dest_bytes", 0.0)
    rec = p["big"] * rec_big + p["small"] * rec_small
    tot = p["fetch"] + p["write"]
    print(f"{lvl:5d} {lf.get(lvl, {}).get('nodes', 0):6d} {p['tiles']:8d} {p['us'] / 1e3:8.3f} {p['fetch'] / 1e9:9.3f} {p['write'] / 1e9:9.3f} {a / 1e9:8.3f} "
          f"{(tot / a if a else 0):11.3f} {rec / 1e9:10.3f} {(p['fetch'] - a / 2 - rec) / 1e9:11.3f} {tot / (p['us'] * 1e-6) / 1e12 if p['us'] else 0:6.2f}")
    for k2, v in (("fetch", p["fetch"]), ("write", p["write"]), ("alg", a), ("rec", rec), ("us", p["us"])): T[k2] += v
    cname = "levels 0-3 (leaves, 64 x 64 tiles)" if lvl <= 3 else "levels 4-10" if lvl <= 10 else "levels 11-29" if lvl <= 29 else "levels 30-59" if lvl <= 59 else "levels 60+ (top separator)"
    c = classes.setdefault(cname, dict(fetch=0.0, write=0.0, alg=0.0, rec=0.0, us=0.0))
    for k2, v in (("fetch", p["fetch"]), ("write", p["write"]), ("alg", a), ("rec", rec), ("us", p["us"])): c[k2] += v
print("# by level class: fetch GB, write GB, algorithmic GB, (fetch + write) / algorithmic, records GB, excess over algorithmic GB, ms")
for cname, c in classes.items():
    print(f"# {cname:36s} {c['fetch'] / 1e9:8.2f} {c['write'] / 1e9:8.2f} {c['alg'] / 1e9:8.2f} {(c['fetch'] + c['write']) / c['alg'] if c['alg'] else 0:6.3f} {c['rec'] / 1e9:7.2f} "
          f"{(c['fetch'] + c['write'] - c['alg']) / 1e9:8.2f} {c['us'] / 1e3:8.2f}")
print(f"# total: fetch {T['fetch'] / 1e9:.1f} GB + write {T['write'] / 1e9:.1f} GB = {(T['fetch'] + T['write']) / 1e9:.1f} GB against {T['alg'] / 1e9:.1f} GB algorithmic = "
      f"{(T['fetch'] + T['write']) / T['alg']:.3f} x; records {T['rec'] / 1e9:.1f} GB; {T['us'] / 1e3:.1f} ms")
