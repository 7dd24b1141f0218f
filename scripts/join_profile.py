"""Development aid: join level_flops.py's table with the SCHUR lines of a SLUAMD_PROFILE_DUMP run (serial profiled step)."""
import sys, json, re, collections
lv = {int(k): v for k, v in json.load(open(sys.argv[1])).items()}
ms = collections.defaultdict(float); tiles = collections.defaultdict(int)
for line in open(sys.argv[2]):
    m = re.match(r"SCHUR level (\d+) pass (\d+) big (\d+) tiles (\d+) max_nsupc (\d+) ms (\S+)", line)
    if m: ms[int(m.group(1))] += float(m.group(6)); tiles[int(m.group(1))] += int(m.group(4))
tot_ms = sum(ms.values())
print("# level nodes big tiles  exact_GF exec_GF  ms  TF_exact frac_of_peak  share_of_time  GB_dest/s")
acc = 0.0
for l in sorted(lv):
    r = lv[l]; t = ms.get(l, 0.0)
    if t <= 0: continue
    acc += t
    print("%3d %6d %5d %8d %10.1f %10.1f %8.3f %7.1f %6.3f %6.3f %6.3f %8.0f" % (l, r["nodes"], r["big"], tiles[l], r["exact"] / 1e9, r["exec_tiles"] / 1e9, t,
          r["exact"] / t / 1e9, r["exact"] / t / 1e9 / 78.6, t / tot_ms, acc / tot_ms, r["dest_bytes"] / t / 1e6))
print("total ms", tot_ms)
