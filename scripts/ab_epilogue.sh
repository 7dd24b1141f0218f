#!/bin/bash
# where the Schur kernel's missing 0.40 goes -- same box, serial (profiling) schedule for schur_ms, look-ahead schedule for factor_ms.
# Variants built by scripts/build_variant.sh: epi0 (no scatter), epi1 (plain load-sub-store), noload (constant operands), noload_epi0 (both).
# Their FACTORS ARE WRONG by construction (timing only); the in-tree library runs first and last.
tag=${1:-r06_ab_epi}
mkdir -p gpurun_out
bash scripts/ab.sh $tag ab/libsluamd_epi0.so ab/libsluamd_epi1.so ab/libsluamd_noload.so ab/libsluamd_noload_epi0.so 2>&1 | tee gpurun_out/${tag}.txt
for i in 1 2 3 4; do python - <<PY >> gpurun_out/${tag}.txt
import json
try:
    j = json.load(open("gpurun_out/${tag}_lib$i.json")); r = j["roofline"]
    print("lib$i by_configuration:", {k: (round(v["ms"], 1), round(v.get("frac", 0), 3)) for k, v in r.get("by_configuration", {}).items()}, "launches", r["launches"])
except Exception as e: print("lib$i", e)
PY
done
python - <<PY >> gpurun_out/${tag}.txt
import json
for nm in ("tree", "tree2"):
    try:
        j = json.load(open("gpurun_out/${tag}_%s.json" % nm)); r = j["roofline"]
        print(nm, "by_configuration:", {k: (round(v["ms"], 1), round(v.get("frac", 0), 3)) for k, v in r.get("by_configuration", {}).items()}, "launches", r["launches"])
    except Exception as e: print(nm, e)
PY
tail -8 gpurun_out/${tag}.txt
