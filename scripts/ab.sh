#!/bin/bash
# A/B timing on ONE box (box-to-box variation is 2-3 %): bench with the in-tree library and with other builds of it.
# usage: bash scripts/ab.sh <tag> [lib paths...]   (the in-tree build always runs first and last)
tag=${1:-ab}; shift
mkdir -p gpurun_out
run() {
  local name=$1 lib=$2
  SLUAMD_LIB=$lib timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-scaling-point --no-configs4 > gpurun_out/${tag}_$name.json 2> gpurun_out/${tag}_$name.err
  python - <<PY
import json
try:
    j=json.load(open("gpurun_out/${tag}_$name.json"))
    print("%-12s value %.0f factor_ms %.1f solve_ms %.2f schur_ms %.1f panel_ms %.1f frac %.3f res %.1e" % ("$name", j["value"], j["factor_ms"], j["solve_ms"], j["roofline"]["schur_ms"], j["roofline"]["panel_ms"], j["roofline"]["frac"], j["residual"]))
except Exception as e:
    print("$name failed", e); print(open("gpurun_out/${tag}_$name.err").read()[-600:])
PY
}
run tree ""
i=0
for l in "$@"; do i=$((i+1)); run lib$i $l; done
run tree2 ""
