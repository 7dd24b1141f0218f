#!/bin/bash
# bench under a list of "NAME=VALUE" environment variants (one run each): bash scripts/env_sweep.sh tag "A=1" "B=2 C=3" ...
tag=$1; shift
mkdir -p gpurun_out
for e in "$@"; do
  env $e timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_sweep.json 2> gpurun_out/${tag}_sweep.err
  python - "$e" <<PY
import json, sys
try:
    j = json.load(open("gpurun_out/${tag}_sweep.json"))
    print("%-40s factor_ms %.1f solve_ms %.2f schur_ms %.1f panel_ms %.1f fused %d res %.1e" % (sys.argv[1], j["factor_ms"], j["solve_ms"], j["roofline"]["schur_ms"], j["roofline"]["panel_ms"], j["fused_level_pairs"], j["residual"]))
except Exception as ex:
    print(sys.argv[1], "failed", ex)
PY
done
