#!/bin/bash
# complex16 configuration under a list of "NAME=VALUE" environment variants: bash scripts/env_sweep_z.sh tag "A=1" "B=2 C=3" ...
tag=$1; shift
mkdir -p gpurun_out
for e in "$@"; do
  env $e timeout 300 python bench.py --workload zgrid2d --n 1000 --steps 5 --warmup 2 --no-cpu-baseline --no-scaling-point --no-configs4 > gpurun_out/${tag}_sweep.json 2> gpurun_out/${tag}_sweep.err
  python - "$e" <<PY
import json, sys
try:
    j = json.load(open("gpurun_out/${tag}_sweep.json"))
    print("%-40s factor_ms %.2f solve_ms %.3f res %.1e launches %s" % (sys.argv[1], j["factor_ms"], j["solve_ms"], j["residual"], j.get("launches_per_factor")))
except Exception as ex:
    print(sys.argv[1], "failed", ex)
PY
done
