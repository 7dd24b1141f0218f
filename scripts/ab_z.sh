#!/bin/bash
# Same-box A/B of the complex16 configuration (configs[4]: 1000^2 grid operator): the in-tree library against other builds.
# usage: bash scripts/ab_z.sh <tag> [lib paths...]
tag=${1:-abz}; shift
mkdir -p gpurun_out
run() {
  local name=$1 lib=$2
  SLUAMD_LIB=$lib timeout 300 python bench.py --workload zgrid2d --n 1000 --steps 5 --warmup 2 --no-cpu-baseline --no-scaling-point --no-configs4 > gpurun_out/${tag}_$name.json 2> gpurun_out/${tag}_$name.err
  python - <<PY
import json
try:
    j=json.load(open("gpurun_out/${tag}_$name.json"))
    print("%-8s factor_ms %.2f [%.2f..%.2f] solve_ms %.3f res %.1e launches %s" % ("$name", j["factor_ms"], j.get("factor_ms_min", 0), j.get("factor_ms_max", 0), j["solve_ms"], j["residual"], j.get("launches_per_factor")))
except Exception as e:
    print("$name failed", e); print(open("gpurun_out/${tag}_$name.err").read()[-600:])
PY
}
run tree ""
i=0
for l in "$@"; do i=$((i+1)); run lib$i $l; done
run tree2 ""
