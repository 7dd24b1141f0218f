#!/bin/bash
# A/B of environment settings on ONE box: bash scripts/env_ab.sh <tag> "ENV1=a ENV2=b" "ENV3=c" ...   (baseline first and last)
tag=${1:-env}; shift
mkdir -p gpurun_out
run() {
  local name=$1; shift
  env "$@" timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-scaling-point --no-configs4 > gpurun_out/${tag}_$name.json 2> gpurun_out/${tag}_$name.err
  python - <<PY
import json
try:
    j=json.load(open("gpurun_out/${tag}_$name.json"))
    print("%-44s factor_ms %.1f solve_ms %.2f schur_ms %.1f panel_ms %.1f frac %.3f res %.1e" % ("$*", j["factor_ms"], j["solve_ms"], j["roofline"]["schur_ms"], j["roofline"]["panel_ms"], j["roofline"]["frac"], j["residual"]))
except Exception as e:
    print("$* failed", e); print(open("gpurun_out/${tag}_$name.err").read()[-400:])
PY
}
run base A=1
i=0
for e in "$@"; do i=$((i+1)); run v$i $e; done
run base2 A=1
