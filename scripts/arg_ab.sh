#!/bin/bash
# A/B of bench.py ARGUMENTS on ONE box: bash scripts/arg_ab.sh <tag> "--relax 128" "--relax 256" ...   (defaults first and last)
tag=${1:-arg}; shift
mkdir -p gpurun_out
run() {
  local name=$1; shift
  timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-scaling-point $* > gpurun_out/${tag}_$name.json 2> gpurun_out/${tag}_$name.err
  python - <<PY
import json
try:
    j=json.load(open("gpurun_out/${tag}_$name.json"))
    print("%-24s value %.0f step_ms %.1f factor_ms %.1f solve_ms %.2f schur_ms %.1f panel_ms %.1f frac %.3f res %.1e nsupers %d flops %.4e" % ("$*", j["value"], j["ms_per_step"], j["factor_ms"], j["solve_ms"], j["roofline"]["schur_ms"], j["roofline"]["panel_ms"], j["roofline"]["frac"], j["residual"], j["config"]["nsupers"], j["flops_per_step"]))
except Exception as e:
    print("$* failed", e); print(open("gpurun_out/${tag}_$name.err").read()[-400:])
PY
}
run base
i=0
for a in "$@"; do i=$((i+1)); run v$i $a; done
run base2
