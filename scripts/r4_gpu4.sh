#!/bin/bash
mkdir -p gpurun_out
rocm-smi --showclocks 2>&1 | head -20 > gpurun_out/g4_clocks.txt
rocm-smi --showperflevel 2>&1 | tail -5 >> gpurun_out/g4_clocks.txt
./scripts/ubench/wave_lu > gpurun_out/g4_wave_lu.txt 2>&1; cat gpurun_out/g4_wave_lu.txt
run() {
  local name=$1; shift
  env "$@" timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-scaling-point --no-configs4 > gpurun_out/g4_$name.json 2> gpurun_out/g4_$name.err
  python - <<PY
import json
try:
    j=json.load(open("gpurun_out/g4_$name.json"))
    print("%-60s factor_ms %.1f solve_ms %.2f schur_ms %.1f panel_ms %.1f frac %.3f res %.1e" % ("$*", j["factor_ms"], j["solve_ms"], j["roofline"]["schur_ms"], j["roofline"]["panel_ms"], j["roofline"]["frac"], j["residual"]))
except Exception as e:
    print("$* failed", e); print(open("gpurun_out/g4_$name.err").read()[-400:])
PY
}
run base A=1
rocm-smi --setperflevel high > gpurun_out/g4_setperf.txt 2>&1; tail -3 gpurun_out/g4_setperf.txt
rocm-smi --showclocks 2>&1 | head -12 >> gpurun_out/g4_clocks.txt
./scripts/ubench/wave_lu 2>&1 | head -4
run perfhigh A=1
rocm-smi --setperflevel auto > /dev/null 2>&1
run base2 A=1
