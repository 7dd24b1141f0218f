#!/bin/bash
# round 4, GPU call 1: parity subset for the pipelined wave LU, A/B of the tail options on one box, then the full default bench line
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_bench_config.py tests/test_gpu_dropin.py -q -x --timeout=600 -k "not at_scale" > gpurun_out/g1_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/g1_pytest.log)
tail -4 gpurun_out/g1_pytest.log
run() {
  local name=$1; shift
  env "$@" timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-scaling-point --no-configs4 > gpurun_out/g1_$name.json 2> gpurun_out/g1_$name.err
  python - <<PY
import json
try:
    j=json.load(open("gpurun_out/g1_$name.json"))
    print("%-60s factor_ms %.1f solve_ms %.2f schur_ms %.1f panel_ms %.1f frac %.3f res %.1e" % ("$*", j["factor_ms"], j["solve_ms"], j["roofline"]["schur_ms"], j["roofline"]["panel_ms"], j["roofline"]["frac"], j["residual"]))
except Exception as e:
    print("$* failed", e); print(open("gpurun_out/g1_$name.err").read()[-400:])
PY
}
run r3lib SLUAMD_LIB=$PWD/ab/libsluamd_r3.so
run tree A=1
run trsm40 SLUAMD_TRSM_TAIL=40
run diag40 SLUAMD_DIAG_TAIL=40
run both40 SLUAMD_TRSM_TAIL=40 SLUAMD_DIAG_TAIL=40
run both70 SLUAMD_TRSM_TAIL=70 SLUAMD_DIAG_TAIL=70
run diag200 SLUAMD_DIAG_TAIL=200
run tree2 A=1
timeout 600 python bench.py > gpurun_out/g1_bench_default.json 2> gpurun_out/g1_bench_default.err
tail -c 1500 gpurun_out/g1_bench_default.json
