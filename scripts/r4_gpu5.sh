#!/bin/bash
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q -x --timeout=900 > gpurun_out/g5_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/g5_pytest.log)
tail -4 gpurun_out/g5_pytest.log
timeout 900 python bench.py > gpurun_out/g5_bench_default.json 2> gpurun_out/g5_bench_default.err
python - <<PY
import json
j=json.load(open("gpurun_out/g5_bench_default.json"))
print({k:j[k] for k in ("value","ms_per_step","factor_ms","solve_ms","residual")})
print(json.dumps(j["roofline"].get("by_configuration"),indent=0))
print(j["cpu_baseline"].get("grid_2x2x2"))
print(j["configs4"]["factor_ms"], j["configs4"]["solve_ms"])
PY
