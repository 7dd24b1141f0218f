#!/bin/bash
mkdir -p gpurun_out
(timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_grid.py tests/test_gpu_fuzz.py tests/test_gpu_bench_config.py -q -x --timeout=900 -k "complex or z_ or cg20 or configs4 or fuzz or tile_records" > gpurun_out/g16_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/g16_pytest.log)
tail -3 gpurun_out/g16_pytest.log
for e in "A=1" "SLUAMD_NO_MERGE_TILES=1" "A=1"; do
  env $e timeout 300 python bench.py --workload zgrid2d --n 1000 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/g16_z.json 2> gpurun_out/g16_z.err
  python - <<PY
import json
j=json.load(open("gpurun_out/g16_z.json"))
print("%-28s factor_ms %.2f solve_ms %.2f schur_ms %.2f panel_ms %.2f frac %.3f res %.1e" % ("$e", j["factor_ms"], j["solve_ms"], j["roofline"]["schur_ms"], j["roofline"]["panel_ms"], j["roofline"]["frac"], j["residual"]))
PY
done
