#!/bin/bash
# One GPU iteration: quick parity subset, reserve-CU sweep of the bench, rocprof kernel table of the default bench.
# usage (on the GPU box, from the repo root): bash scripts/gpu_iter.sh <tag> [reserve values...]
tag=${1:-it}; shift
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_refine.py tests/test_gpu_edge_cases.py tests/test_gpu_grid.py -q --timeout=300 -x > gpurun_out/${tag}_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/${tag}_pytest.log)
grep -E "passed|failed|FAILED|rc " gpurun_out/${tag}_pytest.log | tail -8
for r in "$@"; do
  SLUAMD_RESERVE_CUS=$r timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-scaling-point --no-configs4 > gpurun_out/${tag}_bench_r$r.json 2> gpurun_out/${tag}_bench_r$r.err
  python - <<PY
import json
try:
    j=json.load(open("gpurun_out/${tag}_bench_r$r.json"))
    print("reserve $r: value %.0f factor_ms %.1f solve_ms %.2f schur_ms %.1f panel_ms %.1f frac %.3f res %.1e" % (j["value"], j["factor_ms"], j["solve_ms"], j["roofline"]["schur_ms"], j["roofline"]["panel_ms"], j["roofline"]["frac"], j["residual"]))
except Exception as e:
    print("reserve $r: failed", e); print(open("gpurun_out/${tag}_bench_r$r.err").read()[-800:])
PY
done
