#!/bin/bash
# End-of-round re-check after host-side changes (kernels unchanged since scripts/final_round.sh collected the PMC passes and kernel statistics):
# the whole -m gpu suite and the default bench line on the final tree.
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q -x --timeout=900 > gpurun_out/r05_gputest_final.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05_gputest_final.log)
tail -4 gpurun_out/r05_gputest_final.log
python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' 2>&1 | tail -2
python bench.py > gpurun_out/r05_bench100_final.json 2> gpurun_out/r05_bench100_final.err
python - <<'PY'
import json
j=json.load(open("gpurun_out/r05_bench100_final.json"))
print({k:j.get(k) for k in ("value","factor_ms","solve_ms","residual","setup_s")}, j["roofline"]["frac"], j["roofline"].get("traffic"), j["roofline_solve"]["frac"])
print(j["roofline"]["by_configuration"]["k_schur<64,64,4>"]["atomics"], j["roofline"]["by_configuration"]["k_schur<128,128,8>"]["atomics"])
for k in ("scaling_point","strong_scaling_point"): print(k, j[k]["factor_ms"], j[k]["setup_s"])
print(j["cpu_baseline"]["value"], j["cpu_baseline"]["kind"], j["cpu_baseline"].get("grid_2x2x2",{}).get("value"))
PY
