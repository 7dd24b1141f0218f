# round 5, GPU call 2: split panel solves -- parity subset, then the threshold sweep on one box, then the default-size setup breakdown
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_fuzz.py -q -x --timeout=600 > gpurun_out/g2_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/g2_pytest.log)
tail -3 gpurun_out/g2_pytest.log
for sp in 0 1024 8 128 1000000 0 1024; do
  SLUAMD_PANEL_SPLIT=$sp timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-scaling-point --no-configs4 > gpurun_out/g2_split_$sp.json 2> gpurun_out/g2_split_$sp.err
  python - <<PY
import json
try:
    j=json.load(open("gpurun_out/g2_split_$sp.json"))
    print("split $sp: factor_ms %.2f solve_ms %.2f value %.0f res %.1e launches %d setup %.2f" % (j["factor_ms"], j["solve_ms"], j["value"], j["residual"], j["launches_per_factor"], j["setup_s"]), j["setup_breakdown"])
except Exception as e:
    print("split $sp: failed", e); print(open("gpurun_out/g2_split_$sp.err").read()[-600:])
PY
done
