mkdir -p gpurun_out
(SLUAMD_SOLVE_GROUPS=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_fuzz.py tests/test_gpu_refine.py -q -x --timeout=600 > gpurun_out/g23_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/g23_pytest.log)
tail -3 gpurun_out/g23_pytest.log
for g in 0 1 0 1; do
  SLUAMD_PLAN_DEBUG=1 SLUAMD_SOLVE_GROUPS=$g timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-scaling-point --no-configs4 > gpurun_out/g23_$g.json 2> gpurun_out/g23_$g.err
  grep "merged chain" gpurun_out/g23_$g.err | tail -1
  python - <<PY
import json
try:
    j=json.load(open("gpurun_out/g23_$g.json"))
    print("groups $g: factor_ms %.2f solve_ms %.3f value %.0f res %.1e frac %.3f setup %.2f" % (j["factor_ms"], j["solve_ms"], j["value"], j["residual"], j["roofline_solve"]["frac"], j["setup_s"]))
except Exception as e:
    print("failed", e); print(open("gpurun_out/g23_$g.err").read()[-800:])
PY
done
