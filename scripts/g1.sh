mkdir -p gpurun_out
ls -la /opt/conda/lib/libmkl_rt.so oracle/_ref_mkl/slu_ref_dump > gpurun_out/g1_mkl.txt 2>&1
nproc >> gpurun_out/g1_mkl.txt
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_dropin.py -q -x --timeout=600 -k "not at_scale" > gpurun_out/g1_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/g1_pytest.log)
tail -3 gpurun_out/g1_pytest.log
(time python bench.py > gpurun_out/g1_bench.json 2> gpurun_out/g1_bench.err) 2> gpurun_out/g1_bench.time
tail -3 gpurun_out/g1_bench.time
python - <<'PY'
import json
j=json.load(open("gpurun_out/g1_bench.json"))
print("value %.0f factor_ms %.1f solve_ms %.2f frac %.3f setup %.2f" % (j["value"], j["factor_ms"], j["solve_ms"], j["roofline"]["frac"], j["setup_s"]))
c=j["cpu_baseline"]; print({k:c.get(k) for k in ("kind","value","cores","sample","blas","mkl_threading_layer","reference_mkl_unavailable")}); print(c.get("thread_sweep")); print(c.get("grid_2x2x2")); print((c.get("reference_cblas") or {}).get("value"))
PY
