mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_fuzz.py -q -x --timeout=600 -k "complex or z_ or zg or cg20 or fuzz" > gpurun_out/g13_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/g13_pytest.log)
tail -2 gpurun_out/g13_pytest.log
for z in 0 1 2 4 8 16 32 64 256 0 16; do
  SLUAMD_ZFUSE_MAX_NODES=$z timeout 300 python bench.py --workload zgrid2d --n 1000 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/g13_z.json 2> gpurun_out/g13_z.err
  python - <<PY
import json
try:
    j=json.load(open("gpurun_out/g13_z.json"))
    print("zfuse $z: factor_ms %.2f solve_ms %.3f res %.1e frac %.3f" % (j["factor_ms"], j["solve_ms"], j["residual"], j["roofline_solve"]["frac"]))
except Exception as e:
    print("failed", e); print(open("gpurun_out/g13_z.err").read()[-800:])
PY
done
