mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_fuzz.py tests/test_gpu_grid.py tests/test_gpu_dropin.py -q -x --timeout=600 -k "complex or z_ or zg or cg20 or fuzz" > gpurun_out/g12_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/g12_pytest.log)
tail -3 gpurun_out/g12_pytest.log
for e in "" "SLUAMD_Z_NO_FUSE=1" ""; do
  env $e timeout 300 python bench.py --workload zgrid2d --n 1000 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/g12_z.json 2> gpurun_out/g12_z.err
  python - <<PY
import json
try:
    j=json.load(open("gpurun_out/g12_z.json"))
    print("[$e] factor_ms %.2f solve_ms %.3f res %.1e err %.1e solve frac %.3f" % (j["factor_ms"], j["solve_ms"], j["residual"], j["max_abs_err_vs_xtrue"], j["roofline_solve"]["frac"]))
except Exception as e:
    print("failed", e); print(open("gpurun_out/g12_z.err").read()[-800:])
PY
done
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ksz
rocprofv3 --kernel-trace --stats -d /tmp/ksz -o run -- python $R/bench.py --workload zgrid2d --n 1000 --steps 3 --warmup 2 --no-cpu-baseline > /tmp/ksz.json 2> /tmp/ksz.err
cd $R
python scripts/rocpd_stats.py $(find /tmp/ksz -name "*.db" | head -1) > gpurun_out/g12_zstats.txt 2>&1
head -9 gpurun_out/g12_zstats.txt
