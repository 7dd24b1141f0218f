# round 5, GPU call 5: fast complex pivot reciprocal -- complex parity, then stats of one-wave vs four-wave LU and the bench A/B
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_fuzz.py tests/test_gpu_grid.py -q -x --timeout=600 -k "complex or z_ or zg or cg20 or fuzz" > gpurun_out/g5_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/g5_pytest.log)
tail -3 gpurun_out/g5_pytest.log
for z in 0 100000; do
  cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ksz
  SLUAMD_ZLU4_MAX_NODES=$z rocprofv3 --kernel-trace --stats -d /tmp/ksz -o run -- python $R/bench.py --workload zgrid2d --n 1000 --steps 3 --warmup 2 --no-cpu-baseline > /tmp/ksz.json 2> /tmp/ksz.err
  cd $R
  db=$(find /tmp/ksz -name "*.db" | head -1)
  python scripts/rocpd_stats.py $db > gpurun_out/g5_zstats_$z.txt 2>&1
  python scripts/timeline.py $db 3 > gpurun_out/g5_ztimeline_$z.txt 2>&1
  head -4 gpurun_out/g5_zstats_$z.txt
done
for z in 0 256 16 100000 0 256; do
  SLUAMD_ZLU4_MAX_NODES=$z timeout 300 python bench.py --workload zgrid2d --n 1000 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/g5_z_$z.json 2> gpurun_out/g5_z_$z.err
  python - <<PY
import json
try:
    j=json.load(open("gpurun_out/g5_z_$z.json"))
    print("zlu4 $z: factor_ms %.2f solve_ms %.2f res %.1e panel_ms %.2f schur_ms %.2f" % (j["factor_ms"], j["solve_ms"], j["residual"], j["roofline"]["panel_ms"], j["roofline"]["schur_ms"]))
except Exception as e:
    print("zlu4 $z: failed", e); print(open("gpurun_out/g5_z_$z.err").read()[-600:])
PY
done
