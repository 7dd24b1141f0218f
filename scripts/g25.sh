mkdir -p gpurun_out
(SLUAMD_SOLVE_GROUPS=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_fuzz.py tests/test_gpu_refine.py -q -x --timeout=600 > gpurun_out/g25_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/g25_pytest.log)
tail -3 gpurun_out/g25_pytest.log
for g in 0 1 0 1; do
  SLUAMD_PLAN_DEBUG=1 SLUAMD_SOLVE_GROUPS=$g timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-scaling-point --no-configs4 > gpurun_out/g25_$g.json 2> gpurun_out/g25_$g.err
  python - <<PY
import json
try:
    j=json.load(open("gpurun_out/g25_$g.json"))
    print("groups $g: factor_ms %.2f solve_ms %.3f value %.0f res %.1e frac %.3f setup %.2f launches %s" % (j["factor_ms"], j["solve_ms"], j["value"], j["residual"], j["roofline_solve"]["frac"], j["setup_s"], j.get("stats",{}).get("solve_launches")))
except Exception as e:
    print("failed", e); print(open("gpurun_out/g25_$g.err").read()[-800:])
PY
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pg; SLUAMD_SOLVE_GROUPS=1 timeout 300 rocprofv3 --kernel-trace -d /tmp/pg -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-scaling-point --no-configs4 > /tmp/pg.log 2>&1
cd $GRAFT_REPO_ROOT
db=$(find /tmp/pg -name "*.db" | head -1)
python scripts/rocpd_stats.py $db > gpurun_out/g25_kernel_stats.txt 2>&1
grep -E "gemm_batched|grp_gather|k_sweep|k_fwd_update|k_bwd_update" gpurun_out/g25_kernel_stats.txt
python scripts/solve_timeline.py $db > gpurun_out/g25_solve_timeline.txt 2>&1; tail -5 gpurun_out/g25_solve_timeline.txt
