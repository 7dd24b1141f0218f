mkdir -p gpurun_out
run() {
  tag=$1; shift
  env "$@" timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-scaling-point --no-configs4 > gpurun_out/g38_$tag.json 2> gpurun_out/g38_$tag.err
  python - <<PY
import json
j=json.load(open("gpurun_out/g38_$tag.json"))
sb=j["setup_breakdown"]; ph=sb["handle_create_phases_s"]
print("$tag: factor %.1f res %.1e setup %.3f | problem %.3f symbolic %.3f handle %.3f | arena_wait %.0f table_wait %.0f" % (j["factor_ms"], j["residual"], j["setup_s"], sb["problem_generation_ordering_rhs_s"], sb["symbolic_s"], sb["handle_create_s"], 1e3*ph["arena_alloc_zero_wait"], 1e3*ph["upload.block_tile_tables"]))
PY
}
run first A=1
run second A=1
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_grid.py -q -x --timeout=600 > gpurun_out/g38_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/g38_pytest.log)
tail -3 gpurun_out/g38_pytest.log
run third A=1
