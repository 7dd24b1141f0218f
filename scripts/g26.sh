mkdir -p gpurun_out
run() {
  tag=$1; shift
  env "$@" timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-scaling-point --no-configs4 > gpurun_out/g26_$tag.json 2> gpurun_out/g26_$tag.err
  python - <<PY
import json
try:
    j=json.load(open("gpurun_out/g26_$tag.json"))
    print("$tag: factor_ms %.2f solve_ms %.3f frac %.3f" % (j["factor_ms"], j["solve_ms"], j["roofline_solve"]["frac"]))
except Exception as e:
    print("failed", e); print(open("gpurun_out/g26_$tag.err").read()[-800:])
PY
}
run base SLUAMD_SOLVE_GROUPS=0
run g8 SLUAMD_SOLVE_GROUPS=1
run g8_q8 SLUAMD_SOLVE_GROUPS=1 GPU_MAX_HW_QUEUES=8
run base_q8 SLUAMD_SOLVE_GROUPS=0 GPU_MAX_HW_QUEUES=8
run g2 SLUAMD_SOLVE_GROUPS=1 SLUAMD_SOLVE_GROUP_LEVEL_NODES=2
run g1 SLUAMD_SOLVE_GROUPS=1 SLUAMD_SOLVE_GROUP_LEVEL_NODES=1
run g8_serial SLUAMD_SOLVE_GROUPS=1 SLUAMD_LOOKAHEAD=0
run base_serial SLUAMD_SOLVE_GROUPS=0 SLUAMD_LOOKAHEAD=0
