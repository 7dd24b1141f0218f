mkdir -p gpurun_out
run() {
  tag=$1; shift
  env "$@" timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-scaling-point --no-configs4 > gpurun_out/g31_$tag.json 2> gpurun_out/g31_$tag.err
  python - <<PY
import json
j=json.load(open("gpurun_out/g31_$tag.json"))
print("$tag: factor_ms %.2f solve_ms %.3f setup %.3f" % (j["factor_ms"], j["solve_ms"], j["setup_s"]))
PY
}
run async1 A=1
run sync1 SLUAMD_SYNC_TABLE_UPLOAD=1
run async2 A=1
run sync2 SLUAMD_SYNC_TABLE_UPLOAD=1
run async3 A=1
run sync3 SLUAMD_SYNC_TABLE_UPLOAD=1
