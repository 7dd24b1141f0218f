mkdir -p gpurun_out
run() {
  tag=$1; shift
  env "$@" timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs4 --scale-n2 120 > gpurun_out/g35_$tag.json 2> gpurun_out/g35_$tag.err
  python - <<PY
import json
j=json.load(open("gpurun_out/g35_$tag.json"))
for key in ("scaling_point","strong_scaling_point"):
    sb=j[key]["setup_breakdown"]; ph=sb["handle_create_phases_s"]
    print("$tag %s n=%d: setup %.2f handle %.2f arena_wait %.3f table_wait %.3f first_step %.2f" % (key[:6], round(j[key]["n"]**(1/3)), j[key]["setup_s"], sb["handle_create_s"], ph["arena_alloc_zero_wait"], ph["upload.block_tile_tables"], sb["first_step_s"]))
PY
}
run default A=1
run nothp SLUAMD_NO_THP=1
run synctab SLUAMD_SYNC_TABLE_UPLOAD=1
run default2 A=1
