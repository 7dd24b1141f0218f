mkdir -p gpurun_out
for w in 256 64 128 256 64; do
  SLUAMD_LUWAVE_WG=$w timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-scaling-point --no-configs4 > gpurun_out/g9_$w.json 2> gpurun_out/g9_$w.err
  python - <<PY
import json
try:
    j=json.load(open("gpurun_out/g9_$w.json"))
    print("luwave wg $w: factor_ms %.2f solve_ms %.2f res %.1e setup %.2f first %.2f" % (j["factor_ms"], j["solve_ms"], j["residual"], j["setup_s"], j["setup_breakdown"]["first_step_s"]), {k: round(v, 3) for k, v in j["setup_breakdown"]["handle_create_phases_s"].items() if v > 0.015})
except Exception as e:
    print("wg $w: failed", e); print(open("gpurun_out/g9_$w.err").read()[-600:])
PY
done
