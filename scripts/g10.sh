mkdir -p gpurun_out
(time python bench.py --no-cpu-baseline --no-configs4 > gpurun_out/g10_bench.json 2> gpurun_out/g10_bench.err) 2> gpurun_out/g10_time.txt
tail -3 gpurun_out/g10_time.txt
python - <<'PY'
import json
j=json.load(open("gpurun_out/g10_bench.json"))
print("100: factor %.2f solve %.2f setup %.2f" % (j["factor_ms"], j["solve_ms"], j["setup_s"]), j["setup_breakdown"])
for k in ("scaling_point","strong_scaling_point"):
    s=j.get(k,{})
    if "error" in s: print(k, s); continue
    print(k, "factor %.1f solve %.2f value %.0f setup %.2f hbm %.3f" % (s["factor_ms"], s["solve_ms"], s["value"], s["setup_s"], s.get("solve_hbm_frac",0)), s["setup_breakdown"])
PY
