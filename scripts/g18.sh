mkdir -p gpurun_out
(time python bench.py > gpurun_out/g18_bench.json 2> gpurun_out/g18_bench.err) 2> gpurun_out/g18_time.txt
tail -3 gpurun_out/g18_time.txt
python - <<'PY'
import json
j=json.load(open("gpurun_out/g18_bench.json"))
print("value %.0f factor %.2f solve %.2f frac %.3f setup %.2f solve_frac %.3f traffic %s stale %s" % (j["value"], j["factor_ms"], j["solve_ms"], j["roofline"]["frac"], j["setup_s"], j["roofline_solve"]["frac"], j["roofline"]["traffic"], j["roofline"]["traffic_stale"]))
for k in ("scaling_point","strong_scaling_point"):
    s=j[k]; print(k, {q: s.get(q) for q in ("value","factor_ms","solve_ms","setup_s","solve_hbm_frac","error")})
c=j["configs4"]; print("configs4", {q: c.get(q) for q in ("value","factor_ms","solve_ms","residual","error")}, c["roofline"]["frac"], c["roofline_solve"]["frac"])
c=j["cpu_baseline"]; print({q: c.get(q) for q in ("kind","value","cores","sample")}, (c.get("grid_2x2x2") or {}).get("value"), (c.get("reference_cblas") or {}).get("value"))
PY
