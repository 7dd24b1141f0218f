mkdir -p gpurun_out
timeout 1200 python bench.py --workload audikw_like --n 68 --steps 2 --warmup 1 --no-cpu-baseline --no-scaling-point --no-configs4 > gpurun_out/r05_audikw_like_68.json 2> gpurun_out/r05_audikw_like_68.err
python - <<'PY'
import json
j=json.load(open("gpurun_out/r05_audikw_like_68.json"))
print({k:j.get(k) for k in ("value","factor_ms","solve_ms","residual","setup_s")}, j["roofline"]["frac"], j["roofline_solve"]["frac"])
print(j["setup_breakdown"])
PY
tail -3 gpurun_out/r05_audikw_like_68.err
