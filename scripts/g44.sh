mkdir -p gpurun_out
SLUAMD_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 8 --steps 3 --warmup 1 --no-cpu-baseline --scale-n 120 --scale-n2 150 > gpurun_out/r05_bench100_8ranks_one_gpu.json 2> gpurun_out/r05_bench100_8ranks_one_gpu.err
python - <<'PY'
import json
try:
    j=json.load(open("gpurun_out/r05_bench100_8ranks_one_gpu.json"))
    print({k:j.get(k) for k in ("n_gpus","value","factor_ms","solve_ms","residual","setup_s","scaling")})
    print("phases", j.get("phases"))
    for k in ("scaling_point","strong_scaling_point"):
        sp=j[k]; print(k, sp.get("n"), sp.get("factor_ms"), sp.get("solve_ms"), sp.get("residual"), sp.get("phases"))
    print(j["config"]["parallelism"][:200])
except Exception as e:
    print("failed", e); print(open("gpurun_out/r05_bench100_8ranks_one_gpu.err").read()[-1500:])
PY
