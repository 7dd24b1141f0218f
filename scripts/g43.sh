mkdir -p gpurun_out
for n in 2000 3000; do
timeout 600 python bench.py --workload zgrid2d --n $n --steps 2 --warmup 1 --no-cpu-baseline --no-scaling-point > gpurun_out/r05_zgrid2d_$n.json 2> gpurun_out/r05_zgrid2d_$n.err
python - <<PY
import json
try:
    j=json.load(open("gpurun_out/r05_zgrid2d_$n.json"))
    print($n, {k:j.get(k) for k in ("value","factor_ms","solve_ms","residual","setup_s")}, "schur frac", j["roofline"]["frac"], "solve frac", j["roofline_solve"]["frac"])
except Exception as e:
    print("failed", e); print(open("gpurun_out/r05_zgrid2d_$n.err").read()[-600:])
PY
done
