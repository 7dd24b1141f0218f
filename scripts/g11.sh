mkdir -p gpurun_out
for i in 1 2 3; do
  timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-scaling-point --no-configs4 > gpurun_out/g11_$i.json 2> gpurun_out/g11_$i.err
  python - <<PY
import json
j=json.load(open("gpurun_out/g11_$i.json"))
b=j["setup_breakdown"]
print("run $i: factor_ms %.2f setup %.3f problem %.3f symbolic %.3f create %.3f" % (j["factor_ms"], j["setup_s"], b["problem_generation_ordering_rhs_s"], b["symbolic_s"], b["handle_create_s"]), {k: round(v, 3) for k, v in b["handle_create_phases_s"].items() if v > 0.012})
PY
done
SLUAMD_SYMB_TIMING=1 python bench.py --steps 1 --warmup 2 --no-cpu-baseline --no-scaling-point --no-configs4 2>&1 >/dev/null | grep dsymbfact
