mkdir -p gpurun_out
for i in 1 2; do
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-scaling-point --no-configs4 > gpurun_out/g28_$i.json 2> gpurun_out/g28_$i.err
python - <<PY
import json
j=json.load(open("gpurun_out/g28_$i.json"))
sb=j["setup_breakdown"]
print("factor_ms %.2f solve_ms %.3f res %.1e setup %.3f | problem %.3f symbolic %.3f handle %.3f first_step %.3f" % (j["factor_ms"], j["solve_ms"], j["residual"], j["setup_s"], sb["problem_generation_ordering_rhs_s"], sb["symbolic_s"], sb["handle_create_s"], sb["first_step_s"]))
print(" ".join("%s=%.0f" % (k, 1e3*v) for k, v in sb["handle_create_phases_s"].items()))
PY
done
SLUAMD_SYMB_TIMING=1 python - <<'PY' 2>&1 | tail -8
import time
from superlu_dist_amd import driver, matgen
N=100
n, rp, ci, v = matgen.poisson3d(N); perm = matgen.nd_perm_grid3d(N,N,N,leaf=64)
for rep in range(2):
    t0=time.perf_counter(); symb = driver.Symbolic(n, rp, ci, perm, relax=32, maxsup=256); print("symbolic total %.3f" % (time.perf_counter()-t0)); symb.free()
PY
