mkdir -p gpurun_out
for t in 16 8 4 1; do
echo "== SLUAMD_PLAN_THREADS=$t"
SLUAMD_PLAN_THREADS=$t SLUAMD_SYMB_TIMING=1 python - <<'PY' 2>&1 | grep -v "^$" | tail -14
import time, numpy as np
from superlu_dist_amd import driver, matgen
N=100
t0=time.perf_counter(); n, rp, ci, v = matgen.poisson3d(N); t1=time.perf_counter()
perm = matgen.nd_perm_grid3d(N,N,N,leaf=64); t2=time.perf_counter()
print("gen %.3f perm %.3f" % (t1-t0, t2-t1))
for rep in range(2):
    t0=time.perf_counter()
    symb = driver.Symbolic(n, rp, ci, perm, relax=32, maxsup=256)
    t1=time.perf_counter()
    print("symbolic total %.3f" % (t1-t0))
    symb.free()
PY
done
