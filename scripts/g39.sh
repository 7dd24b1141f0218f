for t in 16 32; do
echo "== SLUAMD_PLAN_THREADS=$t"
SLUAMD_PLAN_THREADS=$t SLUAMD_SYMB_TIMING=1 python - <<'PY' 2>&1 | tail -9
import time
from superlu_dist_amd import driver, matgen
N=150
n, rp, ci, v = matgen.poisson3d(N); perm = matgen.nd_perm_grid3d(N,N,N,leaf=64)
for rep in range(2):
    t0=time.perf_counter(); symb = driver.Symbolic(n, rp, ci, perm, relax=64, maxsup=256); t1=time.perf_counter()
    print("symbolic total %.3f" % (t1-t0))
    if rep == 1:
        h = driver.LUHandle.from_symbolic(symb, v); t2=time.perf_counter()
        print("handle %.3f" % (t2-t1), " ".join("%s=%.0f" % (k.split(".")[-1][:12], 1e3*x) for k, x in h.setup_times().items()))
        h.destroy()
    symb.free()
PY
done
