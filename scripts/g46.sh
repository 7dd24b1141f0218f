for cut in default 100000 26000 13000 6000; do
echo "== cut $cut"
if [ $cut = default ]; then unset SLUAMD_SYMB_CUT; else export SLUAMD_SYMB_CUT=$cut; fi
SLUAMD_SYMB_TIMING=1 python - <<'PY' 2>&1 | grep -E "structure|total" | tail -4
import time
from superlu_dist_amd import driver, matgen
N=150
n, rp, ci, v = matgen.poisson3d(N); perm = matgen.nd_perm_grid3d(N,N,N,leaf=64)
for rep in range(2):
    t0=time.perf_counter(); symb = driver.Symbolic(n, rp, ci, perm, relax=64, maxsup=256); print("N=%d symbolic total %.3f" % (N, time.perf_counter()-t0)); symb.free()
PY
done
