python - <<'PY'
import time, os, subprocess, sys
from superlu_dist_amd import matgen, driver
import numpy as np
t0=time.perf_counter(); n,rp,ci,v=matgen.elasticity3d_like(68, drop=0.05, seed=1); print("gen %.2f s, n %d nnz %d" % (time.perf_counter()-t0, n, len(ci)))
np.savez("/tmp/el68.npz", rp=rp, ci=ci)
for rep in range(2):
    t1=time.perf_counter(); perm=driver.order_nd(n,rp,ci,leaf=64); print("order_nd (default threads) %.2f s" % (time.perf_counter()-t1))
PY
for t in 1 4; do SLUAMD_PLAN_THREADS=$t python - <<'PY'
import time, os, numpy as np
from superlu_dist_amd import driver
d=np.load("/tmp/el68.npz"); rp, ci = d["rp"], d["ci"]; n=len(rp)-1
t1=time.perf_counter(); perm=driver.order_nd(n,rp,ci,leaf=64); print("order_nd SLUAMD_PLAN_THREADS=%s %.2f s" % (os.environ["SLUAMD_PLAN_THREADS"], time.perf_counter()-t1))
PY
done
