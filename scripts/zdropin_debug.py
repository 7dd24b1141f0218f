import os, sys, subprocess, tempfile, numpy as np
sys.path.insert(0, "/root/repo")
from superlu_dist_amd import matgen
tmp = tempfile.mkdtemp()
n, rp, ci, v = matgen.poisson3d(0, 24, 24, 1)
perm = matgen.nd_perm_grid3d(24, 24, 1, leaf=16)
np.savetxt(tmp + "/a.perm", perm, fmt="%d")
v = matgen.complex_shift(v, rp, ci, seed=4)
matgen.write_triplet_dat(tmp + "/a.dat", n, rp, ci, v)
env = dict(os.environ, OMP_NUM_THREADS="1", SLUAMD_BIND_DEBUG="1"); env.pop("LD_LIBRARY_PATH", None)
for flags in (["-e", "0", "-p", "0", "-i", "0", "-P", tmp + "/a.perm"], ["-e", "0", "-p", "0", "-P", tmp + "/a.perm"], ["-i", "0"]):
    r = subprocess.run(["/root/repo/oracle/_ref/slu_ref_zamd", "-Q", "1", "-o", "none"] + flags + [tmp + "/a.dat"], env=env, capture_output=True, text=True)
    print(flags[:6], "rc", r.returncode, r.stdout[-200:].replace("\n", " | "), "ERR:", r.stderr[-400:])
