#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz by running the REAL reference here.

Needs /root/reference and oracle/_ref/slu_ref_dump (`make -C oracle/ref`); runs only in the build
container.  The .npz files are DATA: hot-path inputs (L/U store before pdgstrf3d, permuted right-hand
sides before pdgstrs3d) and the reference's outputs (store after pdgstrf3d, solution after pdgstrs3d,
final x / berr of pdgssvx3d), recorded by oracle/ref/slu_ref_dump.c.  No reference source is stored.

    python tests/golden/make_golden.py [case ...]
"""
import os, subprocess, sys, tempfile
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from superlu_dist_amd import matgen  # noqa: E402
from slud import read_slud  # noqa: E402

DUMP = os.path.join(ROOT, "oracle", "_ref", "slu_ref_dump")
ZDUMP = os.path.join(ROOT, "oracle", "_ref", "slu_ref_zdump")
MKL_DUMP = os.path.join(ROOT, "oracle", "_ref_mkl", "slu_ref_dump")     # the LAPACK build (make -C oracle/ref ref_mkl): DiagInv = YES only
REF_EX = "/root/reference/EXAMPLE"

# name -> dict(matrix=..., grid=(r,c,d), flags=[...], nd=bool)
CASES = {
    # reference defaults: Equil=YES, RowPerm=LargeDiag_MC64, ColPerm=MMD_AT_PLUS_A, IterRefine=DOUBLE
    "g20_1x1x1": dict(matrix=("file", f"{REF_EX}/g20.rua"), grid=(1, 1, 1), flags=[]),
    "g20_1x1x1_nrhs3": dict(matrix=("file", f"{REF_EX}/g20.rua"), grid=(1, 1, 1), flags=["-s", "3"]),
    "g20_1x1x1_legacy": dict(matrix=("file", f"{REF_EX}/g20.rua"), grid=(1, 1, 1), flags=[], env={"NEW3DSOLVE": "0"}),
    "g20_1x1x2": dict(matrix=("file", f"{REF_EX}/g20.rua"), grid=(1, 1, 2), flags=[]),
    "g20_2x2x2": dict(matrix=("file", f"{REF_EX}/g20.rua"), grid=(2, 2, 2), flags=[]),
    "g20_2x1x1": dict(matrix=("file", f"{REF_EX}/g20.rua"), grid=(2, 1, 1), flags=[]),
    # 7-pt Poisson with our geometric ND perm_c (MY_PERMC), no equil / rowperm / refinement
    "poisson8_nd": dict(matrix=("poisson", 8), grid=(1, 1, 1), nd=16, flags=["-e", "0", "-p", "0", "-i", "0"]),
    "poisson10_nd": dict(matrix=("poisson", 10), grid=(1, 1, 1), nd=27, flags=["-e", "0", "-p", "0", "-i", "0"]),
    "poisson8_nd_1x1x2": dict(matrix=("poisson", 8), grid=(1, 1, 2), nd=16, flags=["-e", "0", "-p", "0", "-i", "0"]),
    # unsymmetric pattern -> ragged U skyline, MC64 + MMD defaults
    "unsym300": dict(matrix=("unsym", 300, 0.02, 7), grid=(1, 1, 1), flags=[]),
    # tiny-pivot replacement exercised (ReplaceTinyPivot=YES)
    "unsym120_tiny": dict(matrix=("unsym", 120, 0.05, 3), grid=(1, 1, 1), flags=["-T", "1"]),
    # iterative refinement (pdgsrfs3d, IterRefine=SLU_DOUBLE) on the un-equilibrated, un-row-permuted system
    "poisson8_nd_refine": dict(matrix=("poisson", 8), grid=(1, 1, 1), nd=16, flags=["-e", "0", "-p", "0", "-i", "2"]),
    "weakdiag150_refine": dict(matrix=("weakdiag", 150, 0.04, 11), grid=(1, 1, 1), flags=["-e", "0", "-p", "0", "-i", "2", "-T", "1"]),
    "weakdiag150_refine_nrhs2": dict(matrix=("weakdiag", 150, 0.04, 11), grid=(1, 1, 1), flags=["-e", "0", "-p", "0", "-i", "2", "-T", "1", "-s", "2"]),
    # DiagInv = YES (pdCompute_Diag_Inv, pdgstrs.c:842-959: dtrtri on the factored diagonal blocks; needs LAPACK -> the MKL build of the reference): Linv / Uinv of
    # every diagonal block recorded beside the usual hot-path records -- the direct fixture of the library's k_full_inv / k_full_inv64
    "poisson10_nd_diaginv": dict(matrix=("poisson", 10), grid=(1, 1, 1), nd=27, flags=["-e", "0", "-p", "0", "-i", "0", "-D", "1"], mkl=True),
    "unsym300_diaginv": dict(matrix=("unsym", 300, 0.02, 7), grid=(1, 1, 1), flags=["-D", "1"], mkl=True),
    # ---- complex16 (pzgssvx3d / pzgstrf3d / pzgstrs3d): BASELINE.json config 5 family ----
    "z_cg20_1x1x1": dict(matrix=("file", f"{REF_EX}/cg20.cua"), grid=(1, 1, 1), flags=[], z=True),
    "z_cg20_1x1x1_nrhs2": dict(matrix=("file", f"{REF_EX}/cg20.cua"), grid=(1, 1, 1), flags=["-s", "2"], z=True),
    "z_poisson8_nd": dict(matrix=("zpoisson", 8), grid=(1, 1, 1), nd=16, flags=["-e", "0", "-p", "0", "-i", "0"], z=True),
    "z_unsym200": dict(matrix=("zunsym", 200, 0.03, 9), grid=(1, 1, 1), flags=[], z=True),
    "z_grid24_nd": dict(matrix=("zgrid2d", 24), grid=(1, 1, 1), nd=16, flags=["-e", "0", "-p", "0", "-i", "0"], z=True),
    # complex16 on Z layers (1 x 1 x 2): per-rank records of pzgstrf3d / pzgstrs3d
    "z_cg20_1x1x2": dict(matrix=("file", f"{REF_EX}/cg20.cua"), grid=(1, 1, 2), flags=[], z=True),
    "z_poisson8_nd_1x1x2": dict(matrix=("zpoisson", 8), grid=(1, 1, 2), nd=16, flags=["-e", "0", "-p", "0", "-i", "0"], z=True),
    # complex16 on XY layers (round 3): per-rank records on 2 x 1 x 1, 1 x 2 x 1 and 2 x 2 x 2 grids
    "z_cg20_2x1x1": dict(matrix=("file", f"{REF_EX}/cg20.cua"), grid=(2, 1, 1), flags=[], z=True),
    "z_cg20_1x2x1": dict(matrix=("file", f"{REF_EX}/cg20.cua"), grid=(1, 2, 1), flags=[], z=True),
    "z_cg20_2x2x2": dict(matrix=("file", f"{REF_EX}/cg20.cua"), grid=(2, 2, 2), flags=[], z=True),
}


def build_case(name, spec, tmp):
    kind = spec["matrix"][0]
    flags = list(spec.get("flags", []))
    if kind == "file":
        mpath = spec["matrix"][1]
    else:
        if kind == "zgrid2d":
            N = spec["matrix"][1]
            n, rp, ci, v = matgen.poisson3d(0, N, N, 1)
        elif kind in ("poisson", "zpoisson"):
            N = spec["matrix"][1]
            n, rp, ci, v = matgen.poisson3d(N)
        elif kind == "weakdiag":
            _, nn, dens, seed = spec["matrix"]
            n, rp, ci, v = matgen.random_unsym(nn, dens, seed, diag_scale=0.008)
        else:
            _, nn, dens, seed = spec["matrix"]
            n, rp, ci, v = matgen.random_unsym(nn, dens, seed)
        if kind.startswith("z"):
            v = matgen.complex_shift(v, rp, ci, seed=4 if kind == "zgrid2d" else 1)
        mpath = os.path.join(tmp, name + ".dat")
        matgen.write_triplet_dat(mpath, n, rp, ci, v)
        if spec.get("nd"):
            N = spec["matrix"][1]
            perm = matgen.nd_perm_grid3d(N, N, 1 if kind == "zgrid2d" else N, leaf=spec["nd"])
            ppath = os.path.join(tmp, name + ".perm")
            np.savetxt(ppath, perm, fmt="%d")
            flags += ["-P", ppath]
    r, c, d = spec["grid"]
    nproc = r * c * d
    outp = os.path.join(tmp, name)
    cmd = ["/opt/conda/bin/mpiexec", "-n", str(nproc), ZDUMP if spec.get("z") else MKL_DUMP if spec.get("mkl") else DUMP, "-r", str(r), "-c", str(c), "-d", str(d),
           "-Q", "1", "-o", outp] + flags + [mpath]
    env = dict(os.environ, OMP_NUM_THREADS="1", LD_LIBRARY_PATH="/opt/conda/lib", MKL_THREADING_LAYER="SEQUENTIAL")
    env.update(spec.get("env", {}))
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    if res.returncode != 0:
        print(res.stdout[-2000:], res.stderr[-2000:])
        raise RuntimeError(f"reference run failed for {name}")
    arrays = {"nranks": np.array([nproc]), "grid": np.array([r, c, d])}
    for rank in range(nproc):
        rec = read_slud(f"{outp}.r{rank}.slud")
        for k, a in rec.items():
            arrays[f"r{rank}__{k}"] = a
    out = os.path.join(HERE, name + ".npz")
    np.savez_compressed(out, **arrays)
    print(f"{name}: {os.path.getsize(out) / 1024:.0f} KiB  n={int(arrays['r0__n'][0])} nsupers={int(arrays['r0__nsupers'][0])}")


def main():
    names = sys.argv[1:] or list(CASES)
    with tempfile.TemporaryDirectory() as tmp:
        for nm in names:
            build_case(nm, CASES[nm], tmp)


if __name__ == "__main__":
    main()
