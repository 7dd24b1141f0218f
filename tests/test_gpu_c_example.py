"""The C ABI used from plain C (examples/pddrive3d_amd.c): symbolic -> device distribution -> pdgstrf3d -> pdgstrs3d ->
pdgsrfs3d without any Python in the loop; also exercises the triplet-file reader path on a generated .dat."""
import os, subprocess
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "pddrive3d_amd")


def _build():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)


def test_c_driver_poisson():
    _build()
    r = subprocess.run([EXE, "12"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "||X-Xtrue||/||X||" in r.stdout


def test_c_driver_triplet_file(tmp_path):
    _build()
    from superlu_dist_amd import matgen
    n, rp, ci, v = matgen.random_unsym(300, 0.02, 3)
    path = str(tmp_path / "m.dat")
    matgen.write_triplet_dat(path, n, rp, ci, v)
    r = subprocess.run([EXE, path], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
