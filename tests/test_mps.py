"""CPU: create_model_from_mps against hand-computed models (behaviour list in
hpr-lp-c_amd/csrc/mps_reader.cpp, taken from reference src/mps_reader.cpp)."""
import gzip
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import ROOT, hprlp

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
INF = np.inf


def test_small_lp():
    m = hprlp.Model.from_mps(os.path.join(DATA, "lp_small.mps"))
    assert (m.m, m.n, m.obj_constant) == (2, 2, 0.0)
    rp, ci, v = m.csr()
    assert list(rp) == [0, 2, 4] and list(ci) == [0, 1, 0, 1] and list(v) == [1, 2, 3, 1]
    vec = m.vectors()
    assert list(vec["AL"]) == [-INF, -INF] and list(vec["AU"]) == [10, 12]
    assert list(vec["l"]) == [0, 0] and list(vec["u"]) == [INF, INF] and list(vec["c"]) == [-3, -5]
    m.free()


def test_every_card_type(capfd):
    m = hprlp.Model.from_mps(os.path.join(DATA, "lp_features.mps"))
    err = capfd.readouterr().err
    assert "rim objective row RIMOBJ" in err and "rim RHS OTHER" in err and "rim bound OTHERSET" in err
    assert (m.m, m.n) == (5, 8)
    assert m.obj_constant == 7.0                      # RHS on the objective row: constant = -(-7)
    rp, ci, v = m.csr()
    rows = {  # EQP, EQN, LE1, GE1, GE2 ; columns A..H = 0..7 ; duplicate (LE1,A) entries are summed
        0: {0: 1.0, 3: 3.0}, 1: {1: -1.0, 6: 2.0}, 2: {0: 2.5, 4: 1.0}, 3: {1: 4.0, 5: 1.0}, 4: {2: 1.0, 3: -1.0, 7: 5.0}}
    for i in range(5):
        got = dict(zip(ci[rp[i]:rp[i + 1]].tolist(), v[rp[i]:rp[i + 1]].tolist()))
        assert got == rows[i], i
        assert list(ci[rp[i]:rp[i + 1]]) == sorted(ci[rp[i]:rp[i + 1]])
    vec = m.vectors()
    # RANGES: E with R=+2 -> [4,6]; E with R=-3 -> [-5,-2]; L: [10-4,10]; G: [1,1+6]; GE2 keeps default RHS 0
    assert list(vec["AL"]) == [4, -5, 6, 1, 0] and list(vec["AU"]) == [6, -2, 10, 7, INF]
    #            A     B    C(marked) D(UP<0)  E    F    G    H(BV)
    assert list(vec["l"]) == [-INF, -INF, 0, -INF, 0, 2, 3.5, 0]
    assert list(vec["u"]) == [INF, INF, 1, -1, 5, INF, 3.5, 1]
    assert list(vec["c"]) == [1.5, 0, -2, 0, 0, 0, 0, 0]
    m.free()


def test_gzip_input(tmp_path):
    gz = tmp_path / "lp_small.mps.gz"
    with open(os.path.join(DATA, "lp_small.mps"), "rb") as f, gzip.open(gz, "wb") as g:
        shutil.copyfileobj(f, g)
    m = hprlp.Model.from_mps(gz)
    assert (m.m, m.n) == (2, 2) and list(m.vectors()["AU"]) == [10, 12]
    m.free()


def test_bad_files(tmp_path, capfd):
    L = hprlp.lib()
    assert not L.create_model_from_mps(None)
    assert not L.create_model_from_mps(str(tmp_path / "missing.mps").encode())
    p = tmp_path / "two_rows_sections.mps"
    p.write_text("NAME X\nROWS\n N OBJ\nROWS\n L R1\nENDATA\n")
    assert not L.create_model_from_mps(str(p).encode())
    p = tmp_path / "cols_before_rows.mps"
    p.write_text("NAME X\nCOLUMNS\n X OBJ 1\nENDATA\n")
    assert not L.create_model_from_mps(str(p).encode())
    p = tmp_path / "empty.mps"
    p.write_text("NAME X\nROWS\n N OBJ\nCOLUMNS\nENDATA\n")
    assert not L.create_model_from_mps(str(p).encode())
    assert "Error" in capfd.readouterr().err


def test_cli_driver_plumbing():
    """BASELINE config 1 plumbing: the driver parses its flags and reads the file; without a GPU the
    solve reports ERROR (exit code 2) instead of falling back to a CPU path."""
    exe = os.path.join(ROOT, "bin", "solve_mps_file")
    assert os.path.exists(exe), "build with `make`"
    r = subprocess.run([exe, "-h"], capture_output=True, text=True)
    assert r.returncode == 0 and "--tol" in r.stdout and "--check-iter" in r.stdout
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1 and "Input file is required" in r.stderr
    r = subprocess.run([exe, "-i", "/nonexistent.mps"], capture_output=True, text=True)
    assert r.returncode == 1 and "does not exist" in r.stderr
    r = subprocess.run([exe, "-i", os.path.join(DATA, "lp_small.mps"), "--bogus", "1"], capture_output=True, text=True)
    assert r.returncode == 1 and "Unknown option" in r.stderr
    if not os.path.exists("/dev/kfd"):
        r = subprocess.run([exe, "-i", os.path.join(DATA, "lp_small.mps"), "--tol", "1e-6", "--presolve", "false"],
                           capture_output=True, text=True)
        assert r.returncode == 2 and "nRow = 2, nCol = 2, nnz A = 4" in r.stdout and "status = ERROR" in r.stdout


@pytest.mark.gpu
def test_cli_driver_solves_on_gpu(gpu):
    exe = os.path.join(ROOT, "bin", "solve_mps_file")
    r = subprocess.run([exe, "-i", os.path.join(DATA, "lp_small.mps"), "--tol", "1e-8"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "status = OPTIMAL" in r.stdout and "primal_obj = -26.4" in r.stdout
