"""CPU: create_model_from_mps against hand-computed models (behaviour list in
hpr-lp-c_amd/csrc/mps_reader.cpp, taken from reference src/mps_reader.cpp)."""
import gzip
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, hprlp, lpgen

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


def write_mps(path, lp, two_per_card=True, use_ranges=True):
    """A plain MPS writer for the round-trip tests (test code: the product only reads).  Rows: E when AL == AU, L / G for
    one-sided rows, two-sided rows as L + RANGES (or G + RANGES for every other one); bounds through UP / LO / MI / FX / FR."""
    from scipy import sparse
    m, n = lp["m"], lp["n"]
    A = sparse.csr_matrix((lp["values"], lp["colind"], lp["rowptr"]), shape=(m, n)).tocsc()
    AL, AU, l, u, c = lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"]
    fin_l, fin_u = np.isfinite(AL), np.isfinite(AU)
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "wt") as f:
        f.write("NAME          ROUNDTRIP\nROWS\n N  COST\n")
        kind = []
        for i in range(m):
            if fin_l[i] and fin_u[i]:
                k = "E" if AL[i] == AU[i] else ("L" if i % 2 == 0 else "G")
            elif fin_u[i]:
                k = "L"
            elif fin_l[i]:
                k = "G"
            else:
                k = "N"  # a free row: written as a rim N row, dropped by the reader
            kind.append(k)
            f.write(f" {k}  R{i}\n")
        f.write("COLUMNS\n")
        for j in range(n):
            ent = []
            if c[j] != 0.0:
                ent.append(("COST", c[j]))
            for p in range(A.indptr[j], A.indptr[j + 1]):
                ent.append((f"R{A.indices[p]}", A.data[p]))
            step = 2 if two_per_card else 1
            for q in range(0, len(ent), step):
                f.write(f"    C{j}  " + "  ".join(f"{r}  {repr(float(v))}" for r, v in ent[q:q + step]) + "\n")
        f.write("RHS\n")
        for i in range(m):
            rhs = AU[i] if kind[i] in "EL" else (AL[i] if kind[i] == "G" else 0.0)
            if kind[i] != "N" and rhs != 0.0:
                f.write(f"    RHS  R{i}  {repr(float(rhs))}\n")
        if use_ranges:
            f.write("RANGES\n")
            for i in range(m):
                if kind[i] in "LG" and fin_l[i] and fin_u[i]:
                    f.write(f"    RNG  R{i}  {repr(float(AU[i] - AL[i]))}\n")
        f.write("BOUNDS\n")
        for j in range(n):
            lo, hi = l[j], u[j]
            if lo == hi:
                f.write(f" FX BND  C{j}  {repr(float(lo))}\n")
                continue
            if lo == -INF and hi == INF:
                f.write(f" FR BND  C{j}\n")
                continue
            if lo == -INF:
                f.write(f" MI BND  C{j}\n")
            elif lo != 0.0:
                f.write(f" LO BND  C{j}  {repr(float(lo))}\n")
            if hi != INF:
                f.write(f" UP BND  C{j}  {repr(float(hi))}\n")
        f.write("ENDATA\n")
    return kind


@pytest.mark.parametrize("which,gz", [("c2", False), ("c3", True)])
def test_round_trip_at_benchmark_size(tmp_path, which, gz):
    """BASELINE configs 2 and 3 (Netlib 25fv47 / pds-20 scale stand-ins) written as MPS text and read back through
    create_model_from_mps: every array of the model must come back exactly (values are written with repr, i.e. round-trip
    exact decimal strings; two-sided rows go through RANGES, whose AU - AL arithmetic is allowed one rounding)."""
    lp = lpgen.c2_25fv47_like() if which == "c2" else lpgen.c3_pds20_like()
    lp = dict(lp)
    rng = np.random.default_rng(7)
    n, m = lp["n"], lp["m"]
    # richer bounds than the generator's: some free, some lower-unbounded, some fixed variables; some two-sided rows
    l, u = lp["l"].copy(), lp["u"].copy()
    pick = rng.random(n)
    l[pick < 0.02] = -INF
    u[(pick >= 0.02) & (pick < 0.03)] = l[(pick >= 0.02) & (pick < 0.03)]
    free = (pick >= 0.03) & (pick < 0.04)
    l[free], u[free] = -INF, INF
    neg_up = (pick >= 0.04) & (pick < 0.045) & np.isfinite(u) & (l == 0)
    lp["l"], lp["u"] = l, u
    AL, AU = lp["AL"].copy(), lp["AU"].copy()
    two = np.isfinite(AU) & ~np.isfinite(AL) & (rng.random(m) < 0.2)
    AL[two] = AU[two] - rng.uniform(0.5, 2.0, size=int(two.sum()))
    lp["AL"], lp["AU"] = AL, AU
    del neg_up
    path = tmp_path / (f"{which}.mps.gz" if gz else f"{which}.mps")
    kind = write_mps(path, lp, two_per_card=(which == "c2"))
    mod = hprlp.Model.from_mps(path)
    keep = np.array([k != "N" for k in kind])
    assert (mod.m, mod.n) == (int(keep.sum()), n)
    from scipy import sparse
    A = sparse.csr_matrix((lp["values"], lp["colind"], lp["rowptr"]), shape=(m, n))[keep]
    A.sort_indices()
    rp, ci, v = mod.csr()
    assert np.array_equal(rp, A.indptr) and np.array_equal(ci, A.indices) and np.array_equal(v, A.data)
    vec = mod.vectors()
    assert np.array_equal(vec["c"], lp["c"]) and np.array_equal(vec["l"], lp["l"]) and np.array_equal(vec["u"], lp["u"])
    ALk, AUk, kk = lp["AL"][keep], lp["AU"][keep], np.array(kind)[keep]
    ranged = np.isfinite(ALk) & np.isfinite(AUk) & (kk != "E")
    for got, want in ((vec["AL"], ALk), (vec["AU"], AUk)):
        assert np.array_equal(got[~ranged], want[~ranged])
        np.testing.assert_allclose(got[ranged], want[ranged], rtol=4e-16, atol=1e-15)
    mod.free()


@pytest.mark.gpu
def test_run_mps_dir_emits_the_baseline_table(gpu, tmp_path):
    """tools/run_mps_dir.py: a directory of .mps(.gz) files -> one row of the BASELINE.md section-4 table per instance
    (the reference's driver loop, src/solve_mps_file.cpp:120-131, over a directory).  Here: the config-2 stand-in, a
    small planted LP (gzipped) and the reference's own known answer (data/model.mps content)."""
    import gzip
    import json
    import shutil
    lp2 = lpgen.c2_25fv47_like()
    write_mps(str(tmp_path / "c2like.mps"), lp2)
    lp1 = lpgen.planted_lp(120, 200, 1000, 5)
    write_mps(str(tmp_path / "planted.mps"), lp1)
    with open(tmp_path / "planted.mps", "rb") as f, gzip.open(tmp_path / "planted.mps.gz", "wb") as g:
        g.write(f.read())
    os.remove(tmp_path / "planted.mps")
    shutil.copy(os.path.join(ROOT, "tests", "data", "lp_small.mps"), tmp_path / "model.mps")
    out, js = tmp_path / "table.md", tmp_path / "rows.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_mps_dir.py"), str(tmp_path), "--tol", "1e-6", "--presolve", "false",
                        "--out", str(out), "--json", str(js)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    rows = {q["instance"]: q for q in json.load(open(js))}
    assert set(rows) == {"c2like.mps", "planted.mps.gz", "model.mps"}
    for name, q in rows.items():
        assert q["status"] == "OPTIMAL", (name, q)
        assert max(q["kkt_primal"], q["kkt_dual"], q["kkt_gap"]) <= 1e-5, (name, q)
        assert q["iter4"] <= q["iterations"] and q["time4_s"] <= q["solver_time_s"] + 1e-9
    assert abs(rows["model.mps"]["primal_obj"] - (-26.4)) <= 1e-4          # reference examples/cpp/example_direct_lp.cpp:14
    assert abs(rows["c2like.mps"]["primal_obj"] - lp2["obj_star"]) <= 1e-4 * (1 + abs(lp2["obj_star"]))
    assert "k_small_iterations" in rows["c2like.mps"]["kernels"]
    text = open(out).read()
    assert text.count("\n") == 2 + 3 and "| c2like.mps | 821 | 1571 |" in text
