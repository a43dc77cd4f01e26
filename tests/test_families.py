"""The four Mittelmann-family generators (hpr-lp-c_amd/lpgen.py, round 4: multicommodity flow "pds", QAP relaxation "nug", PDE
boundary control "cont", staircase) -- real LPs with the families' structure, no planted optimum.

CPU: the generators are deterministic (checksums pinned in tests/golden/family_optima.json next to the HiGHS optimum the golden
script computed) and the oracle (restatement of reference src/HPRLP.cu:154-310) reaches that optimum.
GPU: the whole solve through HPRLP_main_solve equals the oracle's -- status, iteration count, objective -- at 1e-4 and 1e-6."""
import json
import os

import numpy as np
import pytest

from conftest import hprlp, lpgen
from oracle import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "family_optima.json")))
NAMES = sorted(lpgen.FAMILIES_SMALL)


def _args(lp):
    return (lp["m"], lp["n"], lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])


@pytest.mark.parametrize("name", NAMES)
def test_generators_are_deterministic_and_the_oracle_reaches_the_highs_optimum(name):
    lp = lpgen.FAMILIES_SMALL[name]()
    g = GOLD[name]
    assert (lp["m"], lp["n"], len(lp["values"])) == (g["m"], g["n"], g["nnz"])
    assert abs(float(np.sum(lp["values"])) - g["checksum_values"]) <= 1e-9 * (1 + abs(g["checksum_values"]))
    assert abs(float(np.sum(lp["c"])) - g["checksum_c"]) <= 1e-9 * (1 + abs(g["checksum_c"]))
    assert len(lp["values"]) <= 100_000
    r = O.solve(*_args(lp), params=O.Params.default(stop_tol=1e-6, max_iter=400_000))
    assert r["status"] == "OPTIMAL", (name, r["iter"])
    assert abs(r["primal_obj"] - g["objective"]) <= 2e-5 * (1 + abs(g["objective"])), (r["primal_obj"], g["objective"])


def test_family_shapes():
    """The structural claims of the generators' docstrings: +-1 node-arc blocks with coupling rows, the nug row / column
    counts (nfac = 8: 912 x 1632 as the nug08 LP), five-point stencil rows, staircase block pattern."""
    nug8 = lpgen.qap_lp_relaxation(8, 1)
    assert (nug8["m"], nug8["n"]) == (912, 1632)
    pds = lpgen.FAMILIES_SMALL["pds_like"]()
    assert set(np.unique(pds["values"])) == {-1.0, 1.0}
    cont = lpgen.pde_control_lp(10, 3)
    stencil_rows = np.diff(cont["rowptr"])[:100]
    assert stencil_rows.max() <= 7 and stencil_rows.min() >= 3    # 5-point stencil (+ two control columns on the boundary)
    st = lpgen.staircase_lp(5, 20, 30, 4, 2)
    A = st["A"].tocoo()
    stage_r, stage_c = A.row // 20, A.col // 30
    assert ((stage_c == stage_r) | (stage_c == stage_r - 1)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("tol", [1e-4, 1e-6])
@pytest.mark.parametrize("name", NAMES)
def test_family_whole_solve_equals_the_oracle(gpu, name, tol):
    lp = lpgen.FAMILIES_SMALL[name]()
    model = hprlp.Model.from_csr(*_args(lp))
    r = model.solve(hprlp.Parameters(stop_tol=tol, use_presolve=False, max_iter=400_000))
    ref = O.solve(*_args(lp), params=O.Params.default(stop_tol=tol, max_iter=400_000))
    assert r.status == ref["status"] == "OPTIMAL"
    # iteration counts agree up to a fork at a thresholded restart decision (parity); most runs agree exactly
    assert abs(r.iter - ref["iter"]) <= 0.1 * ref["iter"] + 150, (r.iter, ref["iter"])
    assert abs(r.primal_obj - ref["primal_obj"]) <= 20 * tol * (1 + abs(ref["primal_obj"])), (r.primal_obj, ref["primal_obj"])
    assert abs(r.primal_obj - GOLD[name]["objective"]) <= 20 * tol * (1 + abs(GOLD[name]["objective"]))
    k = hprlp.original_kkt(model, r.x, r.y, r.z)
    assert max(k["primal_feas"], k["dual_feas"], k["gap"]) <= 10 * tol, k
    # with presolve on: same optimum
    rp = model.solve(hprlp.Parameters(stop_tol=tol, use_presolve=True, max_iter=400_000))
    assert rp.status == "OPTIMAL" and abs(rp.primal_obj - GOLD[name]["objective"]) <= 50 * tol * (1 + abs(GOLD[name]["objective"]))
    model.free()


@pytest.mark.gpu
def test_kernel_form_follows_the_gather_pattern(gpu):
    """Round 4, late (solver.cpp: build_tiled_copy, pb_fallback_wanted).  Which form runs is decided from measurable properties
    of the pattern: rows whose neighbours gather from the same 64-byte lines keep the stream kernel (a grid stencil: cont-like).
    A narrow band with ten entries of a row per 1024-column tile used to be declined by the tiled build (segments of more than four
    entries went to the remainder whole) and kept the stream kernel at 0.29 of 8 TB/s; since round 5 the tile lists have LAYERS
    (tiled.h: kTileLayers) and it takes the lowered fused tiled form with three layers per tile (0.46-0.48).  The iterates follow
    the oracle."""
    import bench_helpers as bh
    lp = lpgen.pde_control_lp(560, 43)
    model = hprlp.Model.from_csr(*_args(lp))
    s = hprlp.Solver(model, hprlp.Parameters(use_presolve=False))
    d = s.describe()
    assert "stream kernel" in d.split("A^T:")[0] and "neighbouring rows gather from the same lines" in d, d
    s.close()
    model.free()

    m = n = 600_000
    lp = bh.banded_lp(m, n, 40, 4000, seed=3)  # ten entries of a row per 1024-column tile: three layers
    model = hprlp.Model.from_csr(m, n, lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])
    s = hprlp.Solver(model, hprlp.Parameters(use_presolve=False))
    d = s.describe()
    assert "all-remainder" not in d and d.count("tiled, fused") == 2 and "tiles of 1024 columns" in d, d
    staged = [int(v) for v in __import__("re").findall(r"(\d+) % of the entries in staged tiles", d)]
    assert len(staged) == 2 and min(staged) >= 85, d      # (31 % before the layers)
    # twelve iterations and a check step against the oracle on the data the device holds
    op = O.Params.default()
    ref = O.ScaledLP(m, n, lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"], op)
    s.scale()
    for name, arr in (("A_val", ref.Av), ("AT_val", ref.ATv), ("AL", ref.AL), ("AU", ref.AU), ("l", ref.l), ("u", ref.u), ("c", ref.c),
                      ("row_norm", ref.row_norm), ("col_norm", ref.col_norm)):
        arr[:] = s.get(name)
    lam = 1.7
    st = ref.new_state()
    s.init(1.0, lam)
    s.iterate(12, True)
    for k in range(12):
        ref.x_half(st, 1.0, k, 0)
        ref.y_half(st, 1.0, lam, k, 0)
    ref.x_half(st, 1.0, 12, 1)
    ref.y_half(st, 1.0, lam, 12, 1)
    for name in ("x", "y", "x_bar", "y_bar"):
        np.testing.assert_allclose(s.get(name), st[name], rtol=1e-10, atol=1e-11, err_msg=name)   # (the tiled kernel's summation order)
    s.close()
    model.free()
