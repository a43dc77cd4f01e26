"""Edge cases of the boundary on the GPU: ragged/empty rows, infinite and free bounds, CSC input,
limits and error statuses (behaviour of reference src/HPRLP.cu:321-524, main_iterate.cu:406-420)."""
import os

import numpy as np
import pytest
from scipy import sparse
from scipy.optimize import linprog

from conftest import hprlp, lpgen
from oracle import oracle as O

pytestmark = pytest.mark.gpu
INF = np.inf


def highs(A, AL, AU, l, u, c):
    """Independent optimum: split AL<=Ax<=AU into equality / one-sided rows for HiGHS."""
    A = sparse.csr_matrix(A)
    eq = np.isfinite(AL) & np.isfinite(AU) & (AL == AU)
    ub_rows, ub_rhs = [], []
    for i in np.where(~eq)[0]:
        if np.isfinite(AU[i]):
            ub_rows.append(A[i]); ub_rhs.append(AU[i])
        if np.isfinite(AL[i]):
            ub_rows.append(-A[i]); ub_rhs.append(-AL[i])
    kw = {}
    if ub_rows:
        kw.update(A_ub=sparse.vstack(ub_rows), b_ub=np.array(ub_rhs))
    if eq.any():
        kw.update(A_eq=A[eq], b_eq=AU[eq])
    bounds = [(None if not np.isfinite(a) else a, None if not np.isfinite(b) else b) for a, b in zip(l, u)]
    r = linprog(c, bounds=bounds, method="highs", **kw)
    assert r.status == 0
    return r.fun


def solve(A, AL, AU, l, u, c, **kw):
    prm = hprlp.Parameters(use_presolve=False, **kw)
    return hprlp.solve(A, np.asarray(AL, float), np.asarray(AU, float), np.asarray(l, float), np.asarray(u, float),
                       np.asarray(c, float), prm)


def test_empty_rows_and_columns(gpu):
    rng = np.random.default_rng(1)
    A = sparse.random(40, 60, density=0.08, random_state=3, format="lil")
    A[5, :] = 0; A[17, :] = 0          # empty rows (0 within [AL,AU])
    A[:, 9] = 0; A[:, 33] = 0          # empty columns (bounded, so the LP stays bounded)
    A = sparse.csr_matrix(A); A.eliminate_zeros()
    x0 = rng.uniform(0, 1, 60)
    b = A @ x0
    AL = np.full(40, -INF); AU = b + 0.5
    AL[5] = -1.0; AU[5] = 2.0; AL[17] = 0.0; AU[17] = 0.0
    c = rng.normal(size=60)
    l = np.zeros(60); u = np.full(60, 3.0)
    r = solve(A, AL, AU, l, u, c, stop_tol=1e-8)
    assert r.status == "OPTIMAL"
    want = highs(A, AL, AU, l, u, c)
    assert abs(r.primal_obj - want) <= 1e-6 * (1 + abs(want))


def test_free_variables_and_two_sided_rows(gpu):
    rng = np.random.default_rng(2)
    A = sparse.random(50, 30, density=0.2, random_state=4, format="csr")
    x0 = rng.normal(size=30)
    b = A @ x0
    AL = b - rng.uniform(0.1, 1.0, 50); AU = b + rng.uniform(0.1, 1.0, 50)   # ranged rows
    c = A.T @ rng.normal(size=50) * 0.1
    l = np.full(30, -INF); u = np.full(30, INF)                               # all free
    l[:5] = -2.0; u[5:10] = 2.0                                               # lower-only / upper-only mixes
    r = solve(A, AL, AU, l, u, c, stop_tol=1e-8)
    want = highs(A, AL, AU, l, u, c)
    assert r.status == "OPTIMAL"
    assert abs(r.primal_obj - want) <= 1e-6 * (1 + abs(want))


def test_row_length_boundaries(gpu):
    """Rows of 255/256/257/511/512/513/1500 nonzeros cross the stream/vector mode and block limits."""
    rng = np.random.default_rng(3)
    n = 2000
    lens = [255, 256, 257, 511, 512, 513, 1500, 1, 0, 3]
    rows, cols, vals = [], [], []
    for i, L in enumerate(lens):
        cc = np.sort(rng.choice(n, size=L, replace=False))
        rows += [i] * L; cols += list(cc); vals += list(rng.normal(size=L))
    A = sparse.csr_matrix((vals, (rows, cols)), shape=(len(lens), n))
    x0 = rng.uniform(0, 1, n)
    b = A @ x0
    AL = b.copy(); AU = b.copy()
    c = rng.normal(size=n)
    l = np.zeros(n); u = np.ones(n)
    r = solve(A, AL, AU, l, u, c, stop_tol=1e-8)
    want = highs(A, AL, AU, l, u, c)
    assert r.status == "OPTIMAL"
    assert abs(r.primal_obj - want) <= 1e-6 * (1 + abs(want))
    # the same matrix as CSC input must give the same model (reference src/HPRLP.cu:354-396)
    Ac = sparse.csc_matrix(A)
    m1 = hprlp.Model.from_csr(len(lens), n, A.indptr, A.indices, A.data, AL, AU, l, u, c)
    m2 = hprlp.Model.from_csr(len(lens), n, Ac.indptr, Ac.indices, Ac.data, AL, AU, l, u, c, is_csc=True)
    for a, b2 in zip(m1.csr(), m2.csr()):
        assert np.array_equal(a, b2)
    m1.free(); m2.free()


def test_one_by_one(gpu):
    r = solve(np.array([[2.0]]), [1.0], [INF], [0.0], [INF], [3.0], stop_tol=1e-9)   # min 3x, 2x>=1
    assert r.status == "OPTIMAL" and abs(r.x[0] - 0.5) < 1e-7 and abs(r.primal_obj - 1.5) < 1e-7


def test_iteration_limit_and_defaults(gpu, model_mps_arrays):
    a = model_mps_arrays
    model = hprlp.Model.from_csr(a["m"], a["n"], a["rowptr"], a["colind"], a["values"], a["AL"], a["AU"], a["l"],
                                 a["u"], a["c"])
    r = model.solve(hprlp.Parameters(max_iter=57, stop_tol=1e-12, use_presolve=False))
    assert r.status == "ITER_LIMIT" and r.iter == 57 and r.x is not None
    ref = O.solve(a["m"], a["n"], a["rowptr"], a["colind"], a["values"], a["AL"], a["AU"], a["l"], a["u"], a["c"],
                  params=O.Params.default(max_iter=57, stop_tol=1e-12))
    assert ref["status"] == "ITER_LIMIT" and ref["iter"] == 57
    np.testing.assert_allclose(r.x, ref["x"], rtol=0, atol=1e-12)
    r = model.solve(hprlp.Parameters(time_limit=0.0, stop_tol=1e-14, use_presolve=False))
    assert r.status == "TIME_LIMIT"
    # param == NULL -> defaults (stop_tol 1e-4): reference src/HPRLP.cu:501-502
    res = hprlp.lib().solve(model._ptr, None)
    r = hprlp.Results(res, 2, 2)
    assert r.status == "OPTIMAL" and r.iter == 180 and abs(r.primal_obj + 26.4) < 1e-2
    # scaling switches are honoured
    r = model.solve(hprlp.Parameters(stop_tol=1e-8, use_CR_scaling=False, use_Ruiz_scaling=False,
                                     use_Pock_Chambolle_scaling=False, use_bc_scaling=False, use_presolve=False))
    assert r.status == "OPTIMAL" and abs(r.primal_obj + 26.4) < 1e-5
    model.free()


def test_null_model_is_error(gpu):
    res = hprlp.lib().solve(None, None)
    assert res.status == b"ERROR" and not res.x and not res.y and not res.z


def test_independent_kkt_on_c3(gpu):
    """BASELINE config 3 size: the reported residuals equal KKT residuals recomputed from x,y,z on the
    UNSCALED problem with scipy (formulas of reference src/pslp_integration.cpp:499-580)."""
    lp = lpgen.c3_pds20_like()
    model = hprlp.Model.from_csr(lp["m"], lp["n"], lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"],
                                 lp["l"], lp["u"], lp["c"])
    r = model.solve(hprlp.Parameters(stop_tol=1e-6, use_presolve=False))
    assert r.status == "OPTIMAL"
    A = lp["A"]
    Ax = A @ r.x
    rp = np.linalg.norm(np.maximum(np.minimum(lp["AU"] - Ax, 0), lp["AL"] - Ax))
    bnorm = np.linalg.norm(np.maximum(np.abs(np.where(np.isinf(lp["AL"]), 0, lp["AL"])), np.abs(np.where(np.isinf(lp["AU"]), 0, lp["AU"]))))
    rd = np.linalg.norm(lp["c"] - A.T @ r.y - r.z)
    assert rp / (1 + bnorm) <= 1e-6 and rd / (1 + np.linalg.norm(lp["c"])) <= 1e-6
    assert abs(lp["c"] @ r.x - r.primal_obj) <= 1e-9 * (1 + abs(r.primal_obj))
    assert abs(r.primal_obj - lp["obj_star"]) <= 1e-5 * (1 + abs(lp["obj_star"]))
    assert (r.x >= lp["l"] - 1e-9).all() and (r.x <= lp["u"] + 1e-9).all()
    model.free()


def test_dense_rows_and_columns_are_split(gpu):
    """A row with 9000 nonzeros and a column with 5000 (both above the 4096 split threshold): chunked over
    several waves, finished by k_long_finish; compared with the oracle step by step and with HiGHS."""
    from test_gpu_kernels import NAMES_M, NAMES_N, adopt_gpu_data, run_steps
    rng = np.random.default_rng(8)
    m, n = 6000, 12000
    A = sparse.random(m, n, density=0.0004, random_state=5, format="lil")
    cols = np.sort(rng.choice(n, size=9000, replace=False))
    A[17, cols] = rng.uniform(0.5, 1.5, size=9000)
    rows = np.sort(rng.choice(m, size=5000, replace=False))
    A[rows, 33] = rng.uniform(0.5, 1.5, size=5000).reshape(-1, 1)
    # a row whose last chunk holds a single entry (4097 = 4096 + 1: round 2 found such a row refused with "bad split-row block")
    A[40, :] = 0
    A[40, np.sort(rng.choice(n, size=4097, replace=False))] = rng.uniform(0.5, 1.5, size=4097)
    A = sparse.csr_matrix(A); A.sort_indices()
    assert np.diff(A.indptr).max() >= 9000 and np.diff(sparse.csc_matrix(A).indptr).max() >= 5000 and np.diff(A.indptr)[40] == 4097
    x0 = rng.uniform(0, 1, n)
    b = A @ x0
    AL = np.full(m, -INF); AU = b + 0.1
    c = rng.uniform(-1, 1, n); l = np.zeros(n); u = np.full(n, 2.0)
    model = hprlp.Model.from_csr(m, n, A.indptr, A.indices, A.data, AL, AU, l, u, c)
    s = hprlp.Solver(model, hprlp.Parameters(use_presolve=False, use_CR_scaling=False))
    ref = O.ScaledLP(m, n, A.indptr, A.indices, A.data, AL, AU, l, u, c, O.Params.default(use_CR_scaling=0))
    s.scale()
    adopt_gpu_data(s, ref)
    st = run_steps(s, ref, 0.7, 1.2, [(12, True), (3, True)])
    for name in NAMES_N + NAMES_M:
        np.testing.assert_allclose(s.get(name), st[name], rtol=1e-10, atol=1e-12, err_msg=name)   # tree/chunk order
    got = s.residuals(16, True)
    assert np.isfinite(got["kkt"]) and np.isfinite(got["weighted_norm"])
    s.close()
    r = model.solve(hprlp.Parameters(stop_tol=1e-6, use_presolve=False))
    assert r.status == "OPTIMAL"
    want = highs(A, AL, AU, l, u, c)
    assert abs(r.primal_obj - want) <= 1e-4 * (1 + abs(want))
    model.free()


def test_dense_rows_of_a_large_matrix_are_cut_by_column_eighths(gpu):
    """Round 4 (solver.cpp: slab_cuts): in a matrix whose gathered vector is beyond one L2 (>= 2^19 columns, >= 2^22 entries)
    a row of 2048 entries and more is cut where its columns cross into another eighth of the vector, the chunks are pinned to
    the XCD share whose own rows read that eighth, k_long_finish adds them in order.  Rows / columns of 2048 .. 9000 entries
    (one of them with all its entries in ONE eighth, one with a 4097-entry run in one eighth: cut again at kSplitRow; 4100 in
    one eighth and 4050 + 60 in two: a short tail that may not join its predecessor) in a
    block-diagonal matrix: iterates against the oracle over normal and check steps, and against the uncut form."""
    from test_gpu_kernels import NAMES_M, NAMES_N, adopt_gpu_data, run_steps
    rng = np.random.default_rng(12)
    m, n, per_row = 560_000, 640_000, 8
    blk = 80   # diagonal blocks: 7000 rows x 8000 columns
    r = np.repeat(np.arange(m), per_row)
    c = (r // (m // blk)) * (n // blk) + rng.integers(0, n // blk, size=len(r))
    rows, cols = [r], [c]
    for i, L in zip(rng.choice(m, 4, replace=False), (2048, 3000, 5000, 9000)):       # dense rows of A over all columns
        rows.append(np.full(L, i)); cols.append(rng.choice(n, L, replace=False))
    i1 = int(rng.integers(0, m))
    rows.append(np.full(2500, i1)); cols.append(3 * (n // 8) + rng.choice(n // 8 - 10, 2500, replace=False))   # all in one eighth
    i2 = int(rng.integers(0, m))
    rows.append(np.full(4500, i2)); cols.append(5 * (n // 8) + rng.choice(n // 8 - 10, 4500, replace=False))   # > kSplitRow in one eighth
    i3, i4 = (int(x) for x in rng.choice(m, 2, replace=False))
    rows.append(np.full(4100, i3)); cols.append(1 * (n // 8) + rng.choice(n // 8 - 10, 4100, replace=False))   # 4096 + a 4-entry tail that must NOT join
    rows.append(np.full(4110, i4)); cols.append(np.concatenate([2 * (n // 8) + rng.choice(n // 8 - 10, 4050, replace=False),
                                                                 6 * (n // 8) + rng.choice(n // 8 - 10, 60, replace=False)]))   # 4050 + 60: the same
    for j, L in zip(rng.choice(n, 3, replace=False), (2100, 4097, 7000)):              # dense columns (rows of A^T)
        rows.append(rng.choice(m, L, replace=False)); cols.append(np.full(L, j))
    rr, cc = np.concatenate(rows), np.concatenate(cols)
    A = sparse.csr_matrix((np.ones(len(rr)), (rr, cc)), shape=(m, n))
    A.sum_duplicates()
    A.data = rng.uniform(0.5, 1.5, size=A.nnz) * rng.choice([-1.0, 1.0], size=A.nnz)
    A.sort_indices()
    assert A.nnz >= 1 << 22 and np.diff(A.indptr).max() >= 9000 and np.diff(A.tocsc().indptr).max() >= 7000
    x0 = rng.uniform(0, 1, n)
    b = A @ x0
    rp, ci, v = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.copy()
    AL, AU, l, u, cost = b - 0.5, b + 0.5, np.zeros(n), np.full(n, 2.0), rng.uniform(-1, 1, n)
    model = hprlp.Model.from_csr(m, n, rp, ci, v, AL, AU, l, u, cost)
    ref = O.ScaledLP(m, n, rp, ci, v, AL, AU, l, u, cost, O.Params.default(use_CR_scaling=0))
    states = {}
    for cut in (True, False):
        os.environ["HPRLP_NO_TILED"] = "1"   # (this pattern would take the tiled kernel with its long rows aside: here the stream kernel's rows are meant)
        if not cut:
            os.environ["HPRLP_NO_SLAB_CUTS"] = "1"
        try:
            s = hprlp.Solver(model, hprlp.Parameters(use_presolve=False, use_CR_scaling=False))
            assert s.info()["tiled"] == 0
            d = s.describe()
            nsplit = [int(x) for x in __import__("re").findall(r"(\d+) split rows", d)]
            assert nsplit == ([8, 3] if cut else [5, 2]), d     # cut: every row of 2048+; uncut: only rows over 4096
            s.scale()
            if cut:
                adopt_gpu_data(s, ref)
                st = run_steps(s, ref, 0.7, 1.2, [(9, True), (3, True), (5, False)])
                for name in NAMES_N + NAMES_M:
                    np.testing.assert_allclose(s.get(name), st[name], rtol=1e-10, atol=1e-12, err_msg=name)   # tree / chunk order
            else:
                s.init(0.7, 1.2)
                s.iterate(9, True); s.iterate(3, True); s.iterate(5, False)
            states[cut] = {k: s.get(k) for k in ("x", "y", "x_bar", "y_bar")}
            s.close()
        finally:
            os.environ.pop("HPRLP_NO_SLAB_CUTS", None)
            os.environ.pop("HPRLP_NO_TILED", None)
    for k in states[True]:
        np.testing.assert_allclose(states[True][k], states[False][k], rtol=1e-10, atol=1e-12, err_msg=k)
    model.free()


def test_duplicate_and_unsorted_entries(gpu):
    """create_model_from_arrays takes the arrays as they are (reference src/HPRLP.cu:343-452 copies them): rows whose
    column indices are not sorted and repeat.  The solve must equal the solve of the canonical form (sorted, duplicates
    summed) -- the kernels add duplicates as separate entries."""
    lp = lpgen.planted_lp(300, 500, 3000, 12)
    m, n = lp["m"], lp["n"]
    A = sparse.csr_matrix((lp["values"], lp["colind"], lp["rowptr"]), shape=(m, n))
    rng = np.random.default_rng(5)
    rp, ci, v = [0], [], []
    for i in range(m):
        cols = A.indices[A.indptr[i]:A.indptr[i + 1]]
        vals = A.data[A.indptr[i]:A.indptr[i + 1]]
        cc, vv = [], []
        for c_, a in zip(cols, vals):
            if rng.random() < 0.3:  # split the entry in two: a = 0.25 a + 0.75 a exactly representable parts
                cc += [c_, c_]; vv += [0.25 * a, 0.75 * a]
            else:
                cc.append(c_); vv.append(a)
        order = rng.permutation(len(cc))
        ci += [cc[k] for k in order]; v += [vv[k] for k in order]
        rp.append(len(ci))
    raw = hprlp.Model.from_csr(m, n, np.array(rp, np.int32), np.array(ci, np.int32), np.array(v), lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])
    canon = hprlp.Model.from_csr(m, n, lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])
    prm = hprlp.Parameters(stop_tol=1e-6, use_presolve=False, max_iter=200000)
    r_raw, r_can = raw.solve(prm), canon.solve(prm)
    assert r_raw.status == r_can.status == "OPTIMAL"
    assert abs(r_raw.primal_obj - lp["obj_star"]) <= 1e-5 * (1 + abs(lp["obj_star"]))
    assert abs(r_raw.primal_obj - r_can.primal_obj) <= 1e-5 * (1 + abs(r_can.primal_obj))
    # (iteration counts differ: the scalings see 0.75 a where the canonical form has a)
    # the optimum is not unique: check the raw solve's primal-dual triple against the CANONICAL model instead
    k = hprlp.original_kkt(canon, r_raw.x, r_raw.y, r_raw.z)
    assert max(k["primal_feas"], k["dual_feas"], k["gap"]) <= 1e-5, k
    # with presolve on the same holds (the presolver counts structure per entry)
    r_pre = raw.solve(hprlp.Parameters(stop_tol=1e-6, use_presolve=True, max_iter=200000))
    assert r_pre.status == "OPTIMAL" and abs(r_pre.primal_obj - lp["obj_star"]) <= 1e-5 * (1 + abs(lp["obj_star"]))
    raw.free(); canon.free()


@pytest.mark.parametrize("nnz_per_row", [3, 9])
def test_out_of_range_column_index_is_refused(gpu, nnz_per_row):
    """The kernels index without bounds checks.  create_model_from_arrays refuses a column index outside [0, n) (NULL model);
    a caller that fills an LP_info_cpu itself (the C ABI allows it: reference include/structs.h:243-252) gets past that, so the
    solver checks again before it launches on the matrix: on the host for small matrices, by a pass over the uploaded copy above
    4e6 entries (hpr-lp-c_amd/csrc/solver.cpp: DeviceMatrix::upload)."""
    m = 500_000
    n = m
    rng = np.random.default_rng(1)
    nnz = m * nnz_per_row     # 1.5e6: host check; 4.5e6: device check
    rp = np.arange(0, nnz + 1, nnz_per_row, dtype=np.int32)
    ci = np.sort(rng.integers(0, n, size=(m, nnz_per_row)), axis=1).astype(np.int32).ravel()
    v = np.ones(nnz)
    args = (np.zeros(m), np.ones(m), np.zeros(n), np.ones(n), np.ones(n))
    bad = ci.copy()
    bad[nnz // 2] = n + 7
    with pytest.raises(Exception):
        hprlp.Model.from_csr(m, n, rp, bad, v, *args)
    model = hprlp.Model.from_csr(m, n, rp, ci, v, *args)
    model._ptr.contents.A.contents.colIndex[nnz // 2] = n + 7      # the model's own copy, behind the library's back
    with pytest.raises(RuntimeError, match="column index out of range"):
        hprlp.Solver(model, hprlp.Parameters(use_presolve=False))
    model.free()


def test_corrupt_row_pointers_of_a_large_hand_built_model_are_refused_before_the_host_reads_them(gpu):
    """Solver::setup samples rowPtr / colIndex on the host (choose_sb_rows, matrices of 4e6 entries and more) BEFORE
    DeviceMatrix::upload validates them: a hand-filled LP_info_cpu with a non-monotone row pointer array must be refused, not
    read out of bounds; rows whose columns are not sorted must not make the span estimate negative (the solver is built and steps)."""
    m = n = 600_000
    per = 8
    nnz = m * per     # 4.8e6: the sampling path
    rng = np.random.default_rng(3)
    rp = np.arange(0, nnz + 1, per, dtype=np.int32)
    ci = ((np.arange(m)[:, None] + rng.integers(-2000, 2000, size=(m, per))) % n).astype(np.int32)   # unsorted within a row
    v = np.ones(nnz)
    args = (np.zeros(m), np.ones(m) * per, np.zeros(n), np.ones(n), np.ones(n))
    model = hprlp.Model.from_csr(m, n, rp, ci.ravel(), v, *args)
    s = hprlp.Solver(model, hprlp.Parameters(use_presolve=False))
    s.scale()
    s.init(1.0, 4.0 * per)
    s.iterate(3, True)
    assert np.isfinite(s.residuals(4)["kkt"])
    s.close()
    A = model._ptr.contents.A.contents
    A.rowPtr[m // 2] = nnz + 12345     # behind the library's back: decreasing afterwards, and past the end of colIndex
    with pytest.raises(RuntimeError, match="row pointer"):
        hprlp.Solver(model, hprlp.Parameters(use_presolve=False))
    A.rowPtr[m // 2] = (m // 2) * per
    A.rowPtr[m] = nnz - 1
    with pytest.raises(RuntimeError, match="row pointer"):
        hprlp.Solver(model, hprlp.Parameters(use_presolve=False))
    A.rowPtr[m] = nnz
    model.free()


def _csr_case(m, n, rp, ci, v, AL, AU, l, u, c):
    f = lambda a: np.asarray(a, float)
    return (m, n, np.asarray(rp, np.int32), np.asarray(ci, np.int32), f(v), f(AL), f(AU), f(l), f(u), f(c))


DEGENERATE = {
    # one dense row over 5000 columns (a single vector-mode row in A, 5000 one-entry rows in A^T)
    "one_row": _csr_case(1, 5000, [0, 5000], np.arange(5000), np.ones(5000), [1.0], [1.0], np.zeros(5000), np.ones(5000),
                         np.arange(1, 5001) / 5000),
    # zero objective: a feasibility problem
    "zero_objective": _csr_case(2, 3, [0, 2, 4], [0, 1, 1, 2], [1, 1, 1, 1], [1, 1], [1, 1], [0, 0, 0], [1, 1, 1], [0, 0, 0]),
    # fixed variables (l == u) and a redundant row
    "fixed_variables": _csr_case(2, 3, [0, 3, 6], [0, 1, 2, 0, 1, 2], [1, 2, 3, 2, 4, 6], [-INF, -INF], [10, 20], [1, 1, 0], [1, 1, 5],
                                 [-1, -1, -1]),
    # infeasible (x >= 2 and x <= 1) and unbounded (min -x, x - y <= 5, both free above): no certificate in HPR-LP, the
    # iteration limit ends the run -- as in the reference (src/main_iterate.cu:406-420 knows OPTIMAL / ITER_LIMIT / TIME_LIMIT only)
    "infeasible": _csr_case(2, 1, [0, 1, 2], [0, 0], [1, 1], [2, -INF], [INF, 1], [0], [10], [1]),
    "unbounded": _csr_case(1, 2, [0, 2], [0, 1], [1, -1], [-INF], [5], [0, 0], [INF, INF], [-1, 0]),
}


@pytest.mark.parametrize("name", sorted(DEGENERATE))
def test_degenerate_lps_follow_the_oracle(gpu, name):
    """Shapes and LPs at the edge of what the iteration is meant for: same status, iteration count and objective as the oracle's
    restatement of the reference loop; with presolve on the answer is the same optimum where there is one."""
    case = DEGENERATE[name]
    m, n = case[0], case[1]
    model = hprlp.Model.from_csr(*case)
    r = model.solve(hprlp.Parameters(stop_tol=1e-8, max_iter=3000, use_presolve=False))
    ref = O.solve(*case, params=O.Params.default(stop_tol=1e-8, max_iter=3000))
    assert r.status == ref["status"] and r.iter == ref["iter"]
    assert r.primal_obj == pytest.approx(ref["primal_obj"], rel=1e-9, abs=1e-12)
    assert (r.status == "ITER_LIMIT") == (name in ("infeasible", "unbounded"))
    if r.status == "OPTIMAL":
        rp = model.solve(hprlp.Parameters(stop_tol=1e-8, max_iter=3000, use_presolve=True))
        assert rp.status == "OPTIMAL" and abs(rp.primal_obj - r.primal_obj) <= 1e-6 * (1 + abs(r.primal_obj))
    model.free()


def test_matrix_without_entries_is_refused_like_the_reference(gpu):
    """create_model_from_arrays with nnz <= 0 returns NULL (reference src/HPRLP.cu:329-332)."""
    with pytest.raises(Exception):
        hprlp.Model.from_csr(3, 4, np.zeros(4, np.int32), np.zeros(0, np.int32), np.zeros(0), [-1, 0, -INF], [1, 0, 2], [0, -1, 0, 2],
                             [1, 1, 5, 3], [1, -2, 0, 3])


def test_a_test_hook_without_the_gate_is_ignored_and_reported(gpu, model_mps_arrays):
    """csrc/env.h (round 5): HPRLP_NO_SMALL=1 (a test hook) in the environment of a process that has NOT set HPRLP_TEST_HOOKS=1
    changes nothing -- the Netlib-scale LP still runs the single-workgroup kernel -- and hprlp_solver_describe says that the
    variable was seen and ignored; with the gate the hook acts and the description names it.  An integrator-facing switch
    (HPRLP_NO_GRAPH) is honoured either way."""
    a = model_mps_arrays
    model = hprlp.Model.from_csr(a["m"], a["n"], a["rowptr"], a["colind"], a["values"], a["AL"], a["AU"], a["l"], a["u"], a["c"])
    saved = {k: os.environ.get(k) for k in ("HPRLP_TEST_HOOKS", "HPRLP_NO_SMALL", "HPRLP_NO_GRAPH")}
    try:
        os.environ["HPRLP_NO_SMALL"] = "1"
        os.environ["HPRLP_NO_GRAPH"] = "1"
        os.environ.pop("HPRLP_TEST_HOOKS", None)
        s = hprlp.Solver(model, hprlp.Parameters(use_presolve=False))
        d = s.describe()
        assert s.info()["tiled"] & 4 and "single-workgroup kernel" in d                       # the hook did nothing
        assert "switches: HPRLP_NO_GRAPH=1" in d and "ignored without HPRLP_TEST_HOOKS=1: HPRLP_NO_SMALL=1" in d, d
        s.close()
        os.environ["HPRLP_TEST_HOOKS"] = "1"
        s = hprlp.Solver(model, hprlp.Parameters(use_presolve=False))
        d = s.describe()
        assert not (s.info()["tiled"] & 4) and "single-workgroup kernel" not in d
        assert "HPRLP_NO_SMALL=1" in d.split("switches:")[1] and "ignored" not in d, d
        s.close()
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    model.free()
