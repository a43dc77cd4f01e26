"""GPU: a large LP handed over in a random row / column order.  The solver finds a locality ordering at set-up
(csrc/reorder.cpp + reorder_dev.hip), runs the column-tiled kernels on P A Q and speaks the caller's numbering at the
boundary: vectors and the solution must match the oracle, which works on the LP exactly as given."""
import os

import numpy as np
import pytest
from scipy import sparse

import bench_helpers as bh
from conftest import hprlp
from oracle import oracle as O
from test_gpu_kernels import NAMES_M, NAMES_N, run_steps

pytestmark = pytest.mark.gpu


def permuted_lp(m, n, per_row, band, seed):
    lp = bh.banded_lp(m, n, per_row, band)
    A = sparse.csr_matrix((lp["values"], lp["colind"], lp["rowptr"]), shape=(m, n))
    rng = np.random.default_rng(seed)
    pr, pc = rng.permutation(m), rng.permutation(n)
    inv_pc = np.empty(n, np.int64); inv_pc[pc] = np.arange(n)
    B = A[pr]
    B = sparse.csr_matrix((B.data, inv_pc[B.indices], B.indptr), shape=(m, n))
    B.sort_indices()
    out = dict(m=m, n=n, rowptr=B.indptr.astype(np.int32), colind=B.indices.astype(np.int32), values=B.data.copy(),
               AL=lp["AL"][pr], AU=lp["AU"][pr], l=lp["l"][pc], u=lp["u"][pc], c=lp["c"][pc], obj_star=lp["obj_star"])
    return out


@pytest.fixture(scope="module")
def lp():
    return permuted_lp(1_600_000, 1_600_000, 10, 16000, 11)


def test_reordered_iterates_match_oracle(gpu, lp):
    m, n = lp["m"], lp["n"]
    model = hprlp.Model.from_csr(m, n, lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])
    s = hprlp.Solver(model, hprlp.Parameters(use_presolve=False, use_CR_scaling=False))
    info = s.info()
    assert info["reordered"] and info["tiled"] == 3, info
    ref = O.ScaledLP(m, n, lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"],
                     O.Params.default(use_CR_scaling=0))
    s.scale()
    # the data vectors come back in the caller's numbering; sums (Pock-Chambolle row sums, norms) run in another order
    for name, want in (("AL", ref.AL), ("AU", ref.AU), ("l", ref.l), ("u", ref.u), ("c", ref.c), ("row_norm", ref.row_norm),
                       ("col_norm", ref.col_norm)):
        got = s.get(name)
        fin = np.isfinite(want)
        assert np.array_equal(np.isfinite(got), fin), name
        np.testing.assert_allclose(got[fin], want[fin], rtol=1e-12, err_msg=name)
    st = run_steps(s, ref, 0.6, 1.4, [(17, True), (5, True), (9, False)])
    for name in NAMES_N + NAMES_M:
        np.testing.assert_allclose(s.get(name), st[name], rtol=1e-11, atol=1e-12, err_msg=name)
    lam, it = s.power_iteration()
    lam_ref, it_ref = ref.power_iteration()
    assert it == it_ref and abs(lam - lam_ref) <= 1e-10 * lam_ref
    s.close(); model.free()


def test_reordered_solve_returns_the_callers_numbering(gpu, lp):
    m, n = lp["m"], lp["n"]
    model = hprlp.Model.from_csr(m, n, lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])
    prm = hprlp.Parameters(stop_tol=1e-4, use_presolve=False, max_iter=20000)
    r = model.solve(prm)
    os.environ["HPRLP_NO_REORDER"] = "1"
    try:
        r0 = model.solve(prm)  # same LP, given order, stream kernel
    finally:
        os.environ.pop("HPRLP_NO_REORDER", None)
    assert r.status == r0.status == "OPTIMAL"
    assert abs(r.iter - r0.iter) <= 0.1 * r0.iter + 150
    assert abs(r.primal_obj - lp["obj_star"]) <= 1e-3 * (1 + abs(lp["obj_star"]))
    np.testing.assert_allclose(r.x, r0.x, rtol=1e-3, atol=1e-3)
    # KKT on the LP as given
    A = sparse.csr_matrix((lp["values"], lp["colind"], lp["rowptr"]), shape=(m, n))
    Ax = A @ r.x
    viol = np.maximum(np.maximum(np.where(np.isfinite(lp["AL"]), lp["AL"] - Ax, 0), np.where(np.isfinite(lp["AU"]), Ax - lp["AU"], 0)), 0)
    b = np.maximum(np.where(np.isfinite(lp["AL"]), np.abs(lp["AL"]), 0), np.where(np.isfinite(lp["AU"]), np.abs(lp["AU"]), 0))
    assert np.linalg.norm(viol) <= 3e-4 * (1 + np.linalg.norm(b))
    rd = lp["c"] - A.T @ r.y - r.z
    assert np.linalg.norm(rd) <= 3e-4 * (1 + np.linalg.norm(lp["c"]))
    model.free()


def test_unstructured_large_matrix_runs_propagation_blocking(gpu):
    """No locality to be found (uniformly random pattern, 4.2 M columns: the gathered vector is far beyond the L2s): the
    ordering is tried and rejected, and the tiled form is accepted WITHOUT a dense-tile requirement -- every entry goes
    through the propagation-blocking remainder.  Same iterates as the stream kernel and as the oracle."""
    rng = np.random.default_rng(21)
    m = n = 4_300_000
    per_row = 3
    cols = rng.integers(0, n, size=(m, per_row))
    cols.sort(axis=1)
    cols[:, 1] = np.where(cols[:, 1] == cols[:, 0], (cols[:, 1] + 1) % n, cols[:, 1])
    cols[:, 2] = np.where((cols[:, 2] == cols[:, 1]) | (cols[:, 2] == cols[:, 0]), (cols[:, 2] + 2) % n, cols[:, 2])
    cols.sort(axis=1)
    keep = np.ones((m, per_row), bool)
    keep[:, 1:] &= cols[:, 1:] != cols[:, :-1]
    rp = np.concatenate([[0], np.cumsum(keep.sum(axis=1))]).astype(np.int32)
    ci = cols[keep].astype(np.int32)
    v = rng.normal(size=len(ci))
    A = sparse.csr_matrix((v, ci, rp), shape=(m, n))
    x0 = rng.uniform(0, 1, size=n)
    b = A @ x0
    AL, AU = b - 1.0, b + 1.0
    l, u, c = np.zeros(n), np.full(n, 2.0), rng.normal(size=n)
    model = hprlp.Model.from_csr(m, n, rp, ci, v, AL, AU, l, u, c)

    def run():
        s = hprlp.Solver(model, hprlp.Parameters(use_presolve=False, use_CR_scaling=False))
        info = s.info()
        s.scale()
        lam, it = s.power_iteration(max_iter=30)
        s.init(0.7, 1.3 * lam)
        s.iterate(12, True)
        out = (info, lam, {k: s.get(k) for k in ("x", "y", "x_bar", "y_bar", "z_bar")}, s.residuals(13, True))
        s.close()
        return out

    info, lam, st, res = run()
    assert info["tiled"] == 3 and not info["reordered"], info
    os.environ["HPRLP_NO_PB_FALLBACK"] = "1"
    try:
        info0, lam0, st0, res0 = run()
    finally:
        os.environ.pop("HPRLP_NO_PB_FALLBACK", None)
    assert info0["tiled"] == 0
    assert abs(lam - lam0) <= 1e-11 * abs(lam0)
    for k in st:
        np.testing.assert_allclose(st[k], st0[k], rtol=1e-10, atol=1e-12, err_msg=k)
    assert abs(res["kkt"] - res0["kkt"]) <= 1e-9 * (1 + abs(res0["kkt"]))
    model.free()


def test_all_remainder_kernel_matches_oracle(gpu):
    """k_pb_fused (kernels.hip, round 4) under the oracle.  A 900 k x 1.1 M LP with 12 uniformly random columns per row plus a
    few rows and columns of several hundred entries: no ordering helps, both matrices take the all-remainder form -- every
    entry's product travels through P, a super-block's rows are summed by the lane-chunk segmented reduction (rows of 12
    span two or three lanes, the long ones whole waves and wave boundaries), the half-steps hand the products over to each
    other.  All iterate vectors of reference src/cuda_kernels/HPR_cuda_kernels.cu:203-295 against the oracle at 1e-11 over
    normal and check iterations, one residual evaluation (src/main_iterate.cu:229-309), lambda_max, and the Curtis-Reid
    scaling through the same kernel."""
    rng = np.random.default_rng(77)
    m, n, per_row = 900_000, 1_100_000, 12
    cols = np.sort(rng.integers(0, n, size=(m, per_row)), axis=1)
    keep = np.ones((m, per_row), bool)
    keep[:, 1:] = cols[:, 1:] != cols[:, :-1]
    r_idx = np.repeat(np.arange(m), keep.sum(axis=1))
    c_idx = cols[keep]
    extra_r, extra_c = [], []
    for i, L in zip(rng.choice(m, 6, replace=False), (70, 130, 400, 700, 900, 1000)):      # long rows of A (<= kTileMaxRow)
        extra_r.append(np.full(L, i)); extra_c.append(rng.choice(n, L, replace=False))
    for j, L in zip(rng.choice(n, 5, replace=False), (90, 300, 650, 800, 990)):             # long columns (= rows of A^T)
        extra_r.append(rng.choice(m, L, replace=False)); extra_c.append(np.full(L, j))
    r_idx = np.concatenate([r_idx] + extra_r); c_idx = np.concatenate([c_idx] + extra_c)
    A = sparse.csr_matrix((np.ones(len(r_idx)), (r_idx, c_idx)), shape=(m, n))
    A.sum_duplicates()
    A.data = rng.normal(size=A.nnz) * 10.0 ** rng.uniform(-1, 1, size=A.nnz)
    A.sort_indices()
    assert np.diff(A.indptr).max() <= 1024 and np.bincount(A.indices, minlength=n).max() <= 1024
    x0 = rng.uniform(0, 1, size=n)
    b = A @ x0
    rp, ci, v = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.copy()
    AL, AU = np.where(rng.random(m) < 0.5, b, -np.inf), b + np.where(rng.random(m) < 0.5, 0.0, 1.0)
    l, u, c = np.zeros(n), np.where(rng.random(n) < 0.3, 2.0, np.inf), rng.normal(size=n)
    model = hprlp.Model.from_csr(m, n, rp, ci, v, AL, AU, l, u, c)
    s = hprlp.Solver(model, hprlp.Parameters(use_presolve=False))
    d = s.describe()
    assert s.info()["tiled"] == 3 and not s.info()["reordered"], d
    assert d.count("0 % of the entries in staged tiles") == 2 and d.count("all-remainder form (k_pb_fused") == 2, d
    ref = O.ScaledLP(m, n, rp, ci, v, AL, AU, l, u, c, O.Params.default())
    s.scale()
    # the Curtis-Reid passes ran through k_pb_fused<CrEpi> on -log|a|: same scaled matrix as the oracle's (device exp / log)
    for name, want in (("A_val", ref.Av), ("AT_val", ref.ATv), ("row_norm", ref.row_norm), ("col_norm", ref.col_norm), ("c", ref.c)):
        np.testing.assert_allclose(s.get(name), want, rtol=1e-11, atol=1e-300, err_msg=name)
    from test_gpu_kernels import adopt_gpu_data
    adopt_gpu_data(s, ref)
    sigma, lam = 0.6, 1.4
    st = run_steps(s, ref, sigma, lam, [(17, True), (3, True), (6, False)])
    for name in NAMES_N + NAMES_M:
        np.testing.assert_allclose(s.get(name), st[name], rtol=1e-11, atol=1e-13, err_msg=name)
    got = s.residuals(150, True)
    sc = ref.sc
    obj_scale = sc.b_scale * sc.c_scale
    pobj = obj_scale * (ref.c @ st["x_bar"])
    ATy = O.spmv(ref.n, ref.ATrp, ref.ATci, ref.ATv, st["y_bar"])
    Ax = O.spmv(ref.m, ref.Arp, ref.Aci, ref.Av, st["x_bar"])
    rd = np.linalg.norm((ref.c - ATy - st["z_bar"]) * ref.col_norm) * sc.c_scale / sc.norm_c_org
    rp_ = np.linalg.norm(np.maximum(np.minimum(ref.AU - Ax, 0.0), ref.AL - Ax) * ref.row_norm) * sc.b_scale / sc.norm_b_org
    assert abs(got["primal_obj"] - pobj) <= 1e-11 * (1 + abs(pobj))
    assert abs(got["err_Rd"] - rd) <= 1e-10 * rd and abs(got["err_Rp"] - rp_) <= 1e-10 * rp_
    lam_g, it = s.power_iteration(max_iter=40)
    lam_ref, it_ref = ref.power_iteration(max_iter=40)
    assert it == it_ref and abs(lam_g - lam_ref) <= 1e-11 * lam_ref
    s.close(); model.free()


def test_permuted_grid_is_reordered_on_the_device(gpu):
    """A five-point grid (large diameter: BFS balls grow polynomially, the Voronoi clustering needs ~100 levels) in random
    numbering: the device clustering + ordering must make it tileable, and the iterates must match the oracle."""
    g = 1300
    m = n = g * g
    idx = np.arange(n).reshape(g, g)
    rows = [idx.ravel()] * 5
    nb = [idx, np.roll(idx, 1, 0), np.roll(idx, -1, 0), np.roll(idx, 1, 1), np.roll(idx, -1, 1)]
    rng = np.random.default_rng(5)
    A = sparse.coo_matrix((rng.uniform(0.5, 1.5, size=5 * n) * rng.choice([-1.0, 1.0], size=5 * n),
                           (np.concatenate(rows), np.concatenate([x.ravel() for x in nb]))), shape=(m, n)).tocsr()
    pr, pc = rng.permutation(m), rng.permutation(n)
    inv_pc = np.empty(n, np.int64); inv_pc[pc] = np.arange(n)
    B = A[pr]
    B = sparse.csr_matrix((B.data, inv_pc[B.indices], B.indptr), shape=(m, n))
    B.sort_indices()
    x0 = rng.uniform(0, 1, size=n)
    b = B @ x0
    AL, AU, l, u, c = b - 1.0, b + 1.0, np.zeros(n), np.full(n, 2.0), rng.normal(size=n)
    rp, ci, v = B.indptr.astype(np.int32), B.indices.astype(np.int32), B.data.copy()
    model = hprlp.Model.from_csr(m, n, rp, ci, v, AL, AU, l, u, c)
    # by default the reordered matrix -- tileable, but five entries per row -- runs the stream kernel (round 5: thin rows); with
    # HPRLP_PIECES_ANYWAY the piece form of the tiled kernel, as until then
    for env, tiled in (({}, 0), ({"HPRLP_PIECES_ANYWAY": "1"}, 3)):
        os.environ.update(env)
        try:
            s = hprlp.Solver(model, hprlp.Parameters(use_presolve=False, use_CR_scaling=False))
        finally:
            for k in env:
                os.environ.pop(k, None)
        info = s.info()
        assert info["reordered"] and info["tiled"] == tiled, info
        assert (s.describe().count("[tiled piece form declined: thin rows]") == 2) == (tiled == 0), s.describe()
        ref = O.ScaledLP(m, n, rp, ci, v, AL, AU, l, u, c, O.Params.default(use_CR_scaling=0))
        s.scale()
        st = run_steps(s, ref, 0.6, 1.4, [(7, True), (4, False)])
        for name in NAMES_N + NAMES_M:
            np.testing.assert_allclose(s.get(name), st[name], rtol=1e-11, atol=1e-12, err_msg=name)
        s.close()
    model.free()


def test_solve_batched_on_a_matrix_the_single_solver_would_reorder(gpu, lp):
    """solve_batched builds its shared matrix with Solver::setup(); that must NOT apply the locality ordering (the panels,
    the scale vectors it downloads and the returned X / Y / Z are all in the caller's numbering).  Two members on the
    permuted 1.6 M x 1.6 M LP, checked by the KKT conditions recomputed here on the LP as given and against single solves
    (which do reorder).  With the ordering applied silently the members would be solved against a different LP."""
    m, n = lp["m"], lp["n"]
    B, tol = 2, 1e-4
    rng = np.random.default_rng(3)
    model = hprlp.Model.from_csr(m, n, lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])
    Cm = np.stack([lp["c"], lp["c"] + 0.05 * rng.uniform(0.0, 1.0, size=n)], axis=1)
    AL = np.repeat(lp["AL"][:, None], B, axis=1)
    AU = np.repeat(lp["AU"][:, None], B, axis=1)
    L = np.repeat(lp["l"][:, None], B, axis=1)
    U = np.where(np.isfinite(lp["u"]), lp["u"], 50.0)
    U = np.repeat(U[:, None], B, axis=1)
    prm = hprlp.Parameters(stop_tol=tol, max_iter=30000, use_presolve=False)
    r = hprlp.solve_batched(model, Cm, AL, AU, L, U, None, prm)
    assert r["status"] == ["OPTIMAL"] * B, r["status"]
    A = sparse.csr_matrix((lp["values"], lp["colind"], lp["rowptr"]), shape=(m, n))
    for k in range(B):
        x, y, z = r["x"][:, k], r["y"][:, k], r["z"][:, k]
        Ax = A @ x
        viol = np.maximum(np.maximum(np.where(np.isfinite(AL[:, k]), AL[:, k] - Ax, 0), np.where(np.isfinite(AU[:, k]), Ax - AU[:, k], 0)), 0)
        b = np.maximum(np.where(np.isfinite(AL[:, k]), np.abs(AL[:, k]), 0), np.where(np.isfinite(AU[:, k]), np.abs(AU[:, k]), 0))
        assert np.linalg.norm(viol) <= 3 * tol * (1 + np.linalg.norm(b)), k
        rd = Cm[:, k] - A.T @ y - z
        assert np.linalg.norm(rd) <= 3 * tol * (1 + np.linalg.norm(Cm[:, k])), k
        assert abs(r["primal_obj"][k] - float(Cm[:, k] @ x)) <= 1e-8 * (1 + abs(r["primal_obj"][k]))
    assert abs(r["primal_obj"][0] - lp["obj_star"]) <= 1e-3 * (1 + abs(lp["obj_star"]))
    mk = hprlp.Model.from_csr(m, n, lp["rowptr"], lp["colind"], lp["values"], AL[:, 1], AU[:, 1], L[:, 1], U[:, 1], Cm[:, 1])
    s1 = mk.solve(prm)  # single-LP path: reorders
    assert s1.status == "OPTIMAL"
    assert abs(s1.primal_obj - r["primal_obj"][1]) <= 10 * tol * (1 + abs(s1.primal_obj))
    mk.free(); model.free()


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_all_remainder_kernel_on_small_random_shapes(gpu, seed):
    """k_pb_fused and the lane-chunk remainder steps (kernels.hip: remainder_steps) on shapes the size thresholds normally keep
    away from them -- a few thousand rows, row lengths from 0 to several hundred, empty rows and columns, sizes that are not
    multiples of anything -- forced by lowering the thresholds (a separate process per case: they are read once).  Iterates over
    mixed normal / check steps, one residual evaluation and lambda_max against the oracle."""
    import subprocess
    import sys
    code = r'''
import os, sys
import numpy as np
from scipy import sparse
sys.path.insert(0, os.path.join(%r, "tests"))
from conftest import hprlp
from oracle import oracle as O
from test_gpu_kernels import NAMES_M, NAMES_N, adopt_gpu_data, run_steps
seed = int(sys.argv[1])
rng = np.random.default_rng(100 + seed)
m = int(rng.integers(2500, 9000)); n = int(rng.integers(2500, 12000))
lens = np.minimum(rng.geometric(0.12, size=m) - 1, n)          # many short rows, zeros included
for i in rng.choice(m, 6, replace=False):
    lens[i] = int(rng.integers(100, min(900, n)))               # a few long ones
rows = np.repeat(np.arange(m), lens)
cols = np.concatenate([rng.choice(n, L, replace=False) for L in lens if L > 0]) if lens.sum() else np.zeros(0, int)
j = int(rng.integers(0, n)); L = int(rng.integers(200, min(1000, m)))
rows = np.concatenate([rows, rng.choice(m, L, replace=False)]); cols = np.concatenate([cols, np.full(L, j)])   # a long column
A = sparse.csr_matrix((np.ones(len(rows)), (rows, cols)), shape=(m, n)); A.sum_duplicates()
A.data = rng.normal(size=A.nnz) * 10.0 ** rng.uniform(-1, 1, size=A.nnz); A.sort_indices()
assert np.diff(A.indptr).max() <= 1024 and np.bincount(A.indices, minlength=n).max() <= 1024
x0 = rng.uniform(0, 1, size=n); b = A @ x0
rp, ci, v = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.copy()
AL, AU = np.where(rng.random(m) < 0.5, b, -np.inf), b + np.where(rng.random(m) < 0.5, 0.0, 1.0)
l, u, c = np.zeros(n), np.where(rng.random(n) < 0.3, 2.0, np.inf), rng.normal(size=n)
model = hprlp.Model.from_csr(m, n, rp, ci, v, AL, AU, l, u, c)
s = hprlp.Solver(model, hprlp.Parameters(use_presolve=False, use_CR_scaling=False))
d = s.describe()
assert d.count("all-remainder form (k_pb_fused") == 2, d
ref = O.ScaledLP(m, n, rp, ci, v, AL, AU, l, u, c, O.Params.default(use_CR_scaling=0))
s.scale(); adopt_gpu_data(s, ref)
st = run_steps(s, ref, 0.6, 1.4, [(9, True), (3, True), (5, False)])
for name in NAMES_N + NAMES_M:
    np.testing.assert_allclose(s.get(name), st[name], rtol=1e-11, atol=1e-13, err_msg=name)
lam_g, it = s.power_iteration(max_iter=30); lam_ref, it_ref = ref.power_iteration(max_iter=30)
assert it == it_ref and abs(lam_g - lam_ref) <= 1e-11 * lam_ref
print("OK", m, n, A.nnz)
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HPRLP_DEVICE_TRANSPOSE_MIN="1000", HPRLP_PB_MIN_COLS="1", HPRLP_PB_MIN_NNZ="1", HPRLP_TILED_MIN_ROWS="1",
               HPRLP_NO_REORDER="1", HPRLP_NO_SMALL="1", HPRLP_TILED_MIN_DENSE="1.01")   # (1.01: the staged-tile form always declines)
    r = subprocess.run([sys.executable, "-c", code, str(seed)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


def test_row_length_rules_keep_skewed_and_unbalanced_patterns_on_the_stream_kernel(gpu):
    """Round 5 (solver.cpp: build_tiled_copy; profiles/r05_form_regret.txt).  Two patterns whose COLUMNS invite the tiled forms and
    whose ROW LENGTHS do not survive them: a graph with hub rows (a third of the entries in rows of more than 256: 'too many
    entries in long rows') and a narrow band with 300 rows of 900 entries behind it (the last block of rows holds 12 x the
    mean: 'unbalanced row blocks').  Both keep the stream kernel, say why, and a band without the coupling rows still takes the
    tiled form; the iterates of the declined form equal those of the forced tiled form to rounding (another summation order)."""
    import bench_helpers as bh
    rng = np.random.default_rng(3)
    m = n = 1_000_000

    def lp_of(A):
        A = sparse.csr_matrix(A); A.sum_duplicates(); A.sort_indices()
        A.data = rng.uniform(0.5, 1.5, size=A.nnz) * rng.choice([-1.0, 1.0], size=A.nnz)
        x0 = rng.uniform(0, 1, A.shape[1]); b = A @ x0
        return dict(m=A.shape[0], n=A.shape[1], rowptr=A.indptr.astype(np.int32), colind=A.indices.astype(np.int32), values=A.data.copy(),
                    AL=b - 0.5, AU=b + 0.5, l=np.zeros(A.shape[1]), u=np.full(A.shape[1], 2.0), c=rng.uniform(-1, 1, A.shape[1]))

    def describe(lp, env=None):
        old = {k: os.environ.get(k) for k in (env or {})}
        os.environ.update(env or {})
        try:
            model = hprlp.Model.from_csr(lp["m"], lp["n"], lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])
            s = hprlp.Solver(model, hprlp.Parameters(use_presolve=False, use_CR_scaling=False))
            d = s.describe()
            s.scale(); s.init(0.7, 1.2); s.iterate(6, True)
            st = {k: s.get(k) for k in ("x", "y", "x_bar", "y_bar")}
            s.close(); model.free()
            return d, st
        finally:
            for k, v in old.items():
                os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)

    # band of 0.8 %, 20 per row: the lowered fused tiled form (control; tools/form_regret.py: band_0.8pct_20)
    r = np.repeat(np.arange(m), 20)
    c = np.abs(r + rng.integers(-4000, 4001, size=len(r))); c = np.where(c > n - 1, 2 * (n - 1) - c, c)
    band = sparse.csr_matrix((np.ones(len(r)), (r, c)), shape=(m, n))
    d0, _ = describe(lp_of(band))
    assert d0.startswith("A: tiled, fused"), d0
    # the same band with 300 rows of 900 entries at the end
    rc = np.repeat(np.arange(m - 300, m), 900); cc = rng.integers(0, n, size=len(rc))
    coupled = band + sparse.csr_matrix((np.ones(len(rc)), (rc, cc)), shape=(m, n))
    lp1 = lp_of(coupled)
    d1, s1 = describe(lp1)
    assert d1.startswith("A: stream kernel") and "unbalanced row blocks" in d1.split("; A^T:")[0], d1
    d1f, s1f = describe(lp1, {"HPRLP_TILED_ANYWAY": "1"})
    assert d1f.startswith("A: tiled"), d1f
    for k in s1:
        np.testing.assert_allclose(s1[k], s1f[k], rtol=1e-9, atol=1e-11, err_msg=k)
    # hub rows: 2000 rows of 300-900 entries hold a third of the entries of an otherwise 4-per-row band
    r = np.repeat(np.arange(m), 4)
    c = np.abs(r + rng.integers(-6000, 6001, size=len(r))); c = np.where(c > n - 1, 2 * (n - 1) - c, c)
    hubs = rng.choice(m, 3500, replace=False)
    lens = rng.integers(300, 900, size=len(hubs))
    rh = np.repeat(hubs, lens); ch = rng.integers(0, n, size=len(rh))
    A = sparse.csr_matrix((np.ones(len(r) + len(rh)), (np.concatenate([r, rh]), np.concatenate([c, ch]))), shape=(m, n))
    d2, _ = describe(lp_of(A))
    assert d2.startswith("A: stream kernel") and "too many entries in long rows" in d2.split("; A^T:")[0], d2


def test_form_regret_tool_on_a_kronecker_graph(gpu, tmp_path):
    """tools/form_regret.py stays runnable, and the pattern with round 4's worst regret stays fixed: a Kronecker (R-MAT) graph of
    2^20 nodes, 7.5e6 entries, 40 % of them in rows of more than 256 -- chosen form = the stream kernel on both matrices, within
    30 % of the best forced form (round 4's rules chose the piece form: 6.5 x the stream kernel's time)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "regret.json")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "form_regret.py"), "--only", "kronecker_20_8", "--steps", "20", "--json", out],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2000:])
    rec = json.load(open(out))["kronecker_20_8"]
    assert rec["forms"]["chosen"]["form"] == "stream/stream", rec["forms"]["chosen"]
    assert rec["regret"] <= 1.3, rec["regret"]
    assert rec["forms"]["tiled_8192"]["it_ms"] > 2.0 * rec["forms"]["chosen"]["it_ms"]      # what the rule avoids


def _popular_columns_lp(m, n, per_row, power, seed):
    """Rows of per_row entries whose columns follow a popularity law c ~ n u^power (set-covering pattern; power 1: uniform)."""
    rng = np.random.default_rng(seed)
    r = np.repeat(np.arange(m), per_row)
    c = np.minimum((n * rng.random(len(r)) ** power).astype(np.int64), n - 1)
    A = sparse.csr_matrix((np.ones(len(r)), (r, c)), shape=(m, n))
    A.sum_duplicates()
    A.sort_indices()
    A.data = rng.normal(size=A.nnz)
    x0 = rng.uniform(0, 1, size=n)
    b = A @ x0
    return A, (m, n, A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.copy(), b - 1.0, b + 1.0, np.zeros(n), np.full(n, 2.0), rng.normal(size=n))


def _iterates(model, steps=12):
    s = hprlp.Solver(model, hprlp.Parameters(use_presolve=False, use_CR_scaling=False))
    d = s.describe()
    s.scale()
    lam, _ = s.power_iteration(max_iter=30)
    s.init(0.7, 1.3 * lam)
    s.iterate(steps, True)
    out = (d, lam, {k: s.get(k) for k in ("x", "y", "x_bar", "y_bar", "z_bar")}, s.residuals(steps + 1, True)["kkt"])
    s.close()
    return out


def test_prepass_work_list_cuts_heavy_source_groups(gpu):
    """Remainder pre-pass (k_far_products / k_far_products_runs, kernels.hip) with the work list of tiled.h f_work: columns with a
    popularity law put a third of the entries into the first source group; its list is cut into chunks, a workgroup each.  Same
    iterates as the stream kernel (reference src/cuda_kernels/HPR_cuda_kernels.cu:203-295) to 1e-10, through the all-remainder
    form with short runs (positions read per entry) and with long runs (run tables); without the list (HPRLP_NO_FAR_WORK) the same again."""
    A, lp = _popular_columns_lp(60_000, 300_000, 24, 3.0, 91)
    model = hprlp.Model.from_csr(*lp)
    pb = {"HPRLP_TILED_MIN_ROWS": "1", "HPRLP_TILED_MIN_DENSE": "1.01", "HPRLP_PB_MIN_COLS": "1", "HPRLP_PB_MIN_NNZ": "1", "HPRLP_DEVICE_TRANSPOSE_MIN": "1",
          "HPRLP_NO_FAR_PUSH": "1"}
    forms = {"short runs (k_far_products)": dict(pb, HPRLP_TILE_ROWS="256"), "long runs (k_far_products_runs)": dict(pb, HPRLP_TILE_ROWS="4096")}

    def under(env):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            return _iterates(model)
        finally:
            for k, v in old.items():
                os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)

    d0, lam0, st0, kkt0 = under({"HPRLP_NO_TILED": "1"})
    assert d0.count("stream kernel") == 2, d0
    for name, env in forms.items():
        d, lam, st, kkt = under(env)
        a_part = d.split("; A^T: ")[0]
        assert "pre-pass work list of" in a_part and "all-remainder form" in a_part, (name, d)       # A's columns are the skewed ones
        assert ("source-side run tables" in a_part) == name.startswith("long"), (name, d)
        assert abs(lam - lam0) <= 1e-11 * abs(lam0), name
        for k in st:
            np.testing.assert_allclose(st[k], st0[k], rtol=1e-10, atol=1e-12, err_msg=f"{name}: {k}")
        assert abs(kkt - kkt0) <= 1e-9 * (1 + abs(kkt0)), name
        d1, lam1, st1, kkt1 = under(dict(env, HPRLP_NO_FAR_WORK="1"))
        assert "pre-pass work list" not in d1, d1
        for k in st:
            np.testing.assert_allclose(st1[k], st0[k], rtol=1e-10, atol=1e-12, err_msg=f"{name}, no work list: {k}")
    model.free()


def test_few_rows_and_random_columns_take_the_all_remainder_form(gpu):
    """Selection (Solver::pb_fallback_wanted, held-out corpus of tools/form_regret.py): a matrix with fewer rows than the staged
    forms ask for whose rows gather at random from millions of columns runs k_pb_fused on 512-row super-blocks, its 3-per-row
    transpose the stream kernel; with popular columns (44 % of the gathers on 2 MB of the vector) it keeps the stream kernel.
    Iterates against the stream kernel's on both."""
    A, lp = _popular_columns_lp(100_000, 2_000_000, 50, 1.0, 92)
    model = hprlp.Model.from_csr(*lp)
    d, lam, st, kkt = _iterates(model, steps=6)
    a_part, at_part = d.split("; A^T: ")[0], d.split("; A^T: ")[1]
    assert "all-remainder form (k_pb_fused" in a_part and "(512 rows" in a_part, d
    assert at_part.startswith("stream kernel"), d
    os.environ["HPRLP_NO_PB_FALLBACK"] = "1"
    try:
        d0, lam0, st0, kkt0 = _iterates(model, steps=6)
    finally:
        os.environ.pop("HPRLP_NO_PB_FALLBACK", None)
    assert "too few rows" in d0.split("; A^T: ")[0], d0
    for k in st:
        np.testing.assert_allclose(st[k], st0[k], rtol=1e-10, atol=1e-12, err_msg=k)
    model.free()
    A, lp = _popular_columns_lp(100_000, 2_000_000, 50, 2.5, 93)
    model = hprlp.Model.from_csr(*lp)
    s = hprlp.Solver(model, hprlp.Parameters(use_presolve=False))
    d = s.describe()
    s.close(); model.free()
    assert d.split("; A^T: ")[0].startswith("A: stream kernel") and "too few rows" in d.split("; A^T: ")[0], d


def test_second_held_out_rules(gpu):
    """DeviceMatrix::build_tiled_copy, second held-out set of tools/form_regret.py.  (a) Two-stage stochastic pattern: 85 % of a
    row's entries in its scenario block, 15 % on 20 000 first-stage columns -- the lowered tiled copy would send the popular share
    through the remainder; it is dropped for the stream kernel.  (b) 50k x 2M with 400 random entries per row (the transpose of a
    10-per-row matrix): fewer rows than a super-block per CU, yet dense full-height tiles -- the piece form, iterates equal to the
    stream kernel's (reference src/cuda_kernels/HPR_cuda_kernels.cu:203-295) to 1e-10."""
    rng = np.random.default_rng(95)
    m, rows, cols, first, per_row = 1_000_000, 500, 700, 20_000, 8
    r = np.repeat(np.arange(m), per_row)
    own = rng.random(len(r)) < 0.85
    c = np.where(own, first + (r // rows) * cols + rng.integers(0, cols, size=len(r)), rng.integers(0, first, size=len(r)))
    n = first + (m // rows) * cols
    A = sparse.csr_matrix((np.ones(len(r)), (r, c)), shape=(m, n)); A.sum_duplicates(); A.sort_indices()
    A.data = rng.normal(size=A.nnz)
    b = A @ rng.uniform(0, 1, size=n)
    model = hprlp.Model.from_csr(m, n, A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.copy(), b - 1.0, b + 1.0, np.zeros(n), np.full(n, 2.0), rng.normal(size=n))
    s = hprlp.Solver(model, hprlp.Parameters(use_presolve=False))
    d = s.describe()
    s.close(); model.free()
    # (15 % on the first-stage columns: the median row's trimmed span stays inside its block, the height is lowered, the copy is built
    # and then dropped; from about 20 % on half of the rows hold two first-stage entries, the height stays full and the one-L2 rule for
    # the piece form says "stream kernel" before the build)
    assert d.startswith("A: stream kernel") and "its remainder gathers from a few popular columns" in d.split("; A^T: ")[0], d

    A, lp = _popular_columns_lp(2_000_000, 50_000, 10, 1.0, 96)
    model = hprlp.Model.from_csr(*lp)
    d, lam, st, kkt = _iterates(model, steps=6)
    at_part = d.split("; A^T: ")[1]
    assert d.startswith("A: stream kernel") and "piece form" in at_part and "7 super-blocks" in at_part, d
    os.environ["HPRLP_NO_TILED"] = "1"
    try:
        d0, lam0, st0, kkt0 = _iterates(model, steps=6)
    finally:
        os.environ.pop("HPRLP_NO_TILED", None)
    assert d0.count("stream kernel") == 2, d0
    assert abs(lam - lam0) <= 1e-11 * abs(lam0)
    for k in st:
        np.testing.assert_allclose(st[k], st0[k], rtol=1e-10, atol=1e-12, err_msg=k)
    model.free()


def test_long_rows_with_unpopular_columns_take_the_all_remainder_form(gpu):
    """Solver::pb_fallback_wanted, rule 13: a b-matching pattern -- node rows with power-law degrees (hubs of tens of thousands of
    entries, a quarter of the entries in rows over 1 024), edge columns of exactly two entries -- is kept off the tiled forms for its
    long rows, yet gathers at random from 3M columns: the all-remainder form with the hubs aside (stream kernel's split rows), its
    two-per-row transpose on the stream kernel.  Iterates against the stream kernel's (HPRLP_NO_PB_LONG_ROWS) to 1e-10; a
    Kronecker graph (popular columns) keeps the stream kernel."""
    rng = np.random.default_rng(97)
    nodes, edges = 300_000, 3_000_000
    w = np.minimum(rng.pareto(1.2, size=nodes) + 1.0, 4000.0)   # (capped: no 4096-row block with more than 1 / 48 of the entries)
    a = np.arange(edges)
    r = np.concatenate([rng.choice(nodes, size=edges, p=w / w.sum()), rng.integers(0, nodes, size=edges)])
    A = sparse.csr_matrix((np.ones(2 * edges), (r, np.concatenate([a, a]))), shape=(nodes, edges)); A.sum_duplicates(); A.sort_indices()
    A.data = rng.normal(size=A.nnz)
    assert np.diff(A.indptr).max() > 4096
    b = A @ rng.uniform(0, 1, size=edges)
    model = hprlp.Model.from_csr(nodes, edges, A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.copy(), b - 1.0, b + 1.0, np.zeros(edges), np.full(edges, 2.0),
                                 rng.normal(size=edges))
    d, lam, st, kkt = _iterates(model, steps=6)
    a_part, at_part = d.split("; A^T: ")
    assert "all-remainder form (k_pb_fused" in a_part and "long rows aside" in a_part and at_part.startswith("stream kernel"), d
    os.environ["HPRLP_NO_PB_LONG_ROWS"] = "1"
    try:
        d0, lam0, st0, kkt0 = _iterates(model, steps=6)
    finally:
        os.environ.pop("HPRLP_NO_PB_LONG_ROWS", None)
    assert d0.count("stream kernel (k_spmv_fused") == 2, d0
    assert abs(lam - lam0) <= 1e-11 * abs(lam0)
    for k in st:
        np.testing.assert_allclose(st[k], st0[k], rtol=1e-10, atol=1e-12, err_msg=k)
    model.free()
