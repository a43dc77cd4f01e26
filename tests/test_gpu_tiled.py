"""The column-tiled fused kernel (hpr-lp-c_amd/csrc/tiled.h) against the oracle.  The tiled path is
normally reserved for matrices with >= 2M rows (one super-block per CU); HPRLP_TILED_MIN_ROWS forces it on a
small banded LP."""
import os

import numpy as np
import pytest

import bench_helpers as bh
from conftest import hprlp
from oracle import oracle as O
from test_gpu_kernels import NAMES_M, NAMES_N, adopt_gpu_data, run_steps

pytestmark = pytest.mark.gpu


@pytest.fixture()
def force_tiled():
    old = {k: os.environ.get(k) for k in ("HPRLP_TILED_MIN_ROWS", "HPRLP_TILED_MIN_DENSE", "HPRLP_NO_TILED")}
    os.environ["HPRLP_TILED_MIN_ROWS"] = "1"
    os.environ["HPRLP_TILED_MIN_DENSE"] = "0.0"
    os.environ.pop("HPRLP_NO_TILED", None)
    yield
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def build(m, n, per_row, band):
    lp = bh.banded_lp(m, n, per_row, band)
    model = hprlp.Model.from_csr(m, n, lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])
    return lp, model


def test_tiled_iterations_match_oracle(gpu, force_tiled):
    m = n = 30000
    lp, model = build(m, n, 12, 600)
    s = hprlp.Solver(model, hprlp.Parameters(use_presolve=False, use_CR_scaling=False))
    ref = O.ScaledLP(m, n, lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"],
                     O.Params.default(use_CR_scaling=0))
    s.scale()
    adopt_gpu_data(s, ref)
    st = run_steps(s, ref, 0.6, 1.4, [(23, True), (5, True), (11, False)])
    for name in NAMES_N + NAMES_M:
        # rows with entries outside the staged tiles (5 % far columns) add those last: rounding only
        np.testing.assert_allclose(s.get(name), st[name], rtol=1e-11, atol=1e-13, err_msg=name)
    lam, it = s.power_iteration()
    lam_ref, it_ref = ref.power_iteration()
    assert it == it_ref and abs(lam - lam_ref) <= 1e-11 * lam_ref
    s.close(); model.free()


def test_tiled_equals_stream_kernel_end_to_end(gpu, force_tiled):
    """Same LP solved with the tiled and with the stream kernel: same iteration count, same optimum."""
    m = n = 20000
    lp, model = build(m, n, 10, 300)
    prm = hprlp.Parameters(stop_tol=1e-4, use_presolve=False, max_iter=3000)
    r_tiled = model.solve(prm)
    os.environ["HPRLP_NO_TILED"] = "1"
    r_stream = model.solve(prm)
    assert r_tiled.status == r_stream.status
    assert abs(r_tiled.primal_obj - r_stream.primal_obj) <= 1e-5 * (1 + abs(r_stream.primal_obj))
    assert abs(r_tiled.iter - r_stream.iter) <= 0.2 * r_stream.iter + 150
    if r_stream.status == "OPTIMAL":
        assert abs(r_tiled.primal_obj - lp["obj_star"]) <= 1e-4 * (1 + abs(lp["obj_star"]))
    model.free()


def test_tiled_handles_long_segments_and_ragged_edges(gpu, force_tiled):
    """Narrow band: a row has all its entries in one tile (segments > 4 go to the remainder list);
    sizes that are not multiples of the super-block / tile sizes."""
    m, n = 9001, 2500
    lp, model = build(m, n, 9, 40)
    s = hprlp.Solver(model, hprlp.Parameters(use_presolve=False, use_CR_scaling=False))
    ref = O.ScaledLP(m, n, lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"],
                     O.Params.default(use_CR_scaling=0))
    s.scale()
    adopt_gpu_data(s, ref)
    st = run_steps(s, ref, 0.9, 1.1, [(9, True)])
    for name in NAMES_N + NAMES_M:
        np.testing.assert_allclose(s.get(name), st[name], rtol=1e-11, atol=1e-13, err_msg=name)
    got = s.residuals(10, True)
    assert np.isfinite(got["kkt"])
    s.close(); model.free()


def test_tiled_runs_are_bit_reproducible(gpu, force_tiled):
    """The rotated sweeps and the persistent schedule fix the summation order by the matrix alone: two solvers on the
    same model produce identical bits, whatever order the workgroups happen to run in."""
    m = n = 40000
    lp, model = build(m, n, 12, 900)
    states = []
    for _ in range(2):
        s = hprlp.Solver(model, hprlp.Parameters(use_presolve=False))
        assert s.info()["tiled"] == 3
        s.scale()
        lam, it = s.power_iteration()
        s.init(-1.0, lam * 1.01)
        s.iterate(60, True)
        states.append((lam, it, {k: s.get(k) for k in ("x", "y", "x_bar", "y_bar", "z_bar")}))
        s.close()
    assert states[0][0] == states[1][0] and states[0][1] == states[1][1]
    for k in states[0][2]:
        assert np.array_equal(states[0][2][k], states[1][2][k]), k
    model.free()


@pytest.mark.parametrize("pieces", [3, 7, 64])
def test_piece_form_matches_oracle(gpu, force_tiled, pieces):
    """Matrices with fewer super-blocks than workgroup slots run the piece form: the tile steps of all super-blocks, end
    to end, cut into equal pieces (3: a piece spans super-blocks; 64: several pieces per super-block), one workgroup
    each, partial row sums added by a finish kernel that runs the epilogue (kernels.hip: k_tiled_part /
    k_tiled_finish).  Same numbers as the oracle to rounding, check variants and reductions included."""
    os.environ["HPRLP_TILE_PIECES"] = str(pieces)
    try:
        m = n = 30000
        lp, model = build(m, n, 12, 600)
        s = hprlp.Solver(model, hprlp.Parameters(use_presolve=False, use_CR_scaling=False))
        assert s.info()["tiled"] == 3
        ref = O.ScaledLP(m, n, lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"],
                         O.Params.default(use_CR_scaling=0))
        s.scale()
        adopt_gpu_data(s, ref)
        st = run_steps(s, ref, 0.6, 1.4, [(23, True), (5, True), (11, False)])
        for name in NAMES_N + NAMES_M:
            np.testing.assert_allclose(s.get(name), st[name], rtol=1e-11, atol=1e-13, err_msg=name)
        got = s.residuals(40, True)
        assert np.isfinite(got["kkt"])
        lam, it = s.power_iteration()
        lam_ref, it_ref = ref.power_iteration()
        assert it == it_ref and abs(lam - lam_ref) <= 1e-11 * lam_ref
        s.close()
        r = model.solve(hprlp.Parameters(stop_tol=1e-4, use_presolve=False, max_iter=3000))
        if r.status == "OPTIMAL":
            assert abs(r.primal_obj - lp["obj_star"]) <= 1e-4 * (1 + abs(lp["obj_star"]))
        model.free()
    finally:
        os.environ.pop("HPRLP_TILE_PIECES", None)


def test_tiled_matrix_with_long_rows(gpu, force_tiled):
    """Rows and columns with hundreds to thousands of entries in a tiled matrix.  Up to kTileMaxRow = 1024 entries a row stays in
    the tiled copy: all but four entries per tile go to the remainder list, where a step that holds a long run of one row is
    added in two levels (tiled.h: kTileRemRun).  Longer rows (a few of them) are left out of the tiled copy and summed by the
    stream kernel's vector / split-row mode into the base vector every tiled launch adds (TiledDev::side_*; 9000 entries: split
    into chunks).  Parity with the oracle as for every tiled matrix; with too many long rows the matrix keeps the stream kernel."""
    from scipy import sparse
    m, n = 26000, 34000
    lp = bh.banded_lp(m, n, 10, 500)
    A0 = sparse.csr_matrix((lp["values"], lp["colind"], lp["rowptr"]), shape=(m, n))
    rng = np.random.default_rng(9)

    def with_long(rows_of_A, rows_of_AT):
        add_r, add_c, add_v = [], [], []
        for i, L in zip(rng.choice(m, len(rows_of_A), replace=False), rows_of_A):
            c = rng.choice(n, L, replace=False)
            add_r.append(np.full(L, i)); add_c.append(c); add_v.append(rng.normal(size=L) * 0.05)
        for j, L in zip(rng.choice(n, len(rows_of_AT), replace=False), rows_of_AT):
            r = rng.choice(m, L, replace=False)
            add_r.append(r); add_c.append(np.full(L, j)); add_v.append(rng.normal(size=L) * 0.05)
        A = (A0 + sparse.csr_matrix((np.concatenate(add_v), (np.concatenate(add_r), np.concatenate(add_c))), shape=(m, n))).tocsr()
        A.sort_indices()
        return A

    for rows_of_A, rows_of_AT in (((300, 500, 700, 950), (250, 600, 900)),             # inside the tiled copy (two-level remainder)
                                  ((700, 1500, 3100, 9000), (900, 2500, 5000, 4097))):  # aside: vector mode and split rows
        A = with_long(rows_of_A, rows_of_AT)
        x0 = np.abs(rng.normal(size=n))
        b = A @ x0
        rp, ci, v = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.copy()
        AL, AU, l, u, c = b - 1.0, b + 1.0, np.zeros(n), np.full(n, 10.0), rng.normal(size=n)
        model = hprlp.Model.from_csr(m, n, rp, ci, v, AL, AU, l, u, c)
        s = hprlp.Solver(model, hprlp.Parameters(use_presolve=False, use_CR_scaling=False))
        assert s.info()["tiled"] == 3
        ref = O.ScaledLP(m, n, rp, ci, v, AL, AU, l, u, c, O.Params.default(use_CR_scaling=0))
        s.scale()
        adopt_gpu_data(s, ref)
        st = run_steps(s, ref, 0.6, 1.4, [(9, True), (4, True), (6, False)])
        for name in NAMES_N + NAMES_M:
            np.testing.assert_allclose(s.get(name), st[name], rtol=1e-10, atol=1e-12, err_msg=name)
        s.close(); model.free()
    # more long rows than the side list takes (64, or 0.1 % of the rows): not tiled
    A2 = with_long(tuple([1100] * 70), ())
    x0 = np.abs(rng.normal(size=n))
    b = A2 @ x0
    model2 = hprlp.Model.from_csr(m, n, A2.indptr.astype(np.int32), A2.indices.astype(np.int32), A2.data, b - 1.0, b + 1.0, np.zeros(n), np.full(n, 10.0),
                                  rng.normal(size=n))
    s2 = hprlp.Solver(model2, hprlp.Parameters(use_presolve=False))
    assert s2.info()["tiled"] & 1 == 0
    s2.close(); model2.free()


def test_multi_round_persistent_schedule_matches_oracle(gpu, force_tiled):
    """The HEADLINE kernel form under the oracle.  Config 5 runs k_tiled_fused with 1221 super-blocks over 512 persistent
    workgroups: 2.38 rounds per slot (workgroup `slot` takes super-blocks slot, slot + 64, ... of its XCD's range,
    tiled_build.hip: finish_schedule), XCD cohorts, rotated sweeps, hand-off of the remainder products between the half-steps
    and the x-rebuild mode inside a run of normal iterations.  Same form on an LP the oracle handles in seconds: 64-row
    super-blocks on a 100 k x 100 k banded LP = 1563 super-blocks, 3.05 rounds per slot, the last round partial.  All iterate
    vectors of reference src/cuda_kernels/HPR_cuda_kernels.cu:203-295 against the oracle at 1e-11 (per-row summation order:
    rotated tile sweep, remainder last), plus one residual evaluation (src/main_iterate.cu:229-309) and lambda_max."""
    old = {k: os.environ.get(k) for k in ("HPRLP_TILE_ROWS", "HPRLP_TILE_PIECES", "HPRLP_NO_FAR_PUSH", "HPRLP_STORE_X")}
    os.environ["HPRLP_TILE_ROWS"] = "64"
    for k in ("HPRLP_TILE_PIECES", "HPRLP_NO_FAR_PUSH", "HPRLP_STORE_X"):
        os.environ.pop(k, None)
    try:
        m = n = 100_000
        lp, model = build(m, n, 12, 1500)
        s = hprlp.Solver(model, hprlp.Parameters(use_presolve=False, use_CR_scaling=False))
        assert s.info()["tiled"] == 3
        d = s.describe()
        nsb = -(-m // 64)
        assert nsb > 3 * 512 and f"{nsb} super-blocks" in d and "tiled, fused" in d and "piece form" not in d, d
        ref = O.ScaledLP(m, n, lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"],
                         O.Params.default(use_CR_scaling=0))
        s.scale()
        adopt_gpu_data(s, ref)
        sigma, lam = 0.6, 1.4
        st = run_steps(s, ref, sigma, lam, [(23, True), (5, True), (11, False)])
        for name in NAMES_N + NAMES_M:
            np.testing.assert_allclose(s.get(name), st[name], rtol=1e-11, atol=1e-13, err_msg=name)
        # one residual evaluation on the state of the last CHECK step (x_bar, y_bar, z_bar, y_obj, x_temp, y_temp are untouched
        # by the 11 normal iterations that followed): oracle formulas in numpy on the oracle's state
        got = s.residuals(150, True)
        sc = ref.sc
        obj_scale = sc.b_scale * sc.c_scale
        pobj = obj_scale * (ref.c @ st["x_bar"])
        dobj = obj_scale * (st["y_obj"] @ st["y_bar"] + st["x_bar"] @ st["z_bar"])
        ATy = O.spmv(ref.n, ref.ATrp, ref.ATci, ref.ATv, st["y_bar"])
        Ax = O.spmv(ref.m, ref.Arp, ref.Aci, ref.Av, st["x_bar"])
        rd = np.linalg.norm((ref.c - ATy - st["z_bar"]) * ref.col_norm) * sc.c_scale / sc.norm_c_org
        rp = np.linalg.norm(np.maximum(np.minimum(ref.AU - Ax, 0.0), ref.AL - Ax) * ref.row_norm) * sc.b_scale / sc.norm_b_org
        Adx = O.spmv(ref.m, ref.Arp, ref.Aci, ref.Av, st["x_temp"])
        wn = np.sqrt(sigma * lam * (st["y_temp"] @ st["y_temp"]) + (st["x_temp"] @ st["x_temp"]) / sigma + 2 * (Adx @ st["y_temp"]))
        assert abs(got["primal_obj"] - pobj) <= 1e-11 * (1 + abs(pobj))
        assert abs(got["dual_obj"] - dobj) <= 1e-11 * (1 + abs(dobj))
        assert abs(got["err_Rd"] - rd) <= 1e-10 * rd and abs(got["err_Rp"] - rp) <= 1e-10 * rp
        assert abs(got["weighted_norm"] - wn) <= 1e-9 * wn
        lam_g, it = s.power_iteration()
        lam_ref, it_ref = ref.power_iteration()
        assert it == it_ref and abs(lam_g - lam_ref) <= 1e-11 * lam_ref
        s.close(); model.free()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("rows_sb,pieces,tile_cols", [(1984, None, 2048), (3008, None, 2048), (2048, 5, 2048), (1984, None, 1024), (8192, 7, 1024)])
def test_lowered_super_block_height_matches_oracle(gpu, force_tiled, rows_sb, pieces, tile_cols):
    """Super-blocks of fewer than 8192 rows (tiled.h: one super-block per workgroup slot for mid-size matrices; any multiple of
    64): fused form, hand-off between the half-steps (a source group of one matrix' remainder = a super-block of the other)
    and the piece form on top of it -- same iterates as the oracle, device and host builders equal array for array.  Round 4:
    the same with tiles of 1024 columns (tiled.h: kTileColsNarrow, what a narrow band gets from Solver::choose_sb_rows)."""
    old = {k: os.environ.get(k) for k in ("HPRLP_TILE_ROWS", "HPRLP_TILE_PIECES", "HPRLP_TILING_CHECK", "HPRLP_TILE_COLS")}
    os.environ["HPRLP_TILE_COLS"] = str(tile_cols)
    os.environ["HPRLP_TILE_ROWS"] = str(rows_sb)
    os.environ["HPRLP_TILING_CHECK"] = "1"  # (the host builder needs host column indices: both matrices have them below 4 M entries)
    os.environ["HPRLP_TILE_PIECES"] = str(pieces or 0)  # (0: the fused form although the test matrix has few super-blocks)
    try:
        m, n = 30011, 26000
        lp, model = build(m, n, 12, 600)
        s = hprlp.Solver(model, hprlp.Parameters(use_presolve=False, use_CR_scaling=False))
        assert s.info()["tiled"] == 3
        d = s.describe()
        assert f"{-(-m // rows_sb)} super-blocks" in d and f"{-(-n // rows_sb)} super-blocks" in d, d
        assert ("piece form" in d) == bool(pieces), d
        assert ("tiles of 1024 columns" in d) == (tile_cols == 1024), d
        ref = O.ScaledLP(m, n, lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"],
                         O.Params.default(use_CR_scaling=0))
        s.scale()
        adopt_gpu_data(s, ref)
        st = run_steps(s, ref, 0.6, 1.4, [(23, True), (5, True), (11, False)])
        for name in NAMES_N + NAMES_M:
            np.testing.assert_allclose(s.get(name), st[name], rtol=1e-11, atol=1e-13, err_msg=name)
        lam, it = s.power_iteration()
        lam_ref, it_ref = ref.power_iteration()
        assert it == it_ref and abs(lam - lam_ref) <= 1e-11 * lam_ref
        s.close(); model.free()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_mid_size_banded_lp_gets_one_super_block_per_slot(gpu):
    """No environment overrides: a 1M x 1M LP with a window of 2e4 columns has 123 full-height super-blocks -- fewer than the
    chip's 512 workgroup slots.  Solver::choose_sb_rows lowers the height so that there are at most 512 and at least 384,
    the half-steps run the FUSED tiled kernel (one launch each), and the whole solve reaches the planted optimum in the
    same number of iterations as with the stream kernel (+- the usual fork at a thresholded restart decision)."""
    m = n = 1_000_000
    lp = bh.banded_lp(m, n, 20, 10_000)
    model = hprlp.Model.from_csr(m, n, lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])
    s = hprlp.Solver(model, hprlp.Parameters(use_presolve=False))
    d = s.describe()
    s.close()
    assert "tiled, fused" in d and "piece form" not in d and "stream kernel" not in d, d
    assert d.count("tiles of 1024 columns") == 2, d   # ~2 entries of a row per 2048-column tile: narrow tiles (choose_sb_rows)
    import re
    nsb = [int(x) for x in re.findall(r"(\d+) super-blocks", d)]
    assert len(nsb) == 2 and all(384 <= k <= 512 for k in nsb), d
    prm = hprlp.Parameters(stop_tol=1e-4, use_presolve=False, max_iter=20000)
    r = model.solve(prm)
    os.environ["HPRLP_NO_TILED"] = "1"
    try:
        r0 = model.solve(prm)
    finally:
        os.environ.pop("HPRLP_NO_TILED", None)
    assert r.status == r0.status == "OPTIMAL"
    assert abs(r.iter - r0.iter) <= 0.1 * r0.iter + 150, (r.iter, r0.iter)
    assert abs(r.primal_obj - lp["obj_star"]) <= 1e-3 * (1 + abs(lp["obj_star"]))
    assert abs(r.primal_obj - r0.primal_obj) <= 1e-4 * (1 + abs(r0.primal_obj))
    model.free()


def test_lowered_heights_differ_between_a_and_its_transpose(gpu):
    """m != n in the mid-size regime: A and A^T get different super-block heights (rows / 512 each); the source groups of one
    matrix' remainder lists are the super-blocks of the other, so the hand-off between the half-steps still applies.  Iterates,
    residuals and lambda_max against the stream kernel on the same LP."""
    m, n = 800_000, 1_100_000
    lp = bh.banded_lp(m, n, 16, 8_000)
    model = hprlp.Model.from_csr(m, n, lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])

    def run():
        s = hprlp.Solver(model, hprlp.Parameters(use_presolve=False))
        d = s.describe()
        s.scale()
        lam, it = s.power_iteration(max_iter=40)
        s.init(0.8, 1.2 * lam)
        s.iterate(25, True)
        s.iterate(6, False)
        out = (d, lam, {k: s.get(k) for k in ("x", "y", "x_hat", "x_bar", "y_bar", "z_bar")}, s.residuals(33, True))
        s.close()
        return out

    d, lam, st, res = run()
    import re
    nsb = [int(x) for x in re.findall(r"(\d+) super-blocks", d)]
    assert "tiled, fused" in d and "stream kernel" not in d and len(nsb) == 2 and all(384 <= k <= 512 for k in nsb), d
    os.environ["HPRLP_NO_TILED"] = "1"
    try:
        d0, lam0, st0, res0 = run()
    finally:
        os.environ.pop("HPRLP_NO_TILED", None)
    assert "tiled" not in d0.replace("tiled form not attempted", "")
    assert abs(lam - lam0) <= 1e-11 * abs(lam0)
    for k in st:
        np.testing.assert_allclose(st[k], st0[k], rtol=1e-10, atol=1e-12, err_msg=k)
    assert abs(res["kkt"] - res0["kkt"]) <= 1e-9 * (1 + abs(res0["kkt"]))
    model.free()


@pytest.mark.parametrize("rows_sb", [0, 1024])
def test_curtis_reid_passes_through_the_tiled_kernel_match_the_oracle(gpu, force_tiled, rows_sb):
    """The 40 Curtis-Reid passes of scale() (reference src/scaling.cu:5-38, 40-83) run through the tiled kernel on the copy's
    -log|a| values (NaN marks padding, so an explicitly stored zero keeps the reference's -log(1e-300) term): scaled matrix,
    bounds and norms against the oracle's scaling, and against the stream-kernel passes (HPRLP_NO_TILED_CR=1)."""
    m, n = 9000, 12000
    lp, model0 = build(m, n, 8, 400)
    model0.free()
    vals = lp["values"].copy()
    vals[[5, 777, 40001]] = 0.0   # explicitly stored zeros
    model = hprlp.Model.from_csr(m, n, lp["rowptr"], lp["colind"], vals, lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])
    old = {k: os.environ.get(k) for k in ("HPRLP_TILE_ROWS", "HPRLP_TILE_PIECES", "HPRLP_NO_TILED_CR")}
    if rows_sb:
        os.environ["HPRLP_TILE_ROWS"] = str(rows_sb)
    os.environ["HPRLP_TILE_PIECES"] = "0"
    try:
        got = {}
        for mode in ("tiled", "stream"):
            if mode == "stream":
                os.environ["HPRLP_NO_TILED_CR"] = "1"
            s = hprlp.Solver(model, hprlp.Parameters(use_presolve=False))
            assert s.info()["tiled"] & 3 == 3
            s.scale()
            got[mode] = {k: s.get(k) for k in ("A_val", "AT_val", "AL", "AU", "l", "u", "c", "row_norm", "col_norm")}
            s.close()
        ref = O.ScaledLP(m, n, lp["rowptr"], lp["colind"], vals, lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"], O.Params.default())
        want = {"A_val": ref.Av, "AT_val": ref.ATv, "AL": ref.AL, "AU": ref.AU, "l": ref.l, "u": ref.u, "c": ref.c,
                "row_norm": ref.row_norm, "col_norm": ref.col_norm}
        for k, w in want.items():
            fin = np.isfinite(w)
            assert np.array_equal(fin, np.isfinite(got["tiled"][k])), k
            np.testing.assert_allclose(got["tiled"][k][fin], w[fin], rtol=1e-11, atol=1e-300, err_msg=k)
            np.testing.assert_allclose(got["stream"][k][fin], w[fin], rtol=1e-11, atol=1e-300, err_msg=k)
        assert not np.array_equal(got["tiled"]["row_norm"], np.ones(m))
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        model.free()
