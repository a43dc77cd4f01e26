"""solve_batched parity: B LPs sharing A against the oracle's restatement of reference
src/batched_solver.cu, and against the golden optima (GPU)."""
import json
import os

import numpy as np
import pytest

from conftest import hprlp, lpgen
from oracle import oracle as O

pytestmark = pytest.mark.gpu
INF = np.inf
HERE = os.path.dirname(os.path.abspath(__file__))


def test_example_batch_matches_golden(gpu):
    """The three LPs of reference examples/c/example_batched_lp.c:37-62."""
    g = json.load(open(os.path.join(HERE, "golden", "known_lps.json")))
    B = len(g)
    model = hprlp.Model.from_csr(2, 2, g[0]["rowptr"], g[0]["colind"], [float(v) for v in g[0]["values"]],
                                 [-INF, -INF], [10, 12], [0, 0], [INF, INF], [-3, -5])
    Cm = np.array([[float(v) for v in c["c"]] for c in g]).T
    AU = np.array([[float(v) for v in c["AU"]] for c in g]).T
    U = np.array([[float(v) for v in c["u"]] for c in g]).T
    AL = np.full((2, B), -INF); L = np.zeros((2, B))
    prm = hprlp.Parameters(stop_tol=1e-8, max_iter=200000, use_presolve=False)
    r = hprlp.solve_batched(model, Cm, AL, AU, L, U, [0.0] * B, prm)
    ref = O.solve_batched(2, 2, g[0]["rowptr"], g[0]["colind"], [float(v) for v in g[0]["values"]], B,
                          Cm.T.ravel(), AL.T.ravel(), AU.T.ravel(), L.T.ravel(), U.T.ravel(), [0.0] * B,
                          params=O.Params.default(stop_tol=1e-8, max_iter=200000))
    assert r["status"] == ["OPTIMAL"] * B == ref["status"]
    assert list(r["iter"]) == list(ref["iter"])
    for k, c in enumerate(g):
        assert abs(r["primal_obj"][k] - c["obj"]) < 1e-6
        np.testing.assert_allclose(r["x"][:, k], c["x"], atol=1e-6)
        np.testing.assert_allclose(r["y"][:, k], c["y"], atol=1e-6)
        np.testing.assert_allclose(r["x"][:, k], ref["x"][k], atol=1e-9)
    assert r["time"] == pytest.approx(r["setup_time"] + r["solve_time"])
    model.free()


def make_batch(lp, B, seed):
    """Perturbed copies of one planted LP (BASELINE config 4 recipe): c_k = c(1+0.1 N), AU_k = AU + |N(0,0.1)|."""
    rng = np.random.default_rng(seed)
    m, n = lp["m"], lp["n"]
    Cm = lp["c"][:, None] * (1 + 0.1 * rng.normal(size=(n, B)))
    AU = lp["AU"][:, None] + np.abs(rng.normal(scale=0.1, size=(m, B)))
    AL = np.repeat(lp["AL"][:, None], B, axis=1)
    AL = np.where(np.isfinite(AL), np.minimum(AL, AU), AL)
    L = np.repeat(lp["l"][:, None], B, axis=1)
    U = np.repeat(lp["u"][:, None], B, axis=1)
    U = np.where(np.isfinite(U), U, 50.0)         # keep every member bounded
    return Cm, AL, AU, L, U


@pytest.mark.parametrize("B", [5, 64, 70])
def test_planted_batch_matches_oracle(gpu, B):
    lp = lpgen.planted_lp(120, 200, 1300, 40 + B)
    Cm, AL, AU, L, U = make_batch(lp, B, B)
    model = hprlp.Model.from_csr(lp["m"], lp["n"], lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"],
                                 lp["l"], lp["u"], lp["c"])
    tol = 1e-6
    prm = hprlp.Parameters(stop_tol=tol, max_iter=60000, use_presolve=False)
    r = hprlp.solve_batched(model, Cm, AL, AU, L, U, None, prm)
    ref = O.solve_batched(lp["m"], lp["n"], lp["rowptr"], lp["colind"], lp["values"], B, Cm.T.ravel(), AL.T.ravel(),
                          AU.T.ravel(), L.T.ravel(), U.T.ravel(), None,
                          params=O.Params.default(stop_tol=tol, max_iter=60000))
    assert r["batch_size"] == B and r["x"].shape == (lp["n"], B) and r["y"].shape == (lp["m"], B)
    assert r["status"] == ref["status"]
    done = [k for k in range(B) if ref["status"][k] == "OPTIMAL"]
    assert len(done) >= B // 2
    # same schedule and same per-problem decisions => same stopping iteration for (nearly) every member;
    # a member may fork at a thresholded restart decision (FP64 reduction order)
    same_iter = sum(int(r["iter"][k] == ref["iter"][k]) for k in done)
    assert same_iter >= 0.8 * len(done)
    for k in done:
        assert abs(r["primal_obj"][k] - ref["primal_obj"][k]) <= 20 * tol * (1 + abs(ref["primal_obj"][k]))
        assert r["residuals"][k] <= tol
    model.free()


@pytest.mark.parametrize("B,chunk", [(64, 8), (64, 16), (70, 32), (70, 8), (24, 0)])
def test_chunk_widths_of_the_panels_give_the_same_solves(gpu, B, chunk):
    """The device panels are stored chunk-major (hpr-lp-c_amd/csrc/batched.hip: layout); HPRLP_BATCH_CHUNK forces chunks of
    8 / 16 / 32 problems where the default is 64 (kb_halfN instead of kb_half64; B = 24 -> one chunk of 32 by itself).  Every
    row sum is added in CSR order by every kernel, so members follow the same trajectory: same status and stopping iteration
    as the oracle (up to forks at thresholded restart decisions), same optimum."""
    lp = lpgen.planted_lp(150, 260, 1700, 70 + B)
    Cm, AL, AU, L, U = make_batch(lp, B, B + 1)
    model = hprlp.Model.from_csr(lp["m"], lp["n"], lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"],
                                 lp["l"], lp["u"], lp["c"])
    tol = 1e-6
    prm = hprlp.Parameters(stop_tol=tol, max_iter=60000, use_presolve=False)
    old = os.environ.get("HPRLP_BATCH_CHUNK")
    try:
        if chunk:
            os.environ["HPRLP_BATCH_CHUNK"] = str(chunk)
        r = hprlp.solve_batched(model, Cm, AL, AU, L, U, None, prm)
    finally:
        os.environ.pop("HPRLP_BATCH_CHUNK", None)
        if old is not None:
            os.environ["HPRLP_BATCH_CHUNK"] = old
    ref = O.solve_batched(lp["m"], lp["n"], lp["rowptr"], lp["colind"], lp["values"], B, Cm.T.ravel(), AL.T.ravel(),
                          AU.T.ravel(), L.T.ravel(), U.T.ravel(), None,
                          params=O.Params.default(stop_tol=tol, max_iter=60000))
    assert r["status"] == ref["status"]
    done = [k for k in range(B) if ref["status"][k] == "OPTIMAL"]
    assert len(done) >= B // 2
    assert sum(int(r["iter"][k] == ref["iter"][k]) for k in done) >= 0.8 * len(done)
    for k in done:
        assert abs(r["primal_obj"][k] - ref["primal_obj"][k]) <= 20 * tol * (1 + abs(ref["primal_obj"][k]))
        assert r["residuals"][k] <= tol
        if r["iter"][k] == ref["iter"][k]:
            np.testing.assert_allclose(r["x"][:, k], ref["x"][k], rtol=1e-7, atol=1e-8)
    model.free()


def test_batched_bad_arguments(gpu, model_mps_arrays):
    a = model_mps_arrays
    model = hprlp.Model.from_csr(a["m"], a["n"], a["rowptr"], a["colind"], a["values"], a["AL"], a["AU"], a["l"], a["u"], a["c"])
    L = hprlp.lib()
    res = L.solve_batched(model._ptr, 2, None, None, None, None, None, None, None)   # NULL panels -> "ERROR" per member
    assert res.batch_size == 2 and not res.x
    raw = bytes(bytearray(res.status[i][0] if isinstance(res.status[i], bytes) else res.status[i] for i in range(128)))
    assert raw[:5] == b"ERROR" and raw[64:69] == b"ERROR"
    import ctypes as C
    L.free_batched_results(C.byref(res))
    assert not res.status and res.batch_size == 0
    model.free()


def test_config4_full_size_against_oracle_and_single_solves(gpu):
    """BASELINE config 4 at its real size: the config-3 matrix (33 874 x 105 728), B = 64 perturbed members (bounded:
    the recipe's c perturbation makes members with infinite upper bounds unbounded), tolerance 1e-4.
    (i) 8 sampled members against the oracle's solve_batched (reference src/batched_solver.cu:1017-1084 restated) run on
    exactly those 8 -- members only share A and lambda_max, so a member's trajectory does not depend on the batch it is
    in: same status, same stopping iteration (a member may fork at a thresholded restart decision), same objective;
    (ii) all 64 against the single-LP path (HPRLP_main_solve, different scaling => different iteration counts): same
    optimum within the tolerance, and the batched result passes the KKT conditions recomputed here on the unscaled LP."""
    from scipy import sparse
    lp = lpgen.c3_pds20_like()
    B, tol = 64, 1e-4
    Cm, AL, AU, L, U = make_batch(lp, B, 4)
    m, n = lp["m"], lp["n"]
    model = hprlp.Model.from_csr(m, n, lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])
    prm = hprlp.Parameters(stop_tol=tol, max_iter=60000, use_presolve=False)
    r = hprlp.solve_batched(model, Cm, AL, AU, L, U, None, prm)
    assert r["status"] == ["OPTIMAL"] * B
    assert np.isfinite(r["x"]).all() and np.isfinite(r["y"]).all() and np.isfinite(r["z"]).all()
    assert (np.asarray(r["residuals"]) <= tol).all()

    # (i) oracle on 8 members
    sel = list(range(0, B, 9))[:8]
    ref = O.solve_batched(m, n, lp["rowptr"], lp["colind"], lp["values"], len(sel), Cm[:, sel].T.ravel(), AL[:, sel].T.ravel(),
                          AU[:, sel].T.ravel(), L[:, sel].T.ravel(), U[:, sel].T.ravel(), None,
                          params=O.Params.default(stop_tol=tol, max_iter=60000))
    assert ref["status"] == ["OPTIMAL"] * len(sel)
    same = 0
    for q, k in enumerate(sel):
        same += int(r["iter"][k] == ref["iter"][q])
        assert abs(r["iter"][k] - ref["iter"][q]) <= 0.1 * ref["iter"][q] + 150, (k, r["iter"][k], ref["iter"][q])
        assert abs(r["primal_obj"][k] - ref["primal_obj"][q]) <= 10 * tol * (1 + abs(ref["primal_obj"][q]))
    assert same >= 6, (same, [r["iter"][k] for k in sel], list(ref["iter"]))

    # (ii) KKT of every member on the unscaled LP, and the single-LP path on every member
    A = sparse.csr_matrix((lp["values"], lp["colind"], lp["rowptr"]), shape=(m, n))
    for k in range(B):
        x, y, z = r["x"][:, k], r["y"][:, k], r["z"][:, k]
        Ax = A @ x
        rp = np.maximum(np.maximum(AL[:, k] - Ax, Ax - AU[:, k]), 0.0)
        rp = np.where(np.isfinite(rp), rp, 0.0)
        bnd = np.maximum(np.maximum(L[:, k] - x, x - U[:, k]), 0.0)
        b = np.maximum(np.where(np.isfinite(AL[:, k]), np.abs(AL[:, k]), 0.0), np.where(np.isfinite(AU[:, k]), np.abs(AU[:, k]), 0.0))
        assert np.linalg.norm(rp) <= 3 * tol * (1 + np.linalg.norm(b)), k
        assert np.linalg.norm(bnd) <= 3 * tol * (1 + np.linalg.norm(b)), k
        rd = Cm[:, k] - A.T @ y - z
        assert np.linalg.norm(rd) <= 3 * tol * (1 + np.linalg.norm(Cm[:, k])), k
        assert abs(r["primal_obj"][k] - float(Cm[:, k] @ x)) <= 1e-8 * (1 + abs(r["primal_obj"][k]))
    model.free()
    for k in range(B):
        mk = hprlp.Model.from_csr(m, n, lp["rowptr"], lp["colind"], lp["values"], AL[:, k], AU[:, k], L[:, k], U[:, k], Cm[:, k])
        sv = hprlp.Solver(mk, hprlp.Parameters(stop_tol=tol, max_iter=60000, use_presolve=False))
        sv.scale()
        lam, _ = sv.power_iteration()
        sv.init(-1.0, lam * 1.01)
        s = sv.run()
        sv.close()
        assert s.status == "OPTIMAL", (k, s.status)
        assert abs(s.primal_obj - r["primal_obj"][k]) <= 10 * tol * (1 + abs(s.primal_obj)), (k, s.primal_obj, r["primal_obj"][k])
        mk.free()


def test_batch_with_an_infeasible_member_and_a_batch_of_one(gpu):
    """Members are independent: an infeasible one runs into the iteration limit (HPR-LP has no certificate) while its neighbours
    stop at the tolerance, with the oracle's statuses and stopping iterations; B = 1 is the single member's own batch."""
    g = json.load(open(os.path.join(HERE, "golden", "known_lps.json")))
    rp, ci, v = g[0]["rowptr"], g[0]["colind"], [float(t) for t in g[0]["values"]]
    model = hprlp.Model.from_csr(2, 2, rp, ci, v, [-INF, -INF], [10, 12], [0, 0], [INF, INF], [-3, -5])
    B = 3
    Cm = np.array([[-3.0, -5.0]] * B).T
    AU = np.array([[10.0, 12.0], [10.0, 12.0], [10.0, 12.0]]).T
    AL = np.full((2, B), -INF)
    L = np.zeros((2, B)); U = np.full((2, B), INF)
    AL[0, 1] = 11.0          # member 1: row 0 must be >= 11 ...
    U[:, 1] = 1.0            # ... with both variables at most 1 (the row reaches 3)
    prm = hprlp.Parameters(stop_tol=1e-6, max_iter=3000, use_presolve=False)
    r = hprlp.solve_batched(model, Cm, AL, AU, L, U, [0.0, 0.0, 7.0], prm)
    ref = O.solve_batched(2, 2, rp, ci, v, B, Cm.T.ravel(), AL.T.ravel(), AU.T.ravel(), L.T.ravel(), U.T.ravel(), [0.0, 0.0, 7.0],
                          params=O.Params.default(stop_tol=1e-6, max_iter=3000))
    assert r["status"] == ref["status"] == ["OPTIMAL", "ITER_LIMIT", "OPTIMAL"]
    assert list(r["iter"]) == list(ref["iter"]) and r["iter"][1] == 3000
    assert abs(r["primal_obj"][0] + 26.4) < 1e-4 and abs(r["primal_obj"][2] - (7.0 - 26.4)) < 1e-4
    one = hprlp.solve_batched(model, Cm[:, :1], AL[:, :1], AU[:, :1], L[:, :1], U[:, :1], [0.0], prm)
    assert one["status"] == ["OPTIMAL"] and one["iter"][0] == r["iter"][0]
    np.testing.assert_allclose(one["x"][:, 0], r["x"][:, 0], rtol=0, atol=1e-12)
    model.free()
