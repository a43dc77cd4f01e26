"""GPU: the RCCL code path of the row-partitioned solver with a ONE-rank communicator (the box has one
GPU).  Exercises librccl loading, communicator creation, the in-place all-gathers after every
half-step and the scalar all-reduces; with one rank they must not change any result."""
import numpy as np
import pytest

from conftest import hprlp, lpgen

pytestmark = pytest.mark.gpu


def test_one_rank_rccl_path_equals_plain_solver(gpu):
    lp = lpgen.planted_lp(400, 650, 4000, 91)
    model = hprlp.Model.from_csr(lp["m"], lp["n"], lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"],
                                 lp["l"], lp["u"], lp["c"])
    prm = hprlp.Parameters(stop_tol=1e-6, use_presolve=False)
    plain = hprlp.Solver(model, prm)
    uid = hprlp.Solver.dist_unique_id()
    dist = hprlp.Solver.create_dist(model, prm, 0, 1, uid)
    out = []
    for s in (plain, dist):
        s.scale()
        lam, it = s.power_iteration()
        s.init(-1.0, lam * 1.01)
        s.iterate(37, True)
        res = s.residuals(38, True)
        state = {k: s.get(k) for k in ("x", "y", "x_bar", "y_bar", "z_bar")}
        r = s.run()
        out.append((lam, it, res, state, r))
        s.close()
    (l0, i0, r0, s0, f0), (l1, i1, r1, s1, f1) = out
    assert (l0, i0) == (l1, i1)
    for k in s0:
        assert np.array_equal(s0[k], s1[k]), k
    for k in r0:
        assert r0[k] == r1[k], k
    assert (f0.status, f0.iter) == (f1.status, f1.iter) and f0.primal_obj == f1.primal_obj
    assert f0.status == "OPTIMAL" and abs(f0.primal_obj - lp["obj_star"]) <= 1e-5 * (1 + abs(lp["obj_star"]))
    model.free()
