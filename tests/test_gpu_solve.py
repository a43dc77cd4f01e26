"""End-to-end parity of the HIP solve against the CPU oracle and against known optima (GPU)."""
import numpy as np
import pytest

from conftest import hprlp, lpgen
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _model(lp):
    return hprlp.Model.from_csr(lp["m"], lp["n"], lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"],
                                lp["l"], lp["u"], lp["c"])


def test_model_mps_known_answer(gpu, model_mps_arrays):
    """reference examples/cpp/example_direct_lp.cpp:14: x=(2.8,3.6), obj=-26.4."""
    a = model_mps_arrays
    model = _model(a)
    res = model.solve(hprlp.Parameters(stop_tol=1e-9, use_presolve=False))
    ref = O.solve(a["m"], a["n"], a["rowptr"], a["colind"], a["values"], a["AL"], a["AU"], a["l"], a["u"], a["c"],
                  params=O.Params.default(stop_tol=1e-9))
    assert res.status == "OPTIMAL"
    assert abs(res.primal_obj + 26.4) < 1e-6
    np.testing.assert_allclose(res.x, [2.8, 3.6], atol=1e-6)
    np.testing.assert_allclose(res.y, [-2.4, -0.2], atol=1e-6)
    assert res.iter == ref["iter"]
    np.testing.assert_allclose(res.x, ref["x"], rtol=0, atol=1e-10)
    model.free()


@pytest.mark.parametrize("m,n,nnz,seed", [(50, 80, 400, 11), (300, 500, 3000, 12)])
def test_planted_lp_matches_oracle(gpu, m, n, nnz, seed):
    lp = lpgen.planted_lp(m, n, nnz, seed)
    model = _model(lp)
    tol = 1e-6
    res = model.solve(hprlp.Parameters(stop_tol=tol, use_presolve=False))
    ref = O.solve(m, n, lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"],
                  params=O.Params.default(stop_tol=tol))
    assert res.status == "OPTIMAL"
    assert abs(res.primal_obj - lp["obj_star"]) / (1 + abs(lp["obj_star"])) <= 10 * tol
    # same schedule => same iteration count unless a thresholded decision forks (FP64 reduction order)
    assert abs(res.iter - ref["iter"]) <= 0.2 * ref["iter"] + 150
    model.free()


def test_reset_iterates_gives_the_run_of_a_fresh_solver(gpu):
    """hprlp_solver_reset_iterates: after some iterations (normal and check variants) the solver is set back to zero iterates;
    the run that follows is the run of a fresh solver, bit for bit (bench.py uses this for the solve to tolerance of a
    multi-GPU run, whose solver also did the timed iterations)."""
    lp = lpgen.planted_lp(2500, 4000, 30000, 21)
    model = hprlp.Model.from_csr(lp["m"], lp["n"], lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])
    prm = hprlp.Parameters(stop_tol=1e-6, use_presolve=False)

    def prepared():
        s = hprlp.Solver(model, prm)
        s.scale()
        lam, _ = s.power_iteration()
        s.init(-1.0, lam * 1.01)
        return s, lam

    fresh, lam = prepared()
    r0 = fresh.run()
    fresh.close()
    used, lam2 = prepared()
    assert lam2 == lam
    used.iterate(37, True)
    used.iterate(150, True)
    used.residuals(189, True)
    used.reset()
    used.init(-1.0, lam * 1.01)
    r1 = used.run()
    used.close()
    assert r1.status == r0.status == "OPTIMAL" and r1.iter == r0.iter
    assert r1.primal_obj == r0.primal_obj and np.array_equal(r1.x, r0.x) and np.array_equal(r1.y, r0.y)
    model.free()


def test_warmup_then_solve(gpu, model_mps_arrays):
    """hprlp_warmup (round 4): runtime, device context, first stream and the library's code objects up front; returns 0 on a GPU
    box, reports its phases, and a solve behind it gives the known answer (reference examples/cpp/example_direct_lp.cpp:14)."""
    import ctypes as C
    L = hprlp.lib()
    L.hprlp_warmup.restype = C.c_int
    assert L.hprlp_warmup(0) == 0
    assert L.hprlp_warmup(99) == -1 and "outside" in hprlp.last_error()   # (a wrong device is refused, not mapped to 0)
    out = (C.c_double * 4)()
    assert L.hprlp_warmup_seconds(out) == 0 and out[3] > 0 and abs(out[0] + out[1] + out[2] - out[3]) <= 1e-6
    a = model_mps_arrays
    model = hprlp.Model.from_csr(a["m"], a["n"], a["rowptr"], a["colind"], a["values"], a["AL"], a["AU"], a["l"], a["u"], a["c"])
    r = model.solve(hprlp.Parameters(stop_tol=1e-6, use_presolve=False))
    assert r.status == "OPTIMAL" and abs(r.primal_obj + 26.4) <= 1e-3
    model.free()
