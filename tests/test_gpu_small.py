"""The single-workgroup small-LP kernel (hpr-lp-c_amd/csrc/small.hip): `count` normal HPR iterations in one
launch with both matrices in registers.  It must produce bit-identical iterates to the regular per-half-step
kernels and to the oracle (same per-row arithmetic), and hand the state back correctly to the check-variant
kernels that run between its launches."""
import os

import numpy as np
import pytest

from conftest import hprlp, lpgen
from oracle import oracle as O
from test_gpu_kernels import NAMES_M, NAMES_N, adopt_gpu_data, run_steps

pytestmark = pytest.mark.gpu

VECS = ("x", "x_hat", "y", "last_x", "last_y")
CHECK_VECS = ("x_bar", "z_bar", "x_temp", "y_bar", "y_obj", "y_temp")  # what a check-variant step leaves besides


def make(lp):
    return hprlp.Model.from_csr(lp["m"], lp["n"], lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"],
                                lp["u"], lp["c"])


def iterate_states(model, prm, plan, no_small):
    old = os.environ.get("HPRLP_NO_SMALL")
    os.environ["HPRLP_NO_SMALL"] = "1" if no_small else "0"
    # (what is compared bit for bit are the ITERATION kernels: lambda_max comes from the regular power iteration in both runs;
    # the single-launch one adds its dot products in another order -- test_small_power_iteration_matches_... below)
    os.environ["HPRLP_NO_SMALL_POWER"] = "1"
    try:
        s = hprlp.Solver(model, prm)
        assert bool(s.info()["tiled"] & 4) == (not no_small)
        s.scale()
        lam, _ = s.power_iteration()
        s.init(-1.0, lam * 1.01)
        out = []
        done = 0
        res_after_check = []
        for normal, check in plan:
            s.iterate(normal, check)
            done += normal + (1 if check else 0)
            out.append({k: s.get(k) for k in VECS + (CHECK_VECS if check else ())})
            out[-1]["k"] = (s.scalars()["kx"], s.scalars()["ky"])
            if check:
                # right behind a check step: the regular residual kernels on the state the check step left
                res_after_check.append(s.residuals(done, True))
        res = s.residuals(done, True)
        s.close()
        return out, res, res_after_check
    finally:
        os.environ.pop("HPRLP_NO_SMALL_POWER", None)
        if old is None:
            os.environ.pop("HPRLP_NO_SMALL", None)
        else:
            os.environ["HPRLP_NO_SMALL"] = old


# (m, n, nnz): one row per thread / two rows per thread / the 512-thread variant (more than 8 entries per thread)
@pytest.mark.parametrize("shape", [(300, 500, 2500), (600, 1500, 3500), (500, 1000, 7500), (821, 1571, 7000),
                                   (821, 1571, 10700), (1900, 2040, 11500)])
def test_small_kernel_equals_regular_kernels_bit_for_bit(gpu, shape):
    m, n, nnz = shape
    lp = lpgen.planted_lp(m, n, nnz, 5, dense_col_frac=0.01 if m != 300 else 0.0)  # (300, 500, 2500): no dense columns, rows <= 64
    model = make(lp)
    prm = hprlp.Parameters(use_presolve=False)
    plan = [(1, False), (7, True), (64, False), (149, True), (3, False)]
    small, res_s, chk_s = iterate_states(model, prm, plan, no_small=False)
    regular, res_r, chk_r = iterate_states(model, prm, plan, no_small=True)
    # The single-workgroup kernel adds every row in CSR order; the regular kernels do so for rows of up to 64 entries
    # (kernels.h: kLongRow) and hand longer rows to a whole wave (strided partial sums + a wave sum).  Shapes without such
    # rows must agree bit for bit, the others to rounding.
    from scipy import sparse
    A = sparse.csr_matrix((lp["values"], lp["colind"], lp["rowptr"]), shape=(m, n))
    longest = max(np.diff(A.indptr).max(), np.diff(A.tocsc().indptr).max())
    for a, b in zip(small, regular):
        assert a["k"] == b["k"]
        for k in [q for q in a if q != "k"]:
            if longest <= 64:
                assert np.array_equal(a[k], b[k]), k
            else:
                np.testing.assert_allclose(a[k], b[k], rtol=1e-11, atol=1e-13, err_msg=k)
    for k in res_r:
        # (the objective terms and |x_temp|, |y_temp| were summed by the last check step: in the fused launch's order on the small path)
        tol = 1e-12 if longest <= 64 else 1e-9
        assert abs(res_s[k] - res_r[k]) <= tol * (1 + abs(res_r[k])), k
    # the residual evaluations formed inside the check step's launch (round 4): the same sums in another order
    assert len(chk_s) == len(chk_r) == 2
    for rs, rr in zip(chk_s, chk_r):
        for k in rr:
            tol = 1e-12 if longest <= 64 else 1e-9
            assert abs(rs[k] - rr[k]) <= tol * (1 + abs(rr[k])), (k, rs[k], rr[k])
    model.free()


@pytest.mark.parametrize("dense", [0.0, 0.01])
def test_small_kernel_matches_oracle(gpu, dense):
    """Normal steps by the single-workgroup kernel, check steps by the regular kernels, against the oracle: bit for bit when no
    row has more than 64 entries (kLongRow: the regular kernels add longer rows wave-wide), to rounding otherwise."""
    m, n = 400, 650
    lp = lpgen.planted_lp(m, n, 4000, 8, dense_col_frac=dense)
    model = make(lp)
    s = hprlp.Solver(model, hprlp.Parameters(use_presolve=False, use_CR_scaling=False))
    assert s.info()["tiled"] & 4
    ref = O.ScaledLP(m, n, lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"],
                     O.Params.default(use_CR_scaling=0))
    s.scale()
    adopt_gpu_data(s, ref)
    st = run_steps(s, ref, 0.7, 1.3, [(23, True), (5, True), (40, False)])
    longest = max(np.diff(ref.Arp).max(), np.diff(ref.ATrp).max())
    assert (longest <= 64) == (dense == 0.0)
    for name in NAMES_N + NAMES_M:
        if longest <= 64:
            assert np.array_equal(s.get(name), st[name]), name
        else:
            np.testing.assert_allclose(s.get(name), st[name], rtol=1e-11, atol=1e-13, err_msg=name)
    s.close(); model.free()


def test_small_power_iteration_matches_regular_kernels_and_oracle(gpu):
    """The whole power iteration in one launch of the single-workgroup kernel (k_small_power, stopping test on the device):
    same number of iterations as the regular kernels (host test every 10th) and as the oracle, lambda equal to rounding
    (the three dot products are added in another order)."""
    for lp in (lpgen.c2_25fv47_like(), lpgen.planted_lp(400, 650, 4000, 8), lpgen.planted_lp(1900, 1200, 9000, 4)):
        model = make(lp)
        s = hprlp.Solver(model, hprlp.Parameters(use_presolve=False))
        assert s.info()["tiled"] & 4
        s.scale()
        lam, it = s.power_iteration()
        os.environ["HPRLP_NO_SMALL_POWER"] = "1"
        try:
            lam_r, it_r = s.power_iteration()
        finally:
            os.environ.pop("HPRLP_NO_SMALL_POWER", None)
        ref = O.ScaledLP(lp["m"], lp["n"], lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"],
                         O.Params.default())
        lam_o, it_o = ref.power_iteration()
        assert it == it_r == it_o, (it, it_r, it_o)
        assert abs(lam - lam_r) <= 1e-12 * lam_r and abs(lam - lam_o) <= 1e-11 * lam_o, (lam, lam_r, lam_o)
        # a capped run stops at the cap and still reports the last lambda it formed
        lam_c, it_c = s.power_iteration(max_iter=25)
        assert it_c == 25 and lam_c > 0
        s.close(); model.free()


def test_whole_solve_on_the_small_path(gpu):
    lp = lpgen.c2_25fv47_like()
    model = make(lp)
    prm = hprlp.Parameters(stop_tol=1e-6, use_presolve=False)
    # (bit-for-bit comparison of the ITERATION kernels: both solves take lambda_max from the regular power iteration --
    # the single-launch one adds its dot products in another order, test above)
    os.environ["HPRLP_NO_SMALL_POWER"] = "1"
    os.environ["HPRLP_NO_SMALL"] = "1"
    try:
        r_reg = model.solve(prm)
        os.environ.pop("HPRLP_NO_SMALL", None)
        r_small = model.solve(prm)
    finally:
        os.environ.pop("HPRLP_NO_SMALL", None)
        os.environ.pop("HPRLP_NO_SMALL_POWER", None)
    assert r_small.status == r_reg.status == "OPTIMAL"
    assert r_small.iter == r_reg.iter and r_small.primal_obj == r_reg.primal_obj
    assert np.array_equal(r_small.x, r_reg.x) and np.array_equal(r_small.y, r_reg.y)
    assert abs(r_small.primal_obj - lp["obj_star"]) <= 1e-5 * (1 + abs(lp["obj_star"]))
    model.free()


def test_rows_too_long_for_the_small_path_fall_back(gpu):
    """A 300-entry row exceeds the per-thread sequential sum limit (256): the regular kernels take over."""
    from scipy import sparse
    rng = np.random.default_rng(3)
    lp = lpgen.planted_lp(200, 600, 1500, 6)
    A = sparse.csr_matrix((lp["values"], lp["colind"], lp["rowptr"]), shape=(200, 600)).tolil()
    for j in rng.choice(600, size=300, replace=False):
        A[0, j] = 1.0
    A = A.tocsr(); A.sort_indices()
    x = np.abs(lp["x_star"]) + 0.1
    b = A @ x
    model = hprlp.Model.from_csr(200, 600, A.indptr, A.indices, A.data, b - 1.0, b + 1.0, np.zeros(600), np.full(600, 10.0), lp["c"])
    s = hprlp.Solver(model, hprlp.Parameters(use_presolve=False))
    assert not (s.info()["tiled"] & 4)
    s.close(); model.free()
