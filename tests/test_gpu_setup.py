"""Device-side set-up: the transpose built on the GPU (csrc/transpose.hip) must equal the
host counting sort entry for entry (both are stable in row order), for ragged matrices with empty rows and columns,
and the solve that follows must not notice the difference; the background build of the tiled copies must give the
same iterates as a solver that never tiles."""
import os

import numpy as np
import pytest
from scipy import sparse

import bench_helpers as bh
from conftest import hprlp, lpgen

pytestmark = pytest.mark.gpu


def with_env(env, fn):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return fn()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_device_transpose_equals_host_transpose(gpu):
    rng = np.random.default_rng(12)
    m, n = 3001, 4507
    A = sparse.random(m, n, density=0.004, random_state=rng, format="lil", data_rvs=lambda k: rng.normal(size=k))
    A[17, :] = 0.0          # empty row
    A[:, 100:140] = 0.0     # empty columns
    for j in rng.choice(n, size=300, replace=False):
        A[5, j] = rng.normal()  # one long row
    A = A.tocsr(); A.eliminate_zeros(); A.sort_indices()
    x = rng.uniform(0, 1, size=n)
    b = A @ x
    model = hprlp.Model.from_csr(m, n, A.indptr, A.indices, A.data, b - 1, b + 1, np.zeros(n), np.full(n, 2.0), rng.normal(size=n))

    def grab():
        s = hprlp.Solver(model, hprlp.Parameters(use_presolve=False))
        at = s.get("AT_val")
        s.scale()
        lam, it = s.power_iteration()
        s.init(-1.0, lam * 1.01)
        s.iterate(20, True)
        out = (at, s.get("AT_val"), s.get("x"), s.get("y"), lam, it)
        s.close()
        return out

    host = with_env({"HPRLP_HOST_TRANSPOSE": "1"}, grab)
    dev = with_env({"HPRLP_DEVICE_TRANSPOSE_MIN": "0", "HPRLP_HOST_TRANSPOSE": "0"}, grab)
    AT = A.T.tocsr(); AT.sort_indices()
    assert np.array_equal(host[0], AT.data) and np.array_equal(dev[0], AT.data)
    for a, b_ in zip(host[1:4], dev[1:4]):
        assert np.array_equal(a, b_)
    assert host[4:] == dev[4:]
    model.free()


def test_background_tiling_gives_the_tiled_kernels_iterates(gpu):
    """With the tiled copy forced on, the solver must behave as if the copy had been there from the start: same
    iterates whether the build is adopted at the end of scale() (normal) or before anything runs (info() call)."""
    m = n = 20000
    lp = bh.banded_lp(m, n, 10, 300)
    model = hprlp.Model.from_csr(m, n, lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])

    def run(early):
        s = hprlp.Solver(model, hprlp.Parameters(use_presolve=False))
        if early:
            assert s.info()["tiled"] & 3 == 3  # info() adopts pending builds
        s.scale()
        lam, it = s.power_iteration()
        s.init(-1.0, lam * 1.01)
        s.iterate(30, True)
        assert s.info()["tiled"] & 3 == 3
        out = (s.get("x"), s.get("y"), lam, it)
        s.close()
        return out

    env = {"HPRLP_TILED_MIN_ROWS": "1", "HPRLP_TILED_MIN_DENSE": "0.0"}
    a = with_env(env, lambda: run(False))
    b = with_env(env, lambda: run(True))
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2:] == b[2:]
    model.free()


@pytest.mark.parametrize("case", ["banded", "narrow_band_long_segments", "ragged_with_empty_rows", "unsorted_columns", "wide_random"])
def test_device_tiled_build_equals_host_build(gpu, case):
    """HPRLP_TILING_CHECK=1 makes the solver build the tiled copies with BOTH builders (device: tiled_build.hip, host:
    tiled.cpp) and compare every array; creation fails with the first difference."""
    rng = np.random.default_rng(5)
    if case == "banded":
        m, n = 30000, 30000
        lp = bh.banded_lp(m, n, 12, 600)
        rp, ci, v = lp["rowptr"], lp["colind"], lp["values"]
    elif case == "narrow_band_long_segments":
        m, n = 9001, 2500
        lp = bh.banded_lp(m, n, 9, 40)
        rp, ci, v = lp["rowptr"], lp["colind"], lp["values"]
    else:
        m, n = (20011, 5003) if case == "ragged_with_empty_rows" else (17000, 40000)
        dens = 0.002 if case != "wide_random" else 0.0006
        A = sparse.random(m, n, density=dens, random_state=rng, format="csr", data_rvs=lambda k: rng.normal(size=k))
        if case == "ragged_with_empty_rows":
            A = A.tolil(); A[100:400, :] = 0.0; A = A.tocsr(); A.eliminate_zeros()
        A.sort_indices()
        rp, ci, v = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.copy()
        if case == "unsorted_columns":  # reverse the entries of every row: CSR order is not column order
            for i in range(m):
                ci[rp[i]:rp[i + 1]] = ci[rp[i]:rp[i + 1]][::-1]
                v[rp[i]:rp[i + 1]] = v[rp[i]:rp[i + 1]][::-1]
    x = rng.uniform(0, 1, size=n)
    b = sparse.csr_matrix((v, ci, rp), shape=(m, n)) @ x
    model = hprlp.Model.from_csr(m, n, rp, ci, v, b - 1, b + 1, np.zeros(n), np.full(n, 2.0), rng.normal(size=n))

    def make():
        s = hprlp.Solver(model, hprlp.Parameters(use_presolve=False))  # raises if the builders disagree
        info = s.info()
        s.scale()
        lam, _ = s.power_iteration()
        s.init(-1.0, lam * 1.01)
        s.iterate(5, True)
        out = (info["tiled"], s.get("x"), s.get("y"))
        s.close()
        return out

    env = {"HPRLP_TILED_MIN_ROWS": "1", "HPRLP_TILED_MIN_DENSE": "0.0", "HPRLP_TILING_CHECK": "1", "HPRLP_HOST_TRANSPOSE": "1"}
    dev = with_env(env, make)
    host = with_env({**env, "HPRLP_TILING_CHECK": "0", "HPRLP_HOST_TILING": "1"}, make)
    assert dev[0] & 3 == 3 and host[0] & 3 == 3
    assert np.array_equal(dev[1], host[1]) and np.array_equal(dev[2], host[2])
    model.free()


def test_power_start_vector_formed_on_the_device_for_long_vectors(gpu):
    """Above 1e6 rows the power iteration's start vector is generated by a kernel (kernels.hip: k_pw_start) instead of on the host
    (host_model.cpp: power_start_vector, what the oracle uses): same counter generator and formula, log / cos from the device
    library.  Same iteration count and lambda_max to 1e-10 as with the host-made vector."""
    m = n = 1_200_000
    lp = bh.banded_lp(m, n, 4, 5000)
    model = hprlp.Model.from_csr(m, n, lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])
    s = hprlp.Solver(model, hprlp.Parameters(use_presolve=False))
    s.scale()
    lam_d, it_d = s.power_iteration()
    lam_h, it_h = with_env({"HPRLP_HOST_POWER_START": "1"}, lambda: s.power_iteration())
    assert it_d == it_h and it_d >= 10
    assert abs(lam_d - lam_h) <= 1e-10 * lam_h
    s.close(); model.free()


def test_overlapped_setup_equals_the_sequential_one(gpu):
    """Round 4: the value upload, the host-side row blocks, the download of A^T's row pointers and the model vectors travel
    beside the device builds (helper threads, own copy streams; solver.cpp: describe_when).  HPRLP_NO_SETUP_OVERLAP=1 runs
    everything in line: the device arrays, the scaled data and the iterates must be the same bits, on a matrix large enough
    for every overlap to be taken (over 4e6 nonzeros, over 1e5 rows, tiled copies)."""
    m = n = 300000
    lp = bh.banded_lp(m, n, 16, 3000, seed=77)

    def grab():
        model = hprlp.Model.from_csr(m, n, lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])
        s = hprlp.Solver(model, hprlp.Parameters(use_presolve=False))
        desc = s.describe()
        raw = [s.get(k) for k in ("A_val", "AT_val", "AL", "AU", "l", "u", "c")]
        s.scale()
        lam, it = s.power_iteration()
        s.init(-1.0, lam * 1.01)
        s.iterate(12, True)
        out = (desc, raw, [s.get(k) for k in ("A_val", "AT_val", "x", "y")], lam, it)
        s.close()
        model.free()
        return out

    seq = with_env({"HPRLP_NO_SETUP_OVERLAP": "1", "HPRLP_TILED_MIN_ROWS": "65536"}, grab)
    ovl = with_env({"HPRLP_TILED_MIN_ROWS": "65536"}, grab)
    forms = lambda d: d.split("; switches:")[0]      # (the description ends with the switches a solver was set up under)
    assert "tiled" in ovl[0] and forms(seq[0]) == forms(ovl[0]) and "HPRLP_NO_SETUP_OVERLAP=1" in seq[0]
    for a, b_ in zip(seq[1] + seq[2], ovl[1] + ovl[2]):
        assert np.array_equal(a, b_)
    assert seq[3:] == ovl[3:]
