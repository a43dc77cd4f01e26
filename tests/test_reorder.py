"""CPU: the set-up time locality ordering (hpr-lp-c_amd/csrc/reorder.cpp, host path) recovers the band of a randomly
permuted banded-random matrix (the BASELINE config-5 generator: 95 % of a row's entries within +-band of the diagonal,
5 % anywhere -- the far entries are what defeats BFS / Cuthill-McKee orderings), and leaves an already local order alone."""
import numpy as np
from scipy import sparse

import bench_helpers as bh
from conftest import hprlp


def order(m, n, rp, ci):
    L = hprlp.lib()
    r = np.zeros(m, np.int32); c = np.zeros(n, np.int32); out = np.zeros(6)
    rc = L.hprlp_locality_ordering(m, n, rp.ctypes.data_as(hprlp.c_int_p), ci.ctypes.data_as(hprlp.c_int_p),
                                   r.ctypes.data_as(hprlp.c_int_p), c.ctypes.data_as(hprlp.c_int_p), out.ctypes.data_as(hprlp.c_dbl_p))
    assert rc == 0, hprlp.last_error()
    return bool(out[0]), out[1], out[2], r, c


def permuted(m, n, per_row, band, seed):
    rp, ci, v = bh.gen_banded(m, n, per_row, band, seed=5)
    A = sparse.csr_matrix((v, ci, rp), shape=(m, n))
    rng = np.random.default_rng(seed)
    pr, pc = rng.permutation(m), rng.permutation(n)
    inv_pc = np.empty(n, np.int64); inv_pc[pc] = np.arange(n)
    B = A[pr]
    B = sparse.csr_matrix((B.data, inv_pc[B.indices], B.indptr), shape=(m, n))
    B.sort_indices()
    return A, B, pr, pc


def test_local_order_is_left_alone():
    m = n = 1_600_000
    rp, ci, v = bh.gen_banded(m, n, 10, 16000, seed=5)
    ok, before, after, _, _ = order(m, n, rp, ci)
    assert not ok and before > 0.8


def test_band_of_a_permuted_matrix_is_recovered():
    m = n = 1_600_000  # large enough for a random order to FAIL the tiling test (fewer than 256 entries per tile)
    band = 16000
    A, B, _, _ = permuted(m, n, 10, band, 3)
    rp, ci = B.indptr.astype(np.int32), B.indices.astype(np.int32)
    ok, before, after, r, c = order(m, n, rp, ci)
    assert ok and before < 0.05 and after > 0.8, (ok, before, after)
    assert np.array_equal(np.sort(r), np.arange(m)) and np.array_equal(np.sort(c), np.arange(n))  # permutations
    # distance from the diagonal in the new numbering: the bulk is back inside a few band widths
    c_old2new = np.empty(n, np.int64); c_old2new[c] = np.arange(n)
    C2 = B[r]
    rows = np.repeat(np.arange(m), np.diff(C2.indptr))
    d = np.abs(c_old2new[C2.indices] - rows)
    q50, q90 = np.quantile(d, [0.5, 0.9])
    assert q50 <= 1.5 * band / 2 * 1.2 and q90 <= 6 * band, (q50, q90)
    # deterministic
    ok2, _, _, r2, c2 = order(m, n, rp, ci)
    assert ok2 and np.array_equal(r, r2) and np.array_equal(c, c2)
