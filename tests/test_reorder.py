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


def block_plan(m, n, rp, ci, cuts=True):
    L = hprlp.lib()
    out = np.zeros(6, np.int64)
    rc = L.hprlp_row_block_plan(m, n, rp.ctypes.data_as(hprlp.c_int_p), ci.ctypes.data_as(hprlp.c_int_p), int(cuts),
                                out.ctypes.data_as(__import__("ctypes").POINTER(__import__("ctypes").c_long)))
    return rc, out


def test_row_blocks_of_dense_rows_cut_by_column_eighths_stay_within_a_chunk():
    """solver.cpp slab_cuts (round-4 advisor finding): a short last chunk joins its predecessor only while the joined chunk is
    at most kSplitRow = 4096 entries.  Rows of 4100 entries in one column eighth ([0,4096) + a 4-entry tail) and of 4050 + 60
    entries in two eighths used to give a 4100 / 4110-entry chunk that the set-up then refused ("bad split-row block").  A
    sweep of random dense rows (about 1 % hit the old defect) is planned and checked on the host."""
    rng = np.random.default_rng(7)
    n = 1 << 20
    W = n // 8
    rows = []
    rows.append(3 * W + rng.choice(W - 10, 4100, replace=False))                                             # 4096 + 4
    rows.append(np.concatenate([2 * W + rng.choice(W - 10, 4050, replace=False), 6 * W + rng.choice(W - 10, 60, replace=False)]))
    rows.append(5 * W + rng.choice(W - 10, 4096 + 95, replace=False))                                        # 4096 + a 95-entry tail: two chunks
    rows.append(5 * W + rng.choice(W - 10, 4000, replace=False))                                             # one chunk
    rows.append(np.concatenate([1 * W + rng.choice(W - 10, 3000, replace=False), 4 * W + rng.choice(W - 10, 50, replace=False)]))  # 3050: the tail joins
    for _ in range(300):
        L = int(rng.integers(2048, 60000))
        k = int(rng.integers(1, 9))
        eighths = rng.choice(8, k, replace=False)
        parts = rng.multinomial(L, rng.dirichlet(np.ones(k)))
        rows.append(np.concatenate([e * W + rng.choice(W - 10, p, replace=False) for e, p in zip(eighths, parts) if p > 0]))
    rows = [np.sort(r) for r in rows]
    rp = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
    ci = np.concatenate(rows).astype(np.int32)
    m = len(rows)
    rc, out = block_plan(m, n, rp, ci)
    assert rc == 0, hprlp.last_error()
    assert out[3] == m and out[1] == m and out[4] <= 4096 and out[5] == rp[-1], out
    # the first rows one by one: chunk counts as the rule gives them
    for i, want in enumerate((2, 2, 2, 1, 1)):
        rc, o1 = block_plan(1, n, np.array([0, len(rows[i])], np.int32), rows[i].astype(np.int32))
        assert rc == 0 and o1[2] == want and o1[4] <= 4096, (i, o1, hprlp.last_error())
    # without cuts the same rows fall to kSplitRow chunks
    rc, out = block_plan(m, n, rp, ci, cuts=False)
    assert rc == 0 and out[3] == 0 and out[4] <= 4096 and out[5] == rp[-1]


def tiled_check(m, n, rp, ci, R, T, min_dense=0.0):
    import ctypes as C
    L = hprlp.lib()
    out = np.zeros(6, np.int64)
    L.hprlp_tiled_host_check.argtypes = [C.c_int, C.c_int, hprlp.c_int_p, hprlp.c_int_p, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_long)]
    rc = L.hprlp_tiled_host_check(m, n, rp.ctypes.data_as(hprlp.c_int_p), ci.ctypes.data_as(hprlp.c_int_p), R, T, min_dense,
                                  out.ctypes.data_as(C.POINTER(C.c_long)))
    return rc, out


def test_layered_tile_lists_keep_the_kernels_invariants():
    """tiled.h: kTileLayers (round 5).  A row's entries beyond four in one tile go to further LAYERS of the tile's list instead of
    sending the row's segment to the remainder.  The host builder's structure on narrow bands (5, 10 and 24 entries of a row per
    tile: two, three and four layers + an overflow to the remainder), on a pattern with far entries and ragged rows, is verified
    entry by entry (hprlp_tiled_host_check): every CSR entry exactly once, codes name their entries, inside a step one chunk per
    accumulator, steps within capacity; the layers show as consecutive steps of one tile and as the staged share."""
    rng = np.random.default_rng(5)

    def band(m, per_row, half, far=0.0):
        r = np.repeat(np.arange(m), per_row)
        c = np.abs(r + rng.integers(-half, half + 1, size=len(r)))
        c = np.where(c > m - 1, 2 * (m - 1) - c, c)
        if far:
            isfar = rng.random(len(r)) < far
            c = np.where(isfar, rng.integers(0, m, size=len(r)), c)
        A = sparse.csr_matrix((np.ones(len(r)), (r, c)), shape=(m, m)); A.sum_duplicates(); A.sort_indices()
        return A.indptr.astype(np.int32), A.indices.astype(np.int32), A.nnz

    m = 40_000
    # 20 per row in 4 000 columns: ~5 per 1024-column tile -> two layers; before the layers two thirds of it went to the remainder
    rp, ci, nnz = band(m, 20, 2000)
    rc, o = tiled_check(m, m, rp, ci, 2048, 1024)
    assert rc == 0, hprlp.last_error()
    assert o[5] >= 850_000 and o[1] <= 0.15 * nnz, o          # >= 85 % staged
    assert o[0] - o[3] + o[1] == nnz                            # tile entries without padding + remainder = all entries
    assert o[4] >= 2                                            # a tile's layers: consecutive steps of one tile
    # 40 per row in 4 000 columns: ~10 per tile -> three layers
    rp, ci, nnz = band(m, 40, 2000)
    rc, o = tiled_check(m, m, rp, ci, 1024, 1024)
    assert rc == 0 and o[5] >= 850_000 and o[0] - o[3] + o[1] == nnz, (o, hprlp.last_error())
    # 60 per row in 2 400 columns: ~24 per tile -> four layers and the rest to the remainder; 10 % far entries; wide tiles
    rp, ci, nnz = band(m, 60, 1200, far=0.1)
    for T in (1024, 2048):
        rc, o = tiled_check(m, m, rp, ci, 1024, T)
        assert rc == 0 and o[0] - o[3] + o[1] == nnz, (T, o, hprlp.last_error())
        assert 400_000 <= o[5] <= 900_000, o                    # 16 of ~24-48 per row and tile staged, the far entries not
    # ragged: empty rows, rows of 1..200 entries, everything in ONE tile column range (up to 50 layers' worth: overflow)
    lens = rng.integers(0, 200, size=5000); lens[::7] = 0
    r = np.repeat(np.arange(5000), lens)
    c = rng.integers(0, 900, size=len(r))
    A = sparse.csr_matrix((np.ones(len(r)), (r, c)), shape=(5000, 4096)); A.sum_duplicates(); A.sort_indices()
    rc, o = tiled_check(5000, 4096, A.indptr.astype(np.int32), A.indices.astype(np.int32), 1024, 1024)
    assert rc == 0 and o[0] - o[3] + o[1] == A.nnz and o[1] > 0, (o, hprlp.last_error())
    # the check itself notices a broken pattern description: the builder refuses nothing here, but a declined build is an error
    rc, _ = tiled_check(m, m, rp, ci, 1024, 1024, min_dense=1.01)
    assert rc == -1 and "declined" in hprlp.last_error()
