"""Deterministic synthetic LP generators with planted primal-dual optimal solutions.

None of the instances named in BASELINE.json (Netlib 25fv47, Mittelmann neos3 / pds-20) ship with
the reference or exist on the GPU box, so tests and bench.py use shape-matched stand-ins (SURVEY.md
§8d).  Every instance is min c'x s.t. AL <= Ax <= AU, l <= x <= u with a known optimal (x*, y*, z*):
c = A'y* + z*, b = Ax*, with sign-consistent multipliers, so the optimal objective c'x* is exact.
Sign convention of the boundary: y <= 0 on an active upper row bound, z >= 0 on an active lower
variable bound.
"""
import numpy as np
from scipy import sparse


def _pattern(rng, m, n, nnz_target, dense_col_frac=0.05, dense_col_max=200):
    """Random pattern: row lengths 1+Poisson, a few dense-ish columns; returns (rows, cols) unique."""
    n_dense = max(1, int(dense_col_frac * n)) if dense_col_frac > 0 else 0
    dense_cols = rng.choice(n, size=n_dense, replace=False) if n_dense else np.zeros(0, dtype=np.int64)
    dense_nnz = min(dense_col_max, max(2, m // 4))
    budget_dense = int(0.25 * nnz_target)
    per_dense = max(2, min(dense_nnz, budget_dense // max(n_dense, 1))) if n_dense else 0
    rows_d = np.concatenate([rng.choice(m, size=per_dense, replace=False) for _ in range(n_dense)]) if n_dense else np.zeros(0, dtype=np.int64)
    cols_d = np.repeat(dense_cols, per_dense) if n_dense else np.zeros(0, dtype=np.int64)
    rest = max(nnz_target - len(rows_d), m)
    lam = max(rest / m - 1.0, 0.05)
    lens = 1 + rng.poisson(lam, size=m)
    rows_s = np.repeat(np.arange(m), lens)
    cols_s = rng.integers(0, n, size=len(rows_s))
    rows = np.concatenate([rows_s, rows_d])
    cols = np.concatenate([cols_s, cols_d])
    # make sure every column appears at least once
    missing = np.setdiff1d(np.arange(n), np.unique(cols), assume_unique=False)
    if len(missing):
        rows = np.concatenate([rows, rng.integers(0, m, size=len(missing))])
        cols = np.concatenate([cols, missing])
    key = rows.astype(np.int64) * n + cols
    key = np.unique(key)
    return (key // n).astype(np.int64), (key % n).astype(np.int64)


def _plant(rng, A, free_frac=0.0):
    """Given sparse A (csr) build bounds, cost and the planted optimum."""
    m, n = A.shape
    # primal
    at_lower = rng.random(n) < 0.5
    x = np.where(at_lower, 0.0, rng.uniform(0.5, 2.0, size=n))
    l = np.zeros(n)
    u = np.full(n, np.inf)
    z = np.where(at_lower, rng.uniform(0.0, 1.0, size=n), 0.0)
    finite_u = rng.random(n) < 0.2
    at_upper = finite_u & ~at_lower & (rng.random(n) < 0.5)
    u[finite_u & at_lower] = rng.uniform(1.0, 3.0, size=int((finite_u & at_lower).sum()))
    u[at_upper] = x[at_upper]
    z[at_upper] = -rng.uniform(0.0, 1.0, size=int(at_upper.sum()))
    inter = finite_u & ~at_lower & ~at_upper
    u[inter] = x[inter] + rng.uniform(0.5, 2.0, size=int(inter.sum()))
    if free_frac > 0:  # free variables: interior, z = 0
        fr = (~at_lower) & (~finite_u) & (rng.random(n) < free_frac)
        l[fr] = -np.inf
    b = A @ x
    # rows: half equality, half "<=" (60 % of those active)
    is_eq = rng.random(m) < 0.5
    active = rng.random(m) < 0.6
    AL = np.full(m, -np.inf)
    AU = b.copy()
    y = np.zeros(m)
    AL[is_eq] = b[is_eq]
    y[is_eq] = rng.normal(size=int(is_eq.sum()))
    le_act = ~is_eq & active
    y[le_act] = -rng.uniform(0.0, 1.0, size=int(le_act.sum()))
    le_in = ~is_eq & ~active
    AU[le_in] = b[le_in] + rng.uniform(0.5, 2.0, size=int(le_in.sum()))
    c = A.T @ y + z
    return dict(AL=AL, AU=AU, l=l, u=u, c=c, x_star=x, y_star=y, z_star=z, obj_star=float(c @ x))


def planted_lp(m, n, nnz, seed, values="general", dense_col_frac=0.05, free_frac=0.0):
    """General planted LP.  values: 'general' = N(0,1)*10^U(-1,1); 'network' = +-1 with 10 % general."""
    rng = np.random.default_rng(seed)
    rows, cols = _pattern(rng, m, n, nnz, dense_col_frac=dense_col_frac)
    k = len(rows)
    if values == "network":
        v = rng.choice([-1.0, 1.0], size=k)
        g = rng.random(k) < 0.1
        v[g] = rng.normal(size=int(g.sum())) * 10.0 ** rng.uniform(-1, 1, size=int(g.sum()))
    else:
        v = rng.normal(size=k) * 10.0 ** rng.uniform(-1, 1, size=k)
    v[np.abs(v) < 1e-3] = 1e-3
    A = sparse.csr_matrix((v, (rows, cols)), shape=(m, n))
    A.sort_indices()
    out = _plant(rng, A, free_frac=free_frac)
    out.update(m=m, n=n, A=A, rowptr=A.indptr.astype(np.int32), colind=A.indices.astype(np.int32), values=A.data.copy())
    return out


def c2_25fv47_like(seed=2):
    """BASELINE config 2 stand-in: 821 x 1571, ~1.1e4 nnz."""
    return planted_lp(821, 1571, 10700, seed, values="general")


def c3_pds20_like(seed=3):
    """BASELINE config 3 stand-in: 33 874 x 105 728, ~2.3e5 nnz, network-flow-like values."""
    return planted_lp(33874, 105728, 230200, seed, values="network", dense_col_frac=0.0005)


def banded_csr(m, n, per_row, band, seed, row0=0, rows=None, frac_random=0.05):
    """Rows [row0,row0+rows) of the BASELINE config-5 matrix: `per_row` entries per row, column =
    row*n/m + U(-band,band) (95 %) or uniform (5 %); value N(0,1).  Each row is generated from its
    own counter so any shard can be produced independently."""
    rows = m - row0 if rows is None else rows
    r = np.arange(row0, row0 + rows, dtype=np.int64)
    ss = np.random.SeedSequence([seed, 0x5eed])
    rng = np.random.Generator(np.random.Philox(key=ss.generate_state(2, np.uint64), counter=[0, 0, 0, int(row0)]))
    center = (r * n) // m
    off = rng.integers(-band, band + 1, size=(rows, per_row))
    col = (center[:, None] + off) % n
    rnd = rng.random((rows, per_row)) < frac_random
    col = np.where(rnd, rng.integers(0, n, size=(rows, per_row)), col)
    col.sort(axis=1)
    val = rng.normal(size=(rows, per_row))
    rowptr = (np.arange(rows + 1, dtype=np.int64) * per_row).astype(np.int32)
    return rowptr, col.astype(np.int32).ravel(), val.ravel()


def banded_lp(m, n, per_row, band, seed):
    """Planted LP on the banded-random matrix (config 5 shape, any size)."""
    rp, ci, v = banded_csr(m, n, per_row, band, seed)
    A = sparse.csr_matrix((v, ci, rp), shape=(m, n))
    A.sum_duplicates()
    A.sort_indices()
    rng = np.random.default_rng(seed + 1)
    out = _plant(rng, A)
    out.update(m=m, n=n, A=A, rowptr=A.indptr.astype(np.int32), colind=A.indices.astype(np.int32), values=A.data.copy())
    return out


def block_angular_csr(K, mb, nb, per_row, link_rows, link_cols, link_len, seed):
    """Block-angular pattern (multi-commodity / stochastic-programming shape): K diagonal blocks of mb x nb with `per_row`
    random entries per row inside the block, plus `link_rows` linking ROWS and `link_cols` linking COLUMNS of `link_len`
    entries each spread over all blocks (the dense rows / columns real LPs of this class carry: the long-row paths of the
    kernels).  Returns (m, n, rowptr, colind, values) with m = K mb + link_rows, n = K nb + link_cols."""
    rng = np.random.default_rng(seed)
    m0, n0 = K * mb, K * nb
    m, n = m0 + link_rows, n0 + link_cols
    r = np.repeat(np.arange(m0, dtype=np.int64), per_row)
    c = (r // mb) * nb + rng.integers(0, nb, size=len(r))
    rows, cols = [r], [c]
    for q in range(link_rows):      # a linking row: link_len columns anywhere
        rows.append(np.full(link_len, m0 + q, np.int64))
        cols.append(rng.choice(n0, size=link_len, replace=False))
    for q in range(link_cols):      # a linking column: link_len rows anywhere
        rows.append(rng.choice(m0, size=link_len, replace=False))
        cols.append(np.full(link_len, n0 + q, np.int64))
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    key = np.unique(rows * n + cols)
    rows, cols = key // n, key % n
    vals = rng.normal(size=len(key))
    vals[np.abs(vals) < 1e-3] = 1e-3
    A = sparse.csr_matrix((vals, (rows, cols)), shape=(m, n))
    A.sort_indices()
    return m, n, A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.copy()


def block_angular_lp(K, mb, nb, per_row, link_rows, link_cols, link_len, seed):
    """Planted LP on the block-angular pattern."""
    m, n, rp, ci, v = block_angular_csr(K, mb, nb, per_row, link_rows, link_cols, link_len, seed)
    A = sparse.csr_matrix((v, ci, rp), shape=(m, n))
    out = _plant(np.random.default_rng(seed + 1), A)
    out.update(m=m, n=n, A=A, rowptr=rp, colind=ci, values=v)
    return out
