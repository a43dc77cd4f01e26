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


# --------------------------------------------------------------------------------------------------------------------
# Families of the Mittelmann LP set (round 4).  The named instances (pds-xx, nug-xx, cont-xx, staircase multi-stage models)
# are not available offline; these are deterministic generators of the same CONSTRUCTIONS at any size -- real LPs with the
# families' structure and degeneracy (no planted optimum: the tests take the optimum from HiGHS at small sizes and from
# the oracle / the KKT conditions beyond).  Every generator returns the usual dict (m, n, rowptr, colind, values, AL, AU, l,
# u, c) plus `family`.
# --------------------------------------------------------------------------------------------------------------------
def _finish(A, AL, AU, l, u, c, family):
    A = sparse.csr_matrix(A)
    A.sum_duplicates()
    A.sort_indices()
    m, n = A.shape
    return dict(m=m, n=n, A=A, rowptr=A.indptr.astype(np.int32), colind=A.indices.astype(np.int32), values=A.data.astype(np.float64),
                AL=np.asarray(AL, float), AU=np.asarray(AU, float), l=np.asarray(l, float), u=np.asarray(u, float), c=np.asarray(c, float),
                family=family)


def multicommodity_flow_lp(grid, commodities, seed, cap_tightness=0.7):
    """pds-like (Patient Distribution System: multicommodity minimum-cost flow).  A grid x grid torus of nodes, arcs to the four
    neighbours plus 20 % random long arcs; K commodities, each ships one unit-scaled demand from a random source to a random
    sink.  Variables: flow x[k, a] >= 0 and one artificial arc per commodity (source -> sink, expensive, uncapacitated: the LP is
    always feasible).  Rows: flow conservation per (commodity, node) -- K node-arc incidence blocks, +-1 entries, equalities --
    and one joint capacity row per arc, sum_k x[k, a] <= cap_a.  Capacities are `cap_tightness` x the load of the
    shortest-path routing on the busiest arcs (so they bind) and generous elsewhere.  Block-angular, +-1 matrix, degenerate."""
    rng = np.random.default_rng(seed)
    V = grid * grid
    idx = np.arange(V).reshape(grid, grid)
    tails = np.concatenate([idx.ravel()] * 4)
    heads = np.concatenate([np.roll(idx, -1, 0).ravel(), np.roll(idx, 1, 0).ravel(), np.roll(idx, -1, 1).ravel(), np.roll(idx, 1, 1).ravel()])
    extra = V // 5 * 4 // 4
    et, eh = rng.integers(0, V, extra), rng.integers(0, V, extra)
    keep = et != eh
    tails, heads = np.concatenate([tails, et[keep]]), np.concatenate([heads, eh[keep]])
    Aarcs = len(tails)
    cost = rng.uniform(1.0, 3.0, Aarcs)
    K = commodities
    src, dst = rng.integers(0, V, K), rng.integers(0, V, K)
    dst = np.where(dst == src, (dst + 1) % V, dst)
    dem = rng.uniform(1.0, 4.0, K)
    # shortest-path routing (by arc cost) of every commodity: the load that sizes the capacities
    from scipy.sparse.csgraph import dijkstra
    Gm = sparse.csr_matrix((cost, (tails, heads)), shape=(V, V))
    Gm.sum_duplicates()
    load = np.zeros(Aarcs)
    arc_of = {}
    for a in range(Aarcs):
        key = (int(tails[a]), int(heads[a]))
        if key not in arc_of or cost[a] < cost[arc_of[key]]:
            arc_of[key] = a
    _, pred = dijkstra(Gm, indices=src, return_predecessors=True)
    for k in range(K):
        v = int(dst[k])
        while v != int(src[k]) and pred[k, v] >= 0:
            p = int(pred[k, v])
            load[arc_of[(p, v)]] += dem[k]
            v = p
    cap = np.where(load > 0, np.maximum(cap_tightness * load, 0.5), rng.uniform(2.0, 6.0, Aarcs))
    nvar_k = Aarcs + 1   # + the artificial arc
    n = K * nvar_k
    m = K * V + Aarcs
    rows, cols, vals = [], [], []
    for k in range(K):
        c0, r0 = k * nvar_k, k * V
        a = np.arange(Aarcs)
        rows += [r0 + tails, r0 + heads, K * V + a]
        cols += [c0 + a, c0 + a, c0 + a]
        vals += [np.ones(Aarcs), -np.ones(Aarcs), np.ones(Aarcs)]
        rows += [np.array([r0 + src[k], r0 + dst[k]])]
        cols += [np.array([c0 + Aarcs, c0 + Aarcs])]
        vals += [np.array([1.0, -1.0])]
    A = sparse.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(m, n))
    b = np.zeros(K * V)
    for k in range(K):
        b[k * V + src[k]] += dem[k]
        b[k * V + dst[k]] -= dem[k]
    AL = np.concatenate([b, np.full(Aarcs, -np.inf)])
    AU = np.concatenate([b, cap])
    c = np.tile(np.concatenate([cost, [60.0]]), K) * np.repeat(rng.uniform(0.8, 1.2, K), nvar_k)
    return _finish(A, AL, AU, np.zeros(n), np.full(n, np.inf), c, "multicommodity flow (pds-like)")


def qap_lp_relaxation(nfac, seed):
    """nug-like: the Adams-Johnson linearisation of a quadratic assignment problem with the symmetry y[i,j,k,l] = y[k,l,i,j]
    substituted out (the form of the nugXX LPs: nfac = 8 gives their 912 x 1632).  Variables x[i,j] (facility i at location j) and
    y[i,j,k,l] for i < k, j != l; rows: the 2 nfac assignment equalities and, for every (i, j), one equality per other location
    l and one per other facility k tying the y's to x[i,j].  Flows and distances are random integers (Nugent-style grid
    distances); all equalities, every vertex massively degenerate -- the hard end of the set for first-order methods."""
    rng = np.random.default_rng(seed)
    nf = nfac
    side = int(np.ceil(np.sqrt(nf)))
    loc = np.array([(q // side, q % side) for q in range(nf)])
    D = np.abs(loc[:, None, :] - loc[None, :, :]).sum(axis=2).astype(float)
    F = np.triu(rng.integers(0, 7, size=(nf, nf)), 1).astype(float)
    F = F + F.T
    nx = nf * nf
    pairs = [(i, k) for i in range(nf) for k in range(i + 1, nf)]
    pidx = {p: q for q, p in enumerate(pairs)}
    jl = [(j, l) for j in range(nf) for l in range(nf) if j != l]
    jlidx = {p: q for q, p in enumerate(jl)}
    ny = len(pairs) * len(jl)
    n = nx + ny

    def yv(i, j, k, l):   # variable of y[i,j,k,l], any order of the two (facility, location) pairs
        if i > k:
            i, j, k, l = k, l, i, j
        return nx + pidx[(i, k)] * len(jl) + jlidx[(j, l)]

    rows, cols, vals = [], [], []
    r = 0
    for j in range(nf):
        for i in range(nf):
            rows.append(r); cols.append(i * nf + j); vals.append(1.0)
        r += 1
    for i in range(nf):
        for j in range(nf):
            rows.append(r); cols.append(i * nf + j); vals.append(1.0)
        r += 1
    for i in range(nf):
        for j in range(nf):
            for l in range(nf):
                if l == j:
                    continue
                for k in range(nf):
                    if k != i:
                        rows.append(r); cols.append(yv(i, j, k, l)); vals.append(1.0)
                rows.append(r); cols.append(i * nf + j); vals.append(-1.0)
                r += 1
            for k in range(nf):
                if k == i:
                    continue
                for l in range(nf):
                    if l != j:
                        rows.append(r); cols.append(yv(i, j, k, l)); vals.append(1.0)
                rows.append(r); cols.append(i * nf + j); vals.append(-1.0)
                r += 1
    m = r
    A = sparse.csr_matrix((vals, (rows, cols)), shape=(m, n))
    b = np.concatenate([np.ones(2 * nf), np.zeros(m - 2 * nf)])
    c = np.zeros(n)
    for (i, k) in pairs:
        for (j, l) in jl:
            c[yv(i, j, k, l)] = 2.0 * F[i, k] * D[j, l]   # both orders of the pair
    return _finish(A, b, b, np.zeros(n), np.full(n, np.inf), c, "QAP relaxation (nug-like)")


def pde_control_lp(N, seed, alpha=0.01):
    """cont-like: boundary control of the Poisson equation on an N x N grid with an L1 tracking objective and a state
    constraint (the construction of Mittelmann's cont-xx LPs).  Variables: states y (N^2), controls u on the boundary cells
    (4 N - 4, -1 <= u <= 1), and t >= |y - y_d| (N^2).  Rows: the five-point stencil 4 y_ij - sum of neighbours = h^2 f + boundary
    control (equalities, one per cell), t - y >= -y_d and t + y >= y_d.  Bounds y <= psi.  min h^2 sum t + alpha h sum |u| is
    taken with u split by its sign bound instead: cost alpha h on u^+ and u^-.  Grid stencil: banded, 5 entries per row."""
    rng = np.random.default_rng(seed)
    h = 1.0 / (N + 1)
    ny = N * N
    bcells = [(i, j) for i in range(N) for j in range(N) if i in (0, N - 1) or j in (0, N - 1)]
    nb = len(bcells)
    n = ny + 2 * nb + ny          # y, u+, u-, t
    oy, oup, oum, ot = 0, ny, ny + nb, ny + 2 * nb
    xs = (np.arange(N) + 1) * h
    X, Y = np.meshgrid(xs, xs, indexing="ij")
    yd = (np.sin(2 * np.pi * X) * np.sin(np.pi * Y) + 0.3 * rng.normal(size=(N, N)) * 0.1).ravel()
    f = (5.0 * np.exp(-20 * ((X - 0.3) ** 2 + (Y - 0.6) ** 2))).ravel()
    rows, cols, vals = [], [], []
    cell = lambda i, j: i * N + j
    for i in range(N):
        for j in range(N):
            r = cell(i, j)
            rows.append(r); cols.append(oy + r); vals.append(4.0)
            for (a, b2) in ((i - 1, j), (i + 1, j), (i, j - 1), (i, j + 1)):
                if 0 <= a < N and 0 <= b2 < N:
                    rows.append(r); cols.append(oy + cell(a, b2)); vals.append(-1.0)
    for q, (i, j) in enumerate(bcells):   # the control enters the boundary cells' equations
        r = cell(i, j)
        rows += [r, r]; cols += [oup + q, oum + q]; vals += [-1.0, 1.0]
    r0 = ny
    for k in range(ny):
        rows += [r0 + k, r0 + k]; cols += [ot + k, oy + k]; vals += [1.0, -1.0]
    r1 = 2 * ny
    for k in range(ny):
        rows += [r1 + k, r1 + k]; cols += [ot + k, oy + k]; vals += [1.0, 1.0]
    m = 3 * ny
    A = sparse.csr_matrix((vals, (rows, cols)), shape=(m, n))
    AL = np.concatenate([h * h * f, -yd, yd])
    AU = np.concatenate([h * h * f, np.full(2 * ny, np.inf)])
    psi = 0.6
    l = np.concatenate([np.full(ny, -np.inf), np.zeros(2 * nb), np.zeros(ny)])
    u = np.concatenate([np.full(ny, psi), np.ones(2 * nb), np.full(ny, np.inf)])
    c = np.concatenate([np.zeros(ny), np.full(2 * nb, alpha * h), np.full(ny, h * h)])
    return _finish(A, AL, AU, l, u, c, "PDE boundary control (cont-like)")


def staircase_lp(stages, rows_per_stage, cols_per_stage, per_row, seed):
    """Staircase (multi-stage / dynamic) LP: stage t has its own variables x_t >= 0 and rows B_t x_{t-1} + A_t x_t = b_t -- the
    matrix is a staircase of (rows_per_stage x cols_per_stage) blocks on the diagonal and the sub-diagonal, `per_row` entries
    per row and block.  b_t comes from a nonnegative point (feasible), costs are nonnegative (bounded), a fifth of the
    variables carry upper bounds."""
    rng = np.random.default_rng(seed)
    T, ms, ns = stages, rows_per_stage, cols_per_stage
    m, n = T * ms, T * ns
    rows, cols, vals = [], [], []
    for t in range(T):
        r = np.repeat(np.arange(t * ms, (t + 1) * ms), per_row)
        cA = t * ns + rng.integers(0, ns, size=len(r))
        rows.append(r); cols.append(cA); vals.append(rng.uniform(0.2, 1.5, len(r)) * rng.choice([-1.0, 1.0], len(r), p=[0.3, 0.7]))
        if t > 0:
            k = max(1, per_row // 2)
            r2 = np.repeat(np.arange(t * ms, (t + 1) * ms), k)
            cB = (t - 1) * ns + rng.integers(0, ns, size=len(r2))
            rows.append(r2); cols.append(cB); vals.append(-rng.uniform(0.2, 1.0, len(r2)))
    A = sparse.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(m, n))
    A.sum_duplicates()
    x0 = np.where(rng.random(n) < 0.6, rng.uniform(0.0, 2.0, n), 0.0)
    b = A @ x0
    u = np.where(rng.random(n) < 0.2, x0 + rng.uniform(0.5, 2.0, n), np.inf)
    c = rng.uniform(0.1, 2.0, n)
    return _finish(A, b, b, np.zeros(n), u, c, "staircase (multi-stage)")


# the four families at test size (<= 1e5 nonzeros) and at table size (1e6 - 1e7)
FAMILIES_SMALL = {
    "pds_like": lambda: multicommodity_flow_lp(12, 8, 41),
    "nug_like": lambda: qap_lp_relaxation(6, 42),
    "cont_like": lambda: pde_control_lp(40, 43),
    "staircase": lambda: staircase_lp(12, 300, 450, 6, 44),
}
FAMILIES_LARGE = {
    "pds_like": lambda: multicommodity_flow_lp(110, 40, 41),
    "nug_like": lambda: qap_lp_relaxation(30, 42),     # 52 260 x 379 350, 1.57e6 nonzeros: the size of nug30
    "cont_like": lambda: pde_control_lp(700, 43),
    "staircase": lambda: staircase_lp(200, 3000, 4500, 8, 44),
}
