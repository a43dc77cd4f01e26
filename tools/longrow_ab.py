#!/usr/bin/env python3
"""Mid-size LP (100 k x 300 k, ~1 M nonzeros) with a few columns of `L` entries: what such rows cost the stream kernel per
half-step.  usage: python tools/longrow_ab.py [L ...]   (developer check)"""
import os
import sys

import numpy as np
from scipy import sparse

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

H, G = bench.H, bench.G
os.dup2(2, 1)
if os.environ.get("LR_BASE") == "banded2m":  # a matrix the tiled kernel takes: 2M x 2M, 20 per row, band 2e4
    base = bench.banded_lp(2_000_000, 2_000_000, 20, 20_000)
    base.update(m=2_000_000, n=2_000_000)
else:
    base = G.planted_lp(100_000, 300_000, 1_000_000, 7, values="general", dense_col_frac=0.0)
m, n = base["m"], base["n"]
A0 = sparse.csr_matrix((base["values"], base["colind"], base["rowptr"]), shape=(m, n))
rng = np.random.default_rng(1)
for L in [int(a) for a in sys.argv[1:]] or [0, 200, 1000, 3000, 10000]:
    A = A0.tolil(copy=True) if False else A0.copy()
    if L > 0:
        cols = rng.choice(n, size=int(os.environ.get("LR_NCOLS", "5")), replace=False)
        rows = np.concatenate([rng.choice(m, size=L, replace=False) for _ in cols])
        cc = np.repeat(cols, L)
        A = (A + sparse.csr_matrix((rng.normal(size=len(rows)), (rows, cc)), shape=(m, n))).tocsr()
        A.sort_indices()
    x0 = np.abs(rng.normal(size=n))
    b = A @ x0
    model = H.Model.from_csr(m, n, A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data, b - 1, b + 1, np.zeros(n), np.full(n, 10.0),
                             rng.normal(size=n))
    s = H.Solver(model, H.Parameters(use_presolve=False))
    import time as _t
    _t0 = _t.time()
    s.scale()
    print("   entries: scale() %.4f s" % (_t.time() - _t0), file=sys.stderr)
    s.init(0.7, 1.3)
    big = m > 1_000_000
    t = s.time_iterations(20 if big else 200, 200 if big else 5000, 0)
    t1 = s.time_iterations(10 if big else 50, 100 if big else 1000, 1)
    t2 = s.time_iterations(5, 200, 2)
    print("   entries: bare SpMV A^T y %.2f us, A x_hat %.2f us per launch; info %s" % (1e3 * t2["xhalf_ms"] / 200, 1e3 * t2["yhalf_ms"] / 200, s.info()), file=sys.stderr)
    print("5 columns of %6d entries: %6.2f us/iteration; eager x-half %6.2f us, y-half %6.2f us" %
          (L, 1e3 * t["total_ms"] / (200 if big else 5000), 1e3 * t1["xhalf_ms"] / (100 if big else 1000), 1e3 * t1["yhalf_ms"] / (100 if big else 1000)), file=sys.stderr)
    s.close(); model.free()
