#!/usr/bin/env python3
"""BASELINE config 4 alone (solve_batched, shared A = config-3 matrix, B = 64, fixed iterations): for rocprofv3 runs."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

H, G = bench.H, bench.G
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 600
lp = G.c3_pds20_like() if os.environ.get("C4_NO_DENSE") != "1" else G.planted_lp(33874, 105728, 230200, 3, values="network", dense_col_frac=0.0)  # C4_NO_DENSE=1: the same shape without the 53 columns of ~205 entries
model = H.Model.from_csr(lp["m"], lp["n"], lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])
B = 64
rng = np.random.default_rng(4)
m, n = lp["m"], lp["n"]
Cm = lp["c"][:, None] * (1 + 0.1 * rng.normal(size=(n, B)))
AU = lp["AU"][:, None] + np.abs(rng.normal(scale=0.1, size=(m, B)))
AL = np.repeat(lp["AL"][:, None], B, axis=1)
AL = np.where(np.isfinite(AL), np.minimum(AL, AU), AL)
L = np.repeat(lp["l"][:, None], B, axis=1)
U = np.where(np.isfinite(lp["u"]), lp["u"], 50.0)[:, None].repeat(B, axis=1)
os.dup2(2, 1)
rb = H.solve_batched(model, Cm, AL, AU, L, U, None, H.Parameters(stop_tol=1e-30, max_iter=iters, use_presolve=False))
nnz = len(lp["values"])
print("batch iterations/s %.0f  (solve %.3f s, %d iterations)" % (iters / rb["solve_time"], rb["solve_time"], iters), file=sys.stderr)
