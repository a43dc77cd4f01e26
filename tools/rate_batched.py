#!/usr/bin/env python3
"""Config 4 (bench.py's recipe: config-3 matrix, 64 perturbed members) at a fixed 1500 iterations: batch-iterations/s, three
runs.  Developer A/B of the batched kernels (HPRLP_LIB picks the library, HPRLP_BATCH_CHUNK the chunk width)."""
import os

os.environ.setdefault("HPRLP_TEST_HOOKS", "1")  # the HPRLP_* switches used here are test hooks (csrc/env.h)
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

H, G = bench.H, bench.G
os.dup2(2, 1)
lp = G.c3_pds20_like()
B, iters = int(os.environ.get("RATE_B", "64")), 1500
rng = np.random.default_rng(4)
m, n = lp["m"], lp["n"]
Cm = lp["c"][:, None] * (1 + 0.1 * rng.normal(size=(n, B)))
AU = lp["AU"][:, None] + np.abs(rng.normal(scale=0.1, size=(m, B)))
AL = np.repeat(lp["AL"][:, None], B, axis=1)
AL = np.where(np.isfinite(AL), np.minimum(AL, AU), AL)
L = np.repeat(lp["l"][:, None], B, axis=1)
U = np.repeat(lp["u"][:, None], B, axis=1)
U = np.where(np.isfinite(U), U, 50.0)
model = H.Model.from_csr(m, n, lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])
rates = []
for rep in range(3):
    rb = H.solve_batched(model, Cm, AL, AU, L, U, None, H.Parameters(stop_tol=1e-30, max_iter=iters, use_presolve=False))
    rates.append(iters / rb["solve_time"])
chk = float(np.abs(rb["x"]).sum())
print("chunk %s B %d: batch-it/s %s   |x|_1 %.12e" % (os.environ.get("HPRLP_BATCH_CHUNK", "default"), B,
                                                       " ".join("%7.0f" % r for r in rates), chk), file=sys.stderr, flush=True)
model.free()
