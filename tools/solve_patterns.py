#!/usr/bin/env python3
"""Whole solves (tol 1e-4, through Model.solve = the C ABI's solve()) of planted LPs on patterns of tools/form_regret.py: status,
iterations, objective against the planted optimum, KKT errors of the returned point recomputed here on the model as given, the kernel
forms.  What the regret table does not show: that scaling, power iteration, restarts and the solution's way back (locality ordering)
work on the forms the selection picks for these shapes.

    python tools/solve_patterns.py PATTERN[,PATTERN..]      (GPU box)
"""
import os
import sys
import time

import numpy as np
from scipy import sparse

os.environ.setdefault("HPRLP_TEST_HOOKS", "1")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import importlib.util
spec = importlib.util.spec_from_file_location("fr", os.path.join(os.path.dirname(os.path.abspath(__file__)), "form_regret.py"))
fr = importlib.util.module_from_spec(spec); spec.loader.exec_module(fr)
bench, H = fr.bench, fr.H

real = os.dup(1); os.dup2(2, 1)
out = lambda s: os.write(real, (s + "\n").encode())
for name in sys.argv[1].split(","):
    gen = fr.HELD_OUT.get(name) or fr.HELD_OUT_2.get(name) or fr.HELD_OUT_3.get(name) or fr.CORPUS.get(name)
    A = gen().tocsr(); A.sort_indices()
    A.data = np.random.default_rng(7).normal(size=A.nnz)
    m, n = A.shape
    lp = bench.planted_on(m, n, A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64))
    model = H.Model.from_csr(m, n, lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])
    t0 = time.time()
    r = model.solve(H.Parameters(stop_tol=1e-4, max_iter=60000, use_presolve=False, time_limit=120.0))
    wall = time.time() - t0
    x, y = np.asarray(r.x), np.asarray(r.y)
    Ax = A @ x
    b = np.maximum(np.where(np.isfinite(lp["AL"]), np.abs(lp["AL"]), 0), np.where(np.isfinite(lp["AU"]), np.abs(lp["AU"]), 0))
    viol = np.maximum(np.maximum(np.where(np.isfinite(lp["AL"]), lp["AL"] - Ax, 0), np.where(np.isfinite(lp["AU"]), Ax - lp["AU"], 0)), 0)
    z = lp["c"] - A.T @ y
    zl = np.where(np.isfinite(lp["l"]), np.maximum(z, 0), 0); zu = np.where(np.isfinite(lp["u"]), np.minimum(z, 0), 0)
    rd = z - zl - zu
    rp_rel = np.linalg.norm(viol) / (1 + np.linalg.norm(b)); rd_rel = np.linalg.norm(rd) / (1 + np.linalg.norm(lp["c"]))
    obj = float(lp["c"] @ x)
    out("%-26s %8d x %8d nnz %9d | %-10s it %6d  %.2f s | obj rel err %.2e | primal %.1e dual %.1e | bounds %.1e" % (
        name, m, n, A.nnz, r.status, r.iter, wall, abs(obj - lp["obj_star"]) / (1 + abs(lp["obj_star"])), rp_rel, rd_rel,
        max(float(np.max(lp["l"] - x, initial=0)), float(np.max(x - lp["u"], initial=0)))))
    model.free()
