#!/usr/bin/env python3
"""solve() with the presolve chain on, over LPs of the test families, against the exact optimum (HiGHS): status, objective
error, original-model KKT, and how often the safety net (solve of the model as given) had to step in.  Developer check."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import test_presolve as T  # noqa: E402
from conftest import hprlp  # noqa: E402

lo, hi = int(sys.argv[1]), int(sys.argv[2])
tol = 1e-6


def gen(seed):
    kind = seed % 3
    if kind == 0:
        return T.decorated_lp(seed)
    if kind == 1:
        lp = T.structured_lp(seed, m0=100 + seed % 150, n0=150 + seed % 200)
        pick = np.random.default_rng(seed).random(lp["n"]) < 0.3
        lp["u"] = np.where(pick & (lp["l"] < lp["u"]), np.inf, lp["u"])
        return lp
    return T.doubleton_lp(seed, m0=150 + seed % 100, n0=250 + seed % 150, pairs=10 + seed % 30, free_share=[0.0, 0.5, 0.3][(seed // 3) % 3])


ran = bad = net = limit_on = limit_off = 0
for seed in range(lo, hi):
    lp = gen(seed)
    try:
        f0, *_ = T.highs(lp["m"], lp["n"], lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])
    except AssertionError:
        continue
    model = T.make_model(lp)
    sys.stdout.flush()
    with tempfile.TemporaryFile(mode="w+") as tf:
        keep = os.dup(1)
        os.dup2(tf.fileno(), 1)
        try:
            r = model.solve(hprlp.Parameters(stop_tol=tol, use_presolve=True, max_iter=300000))
        finally:
            sys.stdout.flush()
            os.dup2(keep, 1)
            os.close(keep)
        tf.seek(0)
        out = tf.read()
    ran += 1
    fell = "solving the original model" in out
    net += fell
    k = hprlp.original_kkt(model, r.x, r.y, r.z)
    err = max(k["primal_feas"], k["dual_feas"], k["gap"])
    oerr = abs(r.primal_obj - f0) / (1 + abs(f0))
    if r.status != "OPTIMAL":
        limit_on += 1
    if r.status == "OPTIMAL" and (err > 20 * tol or oerr > 1e-3):
        bad += 1
        print("BAD", seed, r.status, r.iter, "kkt %.2e obj err %.2e" % (err, oerr), "safety net" if fell else "", file=sys.stderr)
    model.free()
print(f"seeds {lo}..{hi}: ran {ran}, OPTIMAL-but-wrong {bad}, not converged in 300k iterations {limit_on}, safety net used {net}", file=sys.stderr)
