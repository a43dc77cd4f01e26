#!/usr/bin/env python3
"""Iterations of solve() on one test LP with the presolve stages switched on and off (developer check)."""
import os

os.environ.setdefault("HPRLP_TEST_HOOKS", "1")  # the HPRLP_* switches used here are test hooks (csrc/env.h)
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import test_presolve as T  # noqa: E402
from conftest import hprlp  # noqa: E402

seed, fs = int(sys.argv[1]), float(sys.argv[2])
tol = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-7
lp = T.doubleton_lp(seed, m0=200, n0=320, pairs=30, free_share=fs)
f0, *_ = T.highs(lp["m"], lp["n"], lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])
model = T.make_model(lp)
os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
for name, env, pre in (("presolve off", "", False), ("all stages", "", True), ("no bounds", "bounds", True), ("no doubleton", "doubleton", True),
                       ("reductions only", "bounds,doubleton", True)):
    os.environ["HPRLP_PRESOLVE_OFF"] = env
    r = model.solve(hprlp.Parameters(stop_tol=tol, use_presolve=pre, max_iter=400000))
    k = hprlp.original_kkt(model, r.x, r.y, r.z)
    print("%-16s %-10s iter %7d  obj err %.2e  kkt %.2e" % (name, r.status, r.iter, abs(r.primal_obj - f0) / (1 + abs(f0)),
                                                          max(k["primal_feas"], k["dual_feas"], k["gap"])), file=sys.stderr)
