#!/usr/bin/env python3
"""Where the wall time of a whole solve() goes on the small configs (2 and 3): phases from HPRLP_TIMING.  Developer check."""
import os

os.environ.setdefault("HPRLP_TEST_HOOKS", "1")  # the HPRLP_* switches used here are test hooks (csrc/env.h)
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

H, G = bench.H, bench.G
os.dup2(2, 1)
for name, lp in (("c2", G.c2_25fv47_like()), ("c3", G.c3_pds20_like())):
    model = H.Model.from_csr(lp["m"], lp["n"], lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])
    for rep in range(3):
        t0 = time.time()
        r = model.solve(H.Parameters(stop_tol=1e-4, use_presolve=False))
        print("SOLVE %s rep %d: wall %.4f s, reported time %.4f s, iterations %d, status %s" % (name, rep, time.time() - t0, r.time, r.iter, r.status), file=sys.stderr)
    model.free()
for name, lp in (("c2", G.c2_25fv47_like()), ("c3", G.c3_pds20_like())):
    model = H.Model.from_csr(lp["m"], lp["n"], lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])
    for rep in range(2):
        s = H.Solver(model, H.Parameters(use_presolve=False))
        t0 = time.time(); s.scale(); t1 = time.time(); lam, it = s.power_iteration(); t2 = time.time()
        sc = s.scalars()
        print("PHASES %s: setup %.4f s, scaling %.4f s (wall %.4f), power iteration %.4f s / %d its (wall %.4f)" %
              (name, sc["setup_time"], sc["scaling_time"], t1 - t0, sc["power_time"], it, t2 - t1), file=sys.stderr)
        s.close()
    model.free()
