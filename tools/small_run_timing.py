#!/usr/bin/env python3
"""bench.py's side_configs leg for config 2 / 3, repeated: where its time-to-tolerance figure comes from (developer check)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

H, G = bench.H, bench.G
os.dup2(2, 1)
for name, lp in (("c2", G.c2_25fv47_like()), ("c3", G.c3_pds20_like())):
    model = H.Model.from_csr(lp["m"], lp["n"], lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])
    for rep in range(4):
        t0 = time.time()
        s = H.Solver(model, H.Parameters(stop_tol=1e-4, use_presolve=False))
        t1 = time.time(); s.scale(); t2 = time.time()
        lam, it = s.power_iteration(); t3 = time.time()
        s.init(-1.0, lam * 1.01); t4 = time.time()
        r = s.run(); t5 = time.time()
        print("RUN %s rep %d: create %.4f scale %.4f power %.4f (%d its) init %.4f run %.4f s (%d iterations, reported %.4f) -> total %.4f" %
              (name, rep, t1 - t0, t2 - t1, t3 - t2, it, t4 - t3, t5 - t4, r.iter, r.time, t5 - t0), file=sys.stderr)
        s.close()
    model.free()
