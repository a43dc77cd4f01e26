#!/usr/bin/env python3
"""Graph-replay iteration rate of config 3's stand-in (stream kernel) and of a few mid-size planted LPs: developer A/B of
launch-bound changes (HPRLP_LIB picks the library)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

H, G = bench.H, bench.G
os.dup2(2, 1)
cases = [("c3", G.c3_pds20_like()), ("planted_2e4x5e4", G.planted_lp(20000, 50000, 400000, 3)),
         ("planted_1e5x2e5", G.planted_lp(100000, 200000, 2000000, 4))]
for name, lp in cases:
    model = H.Model.from_csr(lp["m"], lp["n"], lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])
    s = H.Solver(model, H.Parameters(stop_tol=1e-4, use_presolve=False))
    s.scale()
    lam, _ = s.power_iteration()
    s.init(-1.0, lam * 1.01)
    rates = []
    for rep in range(3):
        t = s.time_iterations(200, 2000, 0)
        rates.append(2000 / (t["total_ms"] * 1e-3))
    print("%-18s it/s %s" % (name, " ".join("%8.0f" % r for r in rates)), file=sys.stderr, flush=True)
    s.close(); model.free()
