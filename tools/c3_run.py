#!/usr/bin/env python3
"""BASELINE config 3 alone (33 874 x 105 728, 244 k nnz; fixed number of iterations): for rocprofv3 runs."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

H, G = bench.H, bench.G
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
lp = G.c3_pds20_like()
model = H.Model.from_csr(lp["m"], lp["n"], lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])
os.dup2(2, 1)
s = H.Solver(model, H.Parameters(use_presolve=False))
s.scale()
lam, _ = s.power_iteration()
s.init(-1.0, lam * 1.01)
t = s.time_iterations(200, iters, 0)
print("config 3: %.2f us per iteration (graph replay), %.0f it/s; info %s" % (1e3 * t["total_ms"] / iters, iters / (t["total_ms"] * 1e-3), s.info()), file=sys.stderr)
t1 = s.time_iterations(50, 2000, 1)
print("eager with events: x-half %.2f us, y-half %.2f us" % (1e3 * t1["xhalf_ms"] / 2000, 1e3 * t1["yhalf_ms"] / 2000), file=sys.stderr)
