#!/usr/bin/env python3
"""x- / y-half launch times (eager, HIP events) on config-3-sized planted LPs with and without dense columns, and transposed
shape: what the 11 us of the config-3 x-half depend on.  Developer check."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

H, G = bench.H, bench.G
os.dup2(2, 1)
for name, lp in (("c3 (0.05 % dense columns)", G.planted_lp(33874, 105728, 230200, 3, values="network", dense_col_frac=0.0005)),
                 ("c3 without dense columns", G.planted_lp(33874, 105728, 230200, 3, values="network", dense_col_frac=0.0)),
                 ("transposed shape 105728 x 33874", G.planted_lp(105728, 33874, 230200, 3, values="network", dense_col_frac=0.0))):
    model = H.Model.from_csr(lp["m"], lp["n"], lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])
    s = H.Solver(model, H.Parameters(use_presolve=False))
    s.scale()
    s.init(0.7, 1.3)
    t = s.time_iterations(200, 10000, 0)
    t1 = s.time_iterations(50, 2000, 1)
    i = s.info()
    print("%-36s %6.2f us/iteration; eager x-half %5.2f us, y-half %5.2f us; blocks A %d, A^T %d" %
          (name, 1e3 * t["total_ms"] / 10000, 1e3 * t1["xhalf_ms"] / 2000, 1e3 * t1["yhalf_ms"] / 2000, i["blocks_A"], i["blocks_AT"]), file=sys.stderr)
    s.close(); model.free()
