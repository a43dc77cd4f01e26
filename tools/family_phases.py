#!/usr/bin/env python3
"""Where does a whole solve of the family LPs (lpgen.FAMILIES_LARGE) spend its time?  Phase table of the library
(hprlp_last_solve_phases), the rate of bare normal iterations by graph replay, and what that leaves for the check / log steps.
    python tools/family_phases.py [name ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

H, G = bench.H, bench.G
os.dup2(2, 1)
names = sys.argv[1:] or list(G.FAMILIES_LARGE)
for name in names:
    lp = G.FAMILIES_LARGE[name]()
    model = H.Model.from_csr(lp["m"], lp["n"], lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])
    prm = H.Parameters(stop_tol=1e-4, use_presolve=False, max_iter=200000)
    model.solve(prm)  # warm
    r = model.solve(prm)
    ph = H.last_solve_phases()
    s = H.Solver(model, prm)
    s.scale(); lam, pit = s.power_iteration(); s.init(-1.0, lam * 1.01)
    g = s.time_iterations(50, 500, 0)
    e = s.time_iterations(50, 500, 1)
    s.close()
    it_us = 1e3 * g["total_ms"] / 500
    loop = ph["loop"]
    print("PHASES %-10s %d x %d nnz %d | %s %d iterations | setup %.4f scaling %.4f power %.4f (%d its) loop %.4f solution %.4f | graph-replay iteration %.1f us (event windows x %.1f + y %.1f) "
          "-> normal iterations %.4f s of the loop's %.4f s: %.0f %% | whole-solve rate %.0f it/s, bare rate %.0f it/s"
          % (name, lp["m"], lp["n"], len(lp["values"]), r.status, r.iter, ph["device_setup"], ph["scaling"], ph["power_iteration"], pit, loop, ph["collect_solution"],
             it_us, 1e3 * e["xhalf_ms"] / 500, 1e3 * e["yhalf_ms"] / 500, r.iter * it_us * 1e-6, loop, 100 * r.iter * it_us * 1e-6 / loop, r.iter / r.time, 1e6 / it_us), file=sys.stderr)
    model.free()
