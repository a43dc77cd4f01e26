#!/usr/bin/env python3
"""Cold start: what the FIRST solve of a process costs against the following ones (config 2 stand-in by default), itemised.

Per solve: wall of the whole call and the library's own phase table (hprlp_last_solve_phases); HPRLP_TIMING=1 in the
environment adds the set-up phases on stderr.  Run it in a fresh process:
    python tools/cold_start.py [c2|c3] [warm]     ("warm": call hprlp_warmup() first -- what a caller can do at start-up)
"""
import ctypes as C
import os

os.environ.setdefault("HPRLP_TEST_HOOKS", "1")  # the HPRLP_* switches used here are test hooks (csrc/env.h)
import sys
import time

t_imp = time.time()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

H, G = bench.H, bench.G
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
warm = len(sys.argv) > 2 and sys.argv[2] == "warm"
lp = G.c2_25fv47_like() if which == "c2" else G.c3_pds20_like()
os.dup2(2, 1)
t0 = time.time()
L = H.lib()
t_load = time.time() - t0
t_warm = 0.0
if warm:
    t0 = time.time()
    L.hprlp_warmup.restype = C.c_int
    rc = L.hprlp_warmup(0)
    t_warm = time.time() - t0
    print("WARMUP rc %d: %.4f s" % (rc, t_warm), file=sys.stderr)
t0 = time.time()
model = H.Model.from_csr(lp["m"], lp["n"], lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])
t_model = time.time() - t0
print("LOAD library %.4f s, create_model %.4f s (python imports before: %.2f s)" % (t_load, t_model, t0 - t_imp), file=sys.stderr)
RECORDS = {"config": which, "explicit_warmup": warm, "library_load_s": t_load, "warmup_s": t_warm, "create_model_s": t_model, "solves": []}
for rep in range(4):
    t0 = time.time()
    r = model.solve(H.Parameters(stop_tol=1e-4, use_presolve=False))
    w = time.time() - t0
    ph = H.last_solve_phases()
    print("SOLVE %s rep %d: wall %.4f s | reported time %.4f | iterations %d %s | phases: setup %.4f scaling %.4f power %.4f loop %.4f solution %.4f teardown %.4f"
          % (which, rep, w, r.time, r.iter, r.status, ph["device_setup"], ph["scaling"], ph["power_iteration"], ph["loop"], ph["collect_solution"], ph["teardown"]),
          file=sys.stderr)
    RECORDS["solves"].append({"whole_call_wall_s": w, "reported_time_s": r.time, "iterations": r.iter, "status": r.status, "phases_s": ph})
model.free()
if os.environ.get("HPRLP_COLD_START_JSON"):
    import json
    print("COLDJSON " + json.dumps(RECORDS), file=sys.stderr)
