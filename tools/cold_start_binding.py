#!/usr/bin/env python3
"""Cold start through the REFERENCE's own Python binding (oracle/_ref/_hprlp_core*.so: bindings/python/src/hprlp_pybind.cpp compiled
unchanged on top of lib/libhprlp.so): what a user of the reference pays for the first and the following solve() calls of a fresh
process -- the drop-in case, which cannot call hprlp_warmup().  Config 2 stand-in (821 x 1571) by default.
    python tools/cold_start_binding.py [c2|c3]"""
import glob
import importlib.util
import os
import sys
import time

t_start = time.time()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "hpr-lp-c_amd"))
import lpgen as G  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "c2"
lp = G.c2_25fv47_like() if which == "c2" else G.c3_pds20_like()
cand = glob.glob(os.path.join(ROOT, "oracle", "_ref", "_hprlp_core*.so"))
if not cand:
    raise SystemExit("oracle/_ref/_hprlp_core*.so is not built (make -C oracle refbinding, where /root/reference exists)")
os.dup2(2, 1)
t0 = time.time()
spec = importlib.util.spec_from_file_location("_hprlp_core", cand[0])
core = importlib.util.module_from_spec(spec)
spec.loader.exec_module(core)
t_import = time.time() - t0
t0 = time.time()
model = core.create_model_from_arrays(lp["m"], lp["n"], len(lp["values"]), lp["rowptr"].astype(np.int32), lp["colind"].astype(np.int32),
                                      lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"], False)
t_model = time.time() - t0
print("BINDING import %.4f s (loads lib/libhprlp.so), create_model_from_arrays %.4f s (python start-up + numpy + generator before: %.2f s)"
      % (t_import, t_model, t0 - t_start - t_import), file=sys.stderr)
p = core.Parameters()
p.stop_tol = 1e-4
p.use_presolve = False
for rep in range(4):
    t0 = time.time()
    r = core.solve(model, p)
    w = time.time() - t0
    print("BINDING SOLVE %s rep %d: wall %.4f s | HPRLP_results.time %.4f | iterations %d %s" % (which, rep, w, r.time, r.iter, r.status), file=sys.stderr)
core.free_model(model)
