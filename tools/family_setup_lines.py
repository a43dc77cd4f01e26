#!/usr/bin/env python3
"""Set-up decisions of the library (HPRLP_TIMING=1 lines: heights, tile widths, line density, forms) for ladder points / families.
usage: python tools/family_setup_lines.py POINT [POINT ...]   (bench.py: LADDER_POINTS / FAMILY_POINTS keys)"""
import os

os.environ.setdefault("HPRLP_TEST_HOOKS", "1")  # the HPRLP_* switches used here are test hooks (csrc/env.h)
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["HPRLP_TIMING"] = "1"
import bench as B  # noqa: E402
H = B.H
os.dup2(2, 1)
for name in sys.argv[1:]:
    lp = (B.LADDER_POINTS.get(name) or B.FAMILY_POINTS[name])()
    print("=== %s" % name, file=sys.stderr)
    model = H.Model.from_csr(lp["m"], lp["n"], lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])
    s = H.Solver(model, H.Parameters(use_presolve=False))
    print("FORMS " + s.describe(), file=sys.stderr)
    s.close()
    model.free()
