#!/usr/bin/env python3
"""Independent optima (HiGHS through scipy.optimize.linprog) of the four family generators of hpr-lp-c_amd/lpgen.py at test size
(FAMILIES_SMALL) -> tests/golden/family_optima.json.  Run in the build container: python tests/golden/make_family_optima.py"""
import json
import os
import sys

import numpy as np
from scipy import sparse
from scipy.optimize import linprog

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from conftest import lpgen  # noqa: E402

out = {}
for name, make in lpgen.FAMILIES_SMALL.items():
    lp = make()
    A = lp["A"]
    eq = np.isfinite(lp["AL"]) & (lp["AL"] == lp["AU"])
    lo = np.isfinite(lp["AL"]) & ~eq
    hi = np.isfinite(lp["AU"]) & ~eq
    Aub = sparse.vstack([A[hi], -A[lo]]) if (hi.any() or lo.any()) else None
    bub = np.concatenate([lp["AU"][hi], -lp["AL"][lo]]) if Aub is not None else None
    bounds = list(zip(np.where(np.isfinite(lp["l"]), lp["l"], None), np.where(np.isfinite(lp["u"]), lp["u"], None)))
    r = linprog(lp["c"], A_ub=Aub, b_ub=bub, A_eq=A[eq] if eq.any() else None, b_eq=lp["AL"][eq] if eq.any() else None, bounds=bounds,
                method="highs")
    assert r.status == 0, (name, r.message)
    out[name] = {"m": lp["m"], "n": lp["n"], "nnz": int(len(lp["values"])), "objective": float(r.fun), "family": lp["family"],
                 "checksum_values": float(np.sum(lp["values"])), "checksum_c": float(np.sum(lp["c"]))}
    print(name, out[name])
json.dump(out, open(os.path.join(HERE, "family_optima.json"), "w"), indent=1)
