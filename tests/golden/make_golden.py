"""Generates tests/golden/known_lps.json: independent optima (HiGHS via scipy.optimize.linprog) of the
only LPs the reference documents an answer for -- data/model.mps (x=(2.8,3.6), obj=-26.4, reference
examples/cpp/example_direct_lp.cpp:14) and the three members of examples/c/example_batched_lp.c:37-62.
Run:  python tests/golden/make_golden.py
The reference itself cannot be executed here (CUDA only), so these vectors are data, not its output.
"""
import json
import os

import numpy as np
from scipy.optimize import linprog

A = np.array([[1.0, 2.0], [3.0, 1.0]])
cases = [
    dict(name="model_mps", c=[-3, -5], AU=[10, 12], u=[np.inf, np.inf]),
    dict(name="batched_k1", c=[-2, -6], AU=[9, 13], u=[np.inf, np.inf]),
    dict(name="batched_k2", c=[-4, -4], AU=[11, 11], u=[4, np.inf]),
]
out = []
for cs in cases:
    r = linprog(cs["c"], A_ub=A, b_ub=cs["AU"], bounds=[(0, None if np.isinf(ub) else ub) for ub in cs["u"]], method="highs")
    assert r.status == 0
    y = -r.ineqlin.marginals * -1.0           # HPRLP sign: c = A'y + z, y <= 0 on active upper row bounds
    z = np.array(cs["c"]) - A.T @ y
    out.append(dict(name=cs["name"], m=2, n=2, rowptr=[0, 2, 4], colind=[0, 1, 0, 1], values=[1, 2, 3, 1],
                    AL=["-inf", "-inf"], AU=cs["AU"], l=[0, 0], u=[("inf" if np.isinf(v) else v) for v in cs["u"]],
                    c=cs["c"], obj=r.fun, x=list(r.x), y=list(y), z=list(z)))
json.dump(out, open(os.path.join(os.path.dirname(__file__), "known_lps.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
