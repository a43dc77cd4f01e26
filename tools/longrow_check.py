#!/usr/bin/env python3
"""A banded matrix large enough for the FUSED tiled kernel (more than 512 super-blocks) with a few very long rows and columns:
the iterates of the tiled path (long rows aside, hand-off on) against those of the stream kernel on the same model.
usage: python tools/longrow_check.py [rows_in_millions]   (developer check)"""
import os

os.environ.setdefault("HPRLP_TEST_HOOKS", "1")  # the HPRLP_* switches used here are test hooks (csrc/env.h)
import sys

import numpy as np
from scipy import sparse

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

H = bench.H
os.dup2(2, 1)
M = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 5_000_000
base = bench.banded_lp(M, M, 12, 50_000)
A = sparse.csr_matrix((base["values"], base["colind"], base["rowptr"]), shape=(M, M))
rng = np.random.default_rng(2)
rr, cc, vv = [], [], []
for j, L in zip(rng.choice(M, 4, replace=False), (3000, 9000, 30000, 100000)):   # dense columns
    r = rng.choice(M, L, replace=False); rr.append(r); cc.append(np.full(L, j)); vv.append(rng.normal(size=L) * 0.01)
for i, L in zip(rng.choice(M, 3, replace=False), (2000, 5000, 50000)):           # dense rows
    c = rng.choice(M, L, replace=False); rr.append(np.full(L, i)); cc.append(c); vv.append(rng.normal(size=L) * 0.01)
A = (A + sparse.csr_matrix((np.concatenate(vv), (np.concatenate(rr), np.concatenate(cc))), shape=(M, M))).tocsr()
A.sort_indices()
x0 = np.abs(rng.normal(size=M))
b = A @ x0
args = (M, M, A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data, b - 1, b + 1, np.zeros(M), np.full(M, 10.0), rng.normal(size=M))
out = {}
for name, env in (("tiled", {}), ("stream", {"HPRLP_NO_TILED": "1"})):
    for k, v in env.items():
        os.environ[k] = v
    model = H.Model.from_csr(*args)
    s = H.Solver(model, H.Parameters(use_presolve=False))
    info = s.info()
    s.scale()
    lam, it = s.power_iteration(max_iter=40)
    s.init(0.7, 1.3 * lam)
    s.iterate(25, True)
    t = s.time_iterations(5, 50, 1)
    out[name] = dict(info=info, lam=lam, x=s.get("x"), y=s.get("y"), xb=s.get("x_bar"), res=s.residuals(26, True))
    print("%-7s tiled=%d  lambda %.12g  x-half %.1f us  y-half %.1f us  kkt %.6e" % (name, info["tiled"], lam, 1e3 * t["xhalf_ms"] / 50, 1e3 * t["yhalf_ms"] / 50,
                                                                                  out[name]["res"]["kkt"]), file=sys.stderr)
    s.close(); model.free()
    for k in env:
        os.environ.pop(k)
a, b2 = out["tiled"], out["stream"]
assert a["info"]["tiled"] == 3 and b2["info"]["tiled"] == 0
for k in ("x", "y", "xb"):
    d = np.max(np.abs(a[k] - b2[k]) / (1e-12 + np.abs(b2[k]).max()))
    print("max difference of %s relative to its largest entry: %.2e" % (k, d), file=sys.stderr)
    assert d <= 1e-9, k
assert abs(a["lam"] - b2["lam"]) <= 1e-9 * abs(b2["lam"])
assert abs(a["res"]["kkt"] - b2["res"]["kkt"]) <= 1e-7 * (1 + abs(b2["res"]["kkt"]))
print("tiled path with long rows aside agrees with the stream kernel", file=sys.stderr)
