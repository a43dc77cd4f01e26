#!/usr/bin/env python3
"""Unstructured (uniformly random pattern) square LPs of growing size: stream kernel vs the tiled form without a dense-tile
requirement (every entry through the propagation-blocking remainder).  Finds the column count from which the latter wins
(Solver::pb_fallback_wanted).  usage: python tools/unstructured_ab.py [rows_in_millions ...]"""
import os

os.environ.setdefault("HPRLP_TEST_HOOKS", "1")  # the HPRLP_* switches used here are test hooks (csrc/env.h)
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys, time
import numpy as np
sys.path.insert(0, %r)
import bench
H = bench.H
n = m = int(float(sys.argv[1]) * 1e6)
per = int(sys.argv[2])
rng = np.random.default_rng(3)
ci = rng.integers(0, n, size=(m, per), dtype=np.int64)
ci.sort(axis=1)
keep = np.ones((m, per), bool); keep[:, 1:] = ci[:, 1:] != ci[:, :-1]
rp = np.concatenate([[0], np.cumsum(keep.sum(axis=1))]).astype(np.int32)
ci = ci[keep].astype(np.int32)
v = rng.normal(size=len(ci))
x0 = rng.uniform(0, 1, size=n)
from scipy import sparse
b = sparse.csr_matrix((v, ci, rp), shape=(m, n)) @ x0
model = H.Model.from_csr(m, n, rp, ci, v, b - 1.0, b + 1.0, np.zeros(n), np.full(n, 2.0), rng.normal(size=n))
os.dup2(2, 1)
s = H.Solver(model, H.Parameters(use_presolve=False))
info = s.info()
s.scale()
s.init(0.7, 1.3)
s.iterate(20, False)
tm = s.time_iterations(20, 60, 1)
print("RESULT tiled=%%d reordered=%%d  x %%.4f ms  y %%.4f ms" %% (info["tiled"], info["reordered"], tm["xhalf_ms"] / 60, tm["yhalf_ms"] / 60), file=sys.stderr)
''' % ROOT

sizes = [float(a) for a in sys.argv[1:]] or [0.5, 1.0, 2.0, 3.0, 4.2]
for sz in sizes:
    for name, env in (("stream", {"HPRLP_NO_PB_FALLBACK": "1"}), ("all-remainder tiled", {"HPRLP_PB_MIN_COLS": "1", "HPRLP_TILED_MIN_COLS": "1"})):
        e = dict(os.environ, **env)
        r = subprocess.run([sys.executable, "-c", CHILD, str(sz), os.environ.get("AB_PER_ROW", "10")], env=e, capture_output=True, text=True, timeout=600)
        line = [l for l in r.stderr.splitlines() if l.startswith("RESULT")]
        print("%4.2fM x %4.2fM, %s/row  %-20s %s" % (sz, sz, os.environ.get("AB_PER_ROW", "10"), name, line[0] if line else "FAILED " + r.stderr[-300:]), flush=True)
