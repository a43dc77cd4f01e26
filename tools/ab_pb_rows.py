#!/usr/bin/env python3
"""A/B of super-block heights for the all-remainder form (k_pb_fused) against the chosen form and the stream kernel on patterns
of tools/form_regret.py (tuning, held-out or the few-row extras below): half-step windows, KKT error after the run (the forms must
agree), describe line.

    python tools/ab_pb_rows.py PATTERN[,PATTERN..] HEIGHT[,HEIGHT..] [FORM[,FORM..]]      (GPU box; forms: chosen, stream, pb_H, pieces0_H)
"""
import os, sys
os.environ.setdefault("HPRLP_TEST_HOOKS", "1")
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import numpy as np
import importlib.util
spec = importlib.util.spec_from_file_location("fr", "tools/form_regret.py"); fr = importlib.util.module_from_spec(spec); spec.loader.exec_module(fr)
bench, H = fr.bench, fr.H
real = os.dup(1); os.dup2(2, 1)
out = lambda s: os.write(real, (s + "\n").encode())
PB = {"HPRLP_TILED_MIN_ROWS": "1", "HPRLP_TILED_MIN_DENSE": "1.01", "HPRLP_PB_MIN_COLS": "1", "HPRLP_PB_MIN_NNZ": "1"}
TL = {"HPRLP_TILED_MIN_ROWS": "1", "HPRLP_TILED_MIN_DENSE": "0.0", "HPRLP_TILED_ANYWAY": "1", "HPRLP_PIECES_ANYWAY": "1"}
envs = {"chosen": {}, "stream": {"HPRLP_NO_TILED": "1"},
        "stream_reordered": {"HPRLP_TILED_MIN_DENSE": "1.01", "HPRLP_NO_PB_FALLBACK": "1"}}   # locality ordering, then both matrices on the stream kernel
for r in sys.argv[2].split(","):
    envs["pb_%s" % r] = dict(PB, HPRLP_TILE_ROWS=r)
    envs["pieces0_%s" % r] = dict(TL, HPRLP_TILE_ROWS=r)
if len(sys.argv) > 3:
    envs = {k: v for k, v in envs.items() if k in sys.argv[3].split(",")}
EXTRA = {"cd3_25k_x3M": lambda: fr.fixed_column_degree(25_000, 3_000_000, 3, seed=81), "cd3_16k_x3M": lambda: fr.fixed_column_degree(16_000, 3_000_000, 2, seed=98),
         "band_20pct_6": lambda: fr.band(2_000_000, 2_000_000, 6, 0.2, seed=66), "band_10pct_8": lambda: fr.band(3_000_000, 3_000_000, 8, 0.1, seed=67),
         "band_10pct_12": lambda: fr.band(2_000_000, 2_000_000, 12, 0.1, seed=68),
         "cd3_50k": lambda: fr.fixed_column_degree(50_000, 5_000_000, 3, seed=61), "cd3_33k": lambda: fr.fixed_column_degree(33_000, 4_000_000, 3, seed=62),
         "cd4_250k": lambda: fr.fixed_column_degree(250_000, 3_000_000, 4, seed=63), "cd3_50k_local": lambda: fr.band(50_000, 5_000_000, 300, 0.002, seed=64),
         "wide_band_150k": lambda: fr.band(150_000, 3_000_000, 60, 0.05, seed=65)}
for name in sys.argv[1].split(","):
    A = (EXTRA.get(name) or fr.HELD_OUT.get(name) or fr.HELD_OUT_2.get(name) or fr.HELD_OUT_3.get(name) or fr.CORPUS[name])().tocsr(); A.sort_indices()
    A.data = np.random.default_rng(7).normal(size=A.nnz)
    m, n = A.shape
    lp = bench.planted_on(m, n, A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64))
    for k, env in envs.items():
        old = {q: os.environ.get(q) for q in env}; os.environ.update(env)
        try:
            model = H.Model.from_csr(lp["m"], lp["n"], lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])
            s = H.Solver(model, H.Parameters(stop_tol=1e-4, use_presolve=False))
            s.scale(); lam, _ = s.power_iteration(max_iter=20); s.init(-1.0, lam * 1.01)
            t = s.time_iterations(10, 40, 1)
            d = s.describe().split("; switches")[0]
            s.iterate(0, True); kkt = s.residuals(51)["kkt"]
            out("%-22s %-14s x %.4f y %.4f kkt %.12e | %s" % (name, k, t["xhalf_ms"] / 40, t["yhalf_ms"] / 40, kkt, d[:330]))
            s.close(); model.free()
        except Exception as e:
            out("%-22s %-14s ERROR %s" % (name, k, str(e)[:200]))
        finally:
            for q, v in old.items():
                os.environ.pop(q, None) if v is None else os.environ.__setitem__(q, v)
