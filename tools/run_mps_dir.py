#!/usr/bin/env python3
"""Directory of .mps / .mps.gz files -> the table of BASELINE.md section 4, one row per instance.

    python tools/run_mps_dir.py DIR [--tol 1e-4] [--time-limit 3600] [--presolve true|false] [--out table.md] [--json rows.json]

For every file (sorted by size, smallest first): create_model_from_mps + solve() through the C ABI on GPU 0 -- the
reference's driver loop (src/solve_mps_file.cpp:120-131) over a directory -- then the reference's own instruments
(HPRLP_results.iter / iter4 / time4 / time, include/structs.h:50-57), the KKT errors of the returned triple recomputed on the
model AS READ (hprlp_original_kkt), iterations per second of the loop, and which kernel form ran.  Real Netlib /
Mittelmann instances are not on the build or GPU boxes (no network): drop them into a directory and this is the one command.
"""
import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location("hprlp_amd", os.path.join(ROOT, "hpr-lp-c_amd", "hprlp.py"))
H = importlib.util.module_from_spec(_spec)
sys.modules["hprlp_amd"] = H
_spec.loader.exec_module(H)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir", nargs="?", default=None)
    ap.add_argument("--generated", default=None, choices=["families_small", "families_large"],
                    help="instead of a directory: the four Mittelmann-family generators of hpr-lp-c_amd/lpgen.py (pds-, nug-, cont-like, "
                         "staircase) at test size (<= 1e5 nonzeros) or table size (1e6 - 1e7), handed over through create_model_from_arrays")
    ap.add_argument("--tol", type=float, default=1e-4)
    ap.add_argument("--time-limit", type=float, default=3600.0)
    ap.add_argument("--max-iter", type=int, default=2**31 - 1)
    ap.add_argument("--presolve", default="true")
    ap.add_argument("--out", default=None, help="markdown table (default: stdout)")
    ap.add_argument("--json", default=None, help="rows as JSON")
    args = ap.parse_args()
    if args.generated:
        _s2 = importlib.util.spec_from_file_location("hprlp_lpgen", os.path.join(ROOT, "hpr-lp-c_amd", "lpgen.py"))
        G = importlib.util.module_from_spec(_s2)
        _s2.loader.exec_module(G)
        fam = G.FAMILIES_SMALL if args.generated == "families_small" else G.FAMILIES_LARGE

        def from_generator(make):
            lp = make()
            return H.Model.from_csr(lp["m"], lp["n"], lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])
        files = [(f"{k} ({args.generated})", (lambda mk=mk: from_generator(mk))) for k, mk in fam.items()]
    else:
        if not args.dir:
            raise SystemExit("give a directory of .mps files or --generated")
        paths = sorted(glob.glob(os.path.join(args.dir, "*.mps")) + glob.glob(os.path.join(args.dir, "*.mps.gz")) +
                       glob.glob(os.path.join(args.dir, "*.MPS")), key=os.path.getsize)
        if not paths:
            raise SystemExit(f"no .mps / .mps.gz files in {args.dir}")
        files = [(os.path.basename(f), (lambda f=f: H.Model.from_mps(f))) for f in paths]
    presolve = args.presolve.lower() in ("true", "1", "yes")
    real_stdout = os.dup(1)
    os.dup2(2, 1)  # the library prints its banner and iteration log to the C-level stdout
    rows = []
    for name, load in files:
        row = {"instance": name}
        try:
            t0 = time.time()
            model = load()
            row.update(m=model.m, n=model.n, nnz=model.nnz, read_s=time.time() - t0)
            prm = H.Parameters(stop_tol=args.tol, time_limit=args.time_limit, max_iter=args.max_iter, use_presolve=presolve)
            t1 = time.time()
            r = model.solve(prm)
            row.update(status=r.status, iterations=r.iter, iter4=r.iter4, time4_s=r.time4, solver_time_s=r.time,
                       wall_s=time.time() - t1, primal_obj=r.primal_obj, kkt_reported=r.residuals,
                       iterations_per_s=r.iter / max(r.time, 1e-9))
            if r.x is not None and len(r.x) == model.n:
                k = H.original_kkt(model, r.x, r.y, r.z)
                row.update(kkt_primal=k["primal_feas"], kkt_dual=k["dual_feas"], kkt_gap=k["gap"])
            # which kernels the library picks for this matrix (no presolve: the matrix as read)
            try:
                s = H.Solver(model, H.Parameters(use_presolve=False))
                row["kernels"] = s.describe()
                s.close()
            except Exception as e:  # noqa: BLE001
                row["kernels"] = f"(n/a: {e})"
            model.free()
        except Exception as e:  # noqa: BLE001
            row["error"] = str(e)
        rows.append(row)
        print(f"[run_mps_dir] {name}: {row.get('status', row.get('error'))}", file=sys.stderr, flush=True)
    lines = ["| instance | m | n | nnz | status | iterations | iter4 | time4 [s] | solver time [s] | it/s | primal objective | KKT primal / dual / gap (model as read) | kernels |",
             "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for r in rows:
        if "error" in r:
            lines.append(f"| {r['instance']} | | | | ERROR: {r['error']} | | | | | | | | |")
            continue
        kkt = " / ".join(f"{r.get(k, float('nan')):.1e}" for k in ("kkt_primal", "kkt_dual", "kkt_gap"))
        lines.append(f"| {r['instance']} | {r['m']} | {r['n']} | {r['nnz']} | {r['status']} | {r['iterations']} | {r['iter4']} | {r['time4_s']:.3f} | "
                     f"{r['solver_time_s']:.3f} | {r['iterations_per_s']:.0f} | {r['primal_obj']:.8g} | {kkt} | {r.get('kernels', '')} |")
    text = "\n".join(lines) + "\n"
    if args.out:
        open(args.out, "w").write(text)
    else:
        os.write(real_stdout, text.encode())
    if args.json:
        json.dump(rows, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
