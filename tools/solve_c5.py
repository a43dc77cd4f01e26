"""Full solve of the BASELINE config-5 LP (banded 10M x 10M, 2e8 nnz, planted optimum) to stop_tol on one GPU:
time-to-tolerance and objective error against the planted optimum.  usage: python tools/solve_c5.py [workload] [tol]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B
H = B.H
os.dup2(2, 1)
name = sys.argv[1] if len(sys.argv) > 1 else "c5"
tol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-4
m, n, per_row, band = B.WORKLOADS[name]
lp = B.banded_lp(m, n, per_row, band)
model = H.Model.from_csr(m, n, lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])
for rep in range(int(os.environ.get("HPRLP_SOLVE_REPS", "1"))):   # the last of several solves is a warm process's
    t0 = time.time()
    r = model.solve(H.Parameters(stop_tol=tol, use_presolve=False, time_limit=900.0))
    wall = time.time() - t0
    print("[solve_c5] rep %d phases %s" % (rep, H.last_solve_phases()), file=sys.stderr)
rel = abs(r.primal_obj - lp["obj_star"]) / (1 + abs(lp["obj_star"]))
print(f"[solve_c5] {name} tol={tol:g}: status {r.status}, {r.iter} iterations, solver time {r.time:.2f}s (wall incl. set-up {wall:.2f}s), "
      f"primal obj {r.primal_obj:.9e}, planted {lp['obj_star']:.9e}, rel err {rel:.2e}, residual {r.residuals:.2e}", file=sys.stderr)
