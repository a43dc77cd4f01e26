"""Where does a GPU solve leave the oracle's trajectory?  Both logs (every check step: iteration, restart flag, sigma, KKT error)
side by side: the relative difference of sigma / kkt along the common prefix, and the first row where a decision differs.  A fork at
a thresholded decision shows differences that start at rounding level and grow; a defect would show a jump.
usage: python tools/fork_trace.py m n nnz seed [tol]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import hprlp, lpgen  # noqa: E402
from oracle import oracle as O  # noqa: E402

m, n, nnz, seed = (int(v) for v in sys.argv[1:5])
tol = float(sys.argv[5]) if len(sys.argv) > 5 else 1e-6
os.dup2(2, 1)
lp = lpgen.planted_lp(m, n, nnz, seed, dense_col_frac=0.02 if seed % 2 else 0.0, free_frac=0.1 if seed % 3 == 0 else 0.0)
model = hprlp.Model.from_csr(lp["m"], lp["n"], lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])
s = hprlp.Solver(model, hprlp.Parameters(stop_tol=tol, use_presolve=False, max_iter=200000))
s.scale()
lam, it = s.power_iteration()
s.init(-1.0, lam * 1.01)
res = s.run()
ref = O.solve(lp["m"], lp["n"], lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"],
              params=O.Params.default(stop_tol=tol, max_iter=200000))
ref_same_lambda = O.solve(lp["m"], lp["n"], lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"],
                          params=O.Params.default(stop_tol=tol, max_iter=200000), lambda_override=lam * 1.01)
print("TRACE gpu %s %d iterations | oracle %s %d | oracle with the GPU's lambda_max %s %d" % (res.status, res.iter, ref["status"], ref["iter"], ref_same_lambda["status"], ref_same_lambda["iter"]), file=sys.stderr)
for name, rf in (("oracle", ref), ("oracle, GPU's lambda", ref_same_lambda)):
    k = min(len(res.trace), len(rf["trace"]))
    worst, first = 0.0, None
    for i, (a, b) in enumerate(zip(res.trace[:k], rf["trace"][:k])):
        if a["iter"] != b["iter"] or a["restart_flag"] != b["restart_flag"]:
            first = i
            break
        d = max(abs(a["sigma"] - b["sigma"]) / max(abs(b["sigma"]), 1e-300), abs(a["kkt"] - b["kkt"]) / max(abs(b["kkt"]), 1e-300))
        if i in (0, 1, 2, 5, 10, 20, 50, 100, 200, 500, 1000, 2000) or d > 10 * max(worst, 1e-16):
            print("TRACE %-22s row %5d iter %7d restart %d  rel diff sigma/kkt %.2e  (sigma %.6e kkt %.6e)" % (name, i, a["iter"], a["restart_flag"], d, a["sigma"], a["kkt"]), file=sys.stderr)
        worst = max(worst, d)
    print("TRACE %-22s common prefix %s of %d rows, largest relative difference on it %.2e" % (name, first if first is not None else k, k, worst), file=sys.stderr)
    if first is not None:
        a, b = res.trace[first], rf["trace"][first]
        print("TRACE %-22s first different row: gpu iter %d restart %d sigma %.6e | oracle iter %d restart %d sigma %.6e" % (name, a["iter"], a["restart_flag"], a["sigma"], b["iter"], b["restart_flag"], b["sigma"]), file=sys.stderr)
s.close()
model.free()
