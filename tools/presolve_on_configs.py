import os, sys, time
sys.path.insert(0, "/root/repo")
import bench
H, G = bench.H, bench.G
os.dup2(2, 1)
for name, lp in (("c2", G.c2_25fv47_like()), ("c3", G.c3_pds20_like())):
    model = H.Model.from_csr(lp["m"], lp["n"], lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])
    for pre in (False, True):
        t0 = time.time()
        r = model.solve(H.Parameters(stop_tol=1e-4, use_presolve=pre))
        k = H.original_kkt(model, r.x, r.y, r.z)
        print("RESULT %s presolve=%s: %s, %d iterations, wall %.3f s, rel obj err %.2e, original KKT %.2e" % (name, pre, r.status, r.iter, time.time() - t0,
              abs(r.primal_obj - lp["obj_star"]) / (1 + abs(lp["obj_star"])), max(k["primal_feas"], k["dual_feas"], k["gap"])), file=sys.stderr)
    model.free()
