import os, sys, time
ROOT = "/root/repo"
sys.path.insert(0, ROOT)
mode = sys.argv[1]
if mode in ("torch", "torch_sync"):
    import torch
    torch.cuda.set_device(0)
    if mode == "torch_sync":
        torch.cuda.synchronize()
import bench
H, G = bench.H, bench.G
os.dup2(2, 1)
lp = G.c2_25fv47_like()
model = H.Model.from_csr(lp["m"], lp["n"], lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])
if mode == "timeit":
    s = H.Solver(model, H.Parameters(stop_tol=1e-4, use_presolve=False)); s.scale(); lam, _ = s.power_iteration(); s.init(-1.0, lam * 1.01)
    s.time_iterations(200, 2000, 0); s.close()
for rep in range(3):
    s = H.Solver(model, H.Parameters(stop_tol=1e-4, use_presolve=False))
    s.scale(); t2 = time.time()
    lam, it = s.power_iteration(); t3 = time.time()
    s.init(-1.0, lam * 1.01); t4 = time.time()
    r = s.run(); t5 = time.time()
    print("MODE %s rep %d: power %.4f run %.4f (%d it)" % (mode, rep, t3 - t2, t5 - t4, r.iter), file=sys.stderr)
    s.close()
