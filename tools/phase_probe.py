"""Developer probe: wall time of the set-up phases (model build, upload + transpose, scaling, power iteration)
on the banded benchmark matrix.  usage: python tools/phase_probe.py [c5|c5_small|c5_tiny]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B
H = B.H
os.dup2(2, 1)
name = sys.argv[1] if len(sys.argv) > 1 else "c5"
m, n, per_row, band = B.WORKLOADS[name]
t0 = time.time(); lp = B.banded_lp(m, n, per_row, band); t1 = time.time()
model = H.Model.from_csr(m, n, lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"]); t2 = time.time()
s = H.Solver(model, H.Parameters(use_presolve=False)); t3 = time.time()
s.scale(); t4 = time.time()
lam, it = s.power_iteration(); t5 = time.time()
print(f"{name}: generate {t1-t0:.2f}s  model copy {t2-t1:.2f}s  solver create (transpose+upload+tiling) {t3-t2:.2f}s  "
      f"scale {t4-t3:.3f}s  power iteration {t5-t4:.3f}s ({it} its, lambda {lam:.6g})", file=sys.stderr)
s.close(); model.free()
