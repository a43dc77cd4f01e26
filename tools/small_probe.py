"""Developer probe: iterations/s of the single-workgroup small-LP kernel (small.hip) against the regular
per-half-step kernels (graph replay) on Netlib-scale stand-ins."""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, rel)); mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod; spec.loader.exec_module(mod); return mod
H = _load("hprlp_amd", "hpr-lp-c_amd/hprlp.py"); G = _load("hprlp_lpgen", "hpr-lp-c_amd/lpgen.py")
os.dup2(2, 1)
cases = {"c2 821x1571 nnz10.5k": G.c2_25fv47_like(), "300x500 nnz2.5k": G.planted_lp(300, 500, 2500, 5, dense_col_frac=0.01),
         "500x1000 nnz7.5k": G.planted_lp(500, 1000, 7500, 5, dense_col_frac=0.01)}
for key, lp in cases.items():
    for no_small in ("1", "0"):
        os.environ["HPRLP_NO_SMALL"] = no_small
        model = H.Model.from_csr(lp["m"], lp["n"], lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])
        s = H.Solver(model, H.Parameters(use_presolve=False)); s.scale(); lam, _ = s.power_iteration(); s.init(-1.0, lam * 1.01)
        small = bool(s.info()["tiled"] & 4)
        g = s.time_iterations(300, 3000, 0)
        g2 = s.time_iterations(0, 149, 0)
        print(f"{key:22s} small={small}: {g['total_ms']/3000*1e3:6.2f} us/iter over 3000 its; one 149-iteration launch {g2['total_ms']*1e3:7.1f} us", file=sys.stderr)
        s.close(); model.free()
