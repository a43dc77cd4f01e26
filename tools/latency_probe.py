"""Developer probe: per-kernel time of the x-half / y-half on the config-2/3 stand-ins for different
row-block shapes (HPRLP_STREAM_ROWS / HPRLP_STREAM_NNZ) and launch modes."""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, rel)); mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod; spec.loader.exec_module(mod); return mod
H = _load("hprlp_amd", "hpr-lp-c_amd/hprlp.py"); G = _load("hprlp_lpgen", "hpr-lp-c_amd/lpgen.py")
os.dup2(2, 1)
lps = {"c2": G.c2_25fv47_like(), "c3": G.c3_pds20_like()}
for rows, nnz, nt in [(64, 512, 1), (64, 256, 1), (64, 128, 1), (64, 64, 1), (32, 64, 1), (64, 32, 1)]:
    os.environ["HPRLP_STREAM_ROWS"] = str(rows); os.environ["HPRLP_STREAM_NNZ"] = str(nnz); os.environ["HPRLP_NT"] = str(nt)
    for key, lp in lps.items():
        model = H.Model.from_csr(lp["m"], lp["n"], lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])
        s = H.Solver(model, H.Parameters(use_presolve=False)); s.scale(); lam, _ = s.power_iteration(); s.init(-1.0, lam * 1.01)
        info = s.info()
        g = s.time_iterations(200, 2000, 0); e = s.time_iterations(50, 500, 1)
        print(f"rows<={rows:3d} nnz<={nnz:3d} nt={nt} {key}: blocks A/AT {info['blocks_A']}/{info['blocks_AT']}  graph {g['total_ms']/2000*1e3:6.2f} us/iter   "
              f"eager x {e['xhalf_ms']/500*1e3:6.2f} y {e['yhalf_ms']/500*1e3:6.2f} us", file=sys.stderr)
        s.close(); model.free()
