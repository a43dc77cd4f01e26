import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["HPRLP_TIMING"] = "1"
import bench as B
H, G = B.H, B.G
name = sys.argv[1]
lp = G.FAMILIES_LARGE[name]()
os.dup2(2, 1)
model = H.Model.from_csr(lp["m"], lp["n"], lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])
s = H.Solver(model, H.Parameters(use_presolve=False))
print(s.describe(), file=sys.stderr)
