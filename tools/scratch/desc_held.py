import os, sys
os.environ.setdefault("HPRLP_TEST_HOOKS", "1")
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import numpy as np
import importlib.util
spec = importlib.util.spec_from_file_location("fr", "tools/form_regret.py"); fr = importlib.util.module_from_spec(spec); spec.loader.exec_module(fr)
bench, H = fr.bench, fr.H
real = os.dup(1); os.dup2(2, 1)
for name in sys.argv[1].split(","):
    A = fr.HELD_OUT[name]().tocsr(); A.sort_indices()
    A.data = np.random.default_rng(7).normal(size=A.nnz)
    m, n = A.shape
    lp = bench.planted_on(m, n, A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64))
    model = H.Model.from_csr(lp["m"], lp["n"], lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])
    s = H.Solver(model, H.Parameters(stop_tol=1e-4, use_presolve=False))
    os.write(real, (name + " :: " + s.describe() + "\n").encode())
    s.close(); model.free()
