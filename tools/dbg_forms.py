import importlib.util, os, sys, numpy as np
os.environ["HPRLP_TIMING"]="1"; os.environ["HPRLP_TEST_HOOKS"]="1"
sys.path.insert(0,'/root/repo')
spec=importlib.util.spec_from_file_location('fr','/root/repo/tools/form_regret.py'); fr=importlib.util.module_from_spec(spec); spec.loader.exec_module(fr)
import bench
H=bench.H
os.dup2(2,1)
name=sys.argv[1]
A=fr.CORPUS[name]().tocsr(); A.sort_indices(); A.data=np.random.default_rng(7).normal(size=A.nnz)
m,n=A.shape
lp=bench.planted_on(m,n,A.indptr.astype(np.int32),A.indices.astype(np.int32),A.data)
for env in ({}, {"HPRLP_NO_TILED":"1"}):
    os.environ.update(env)
    print("=====",name,env,file=sys.stderr)
    model=H.Model.from_csr(m,n,lp["rowptr"],lp["colind"],lp["values"],lp["AL"],lp["AU"],lp["l"],lp["u"],lp["c"])
    s=H.Solver(model,H.Parameters(use_presolve=False))
    print("DESCRIBE",s.describe(),file=sys.stderr)
    s.scale(); lam,_=s.power_iteration(max_iter=20); s.init(-1.0,lam*1.01)
    t=s.time_iterations(10,40,1); print("TIMES x %.4f y %.4f"%(t["xhalf_ms"]/40,t["yhalf_ms"]/40),file=sys.stderr)
    s.close(); model.free()
