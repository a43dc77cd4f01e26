"""Worker of tests/test_gpu_dist.py::test_process_ranks_over_shared_memory_*: ONE rank of the row-partitioned solver in its own
process, all ranks on device 0, exchanging through the shared-memory transport (csrc/dist.cpp ShmComm).  Writes what
run_ranks() of the thread-rank tests collects to <out>.<rank>.pkl."""
import os
import pickle
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from conftest import hprlp, lpgen  # noqa: E402
import bench_helpers as bh  # noqa: E402


def make_lp(kind):
    if kind == "unstructured":
        return lpgen.planted_lp(401, 653, 4000, 92)
    if kind == "banded":
        return bh.banded_lp(6001, 6001, 8, 150)
    if kind == "banded_large":   # tiled kernels on the shards, neighbour exchange of a halo
        return bh.banded_lp(1_200_000, 1_200_000, 10, 12000)
    raise SystemExit("unknown LP kind " + kind)


def main():
    rank, world, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    uid = np.frombuffer(bytes.fromhex(sys.argv[4]), np.uint8).copy()
    kind, out = sys.argv[5], sys.argv[6]
    lp = make_lp(kind)
    model = hprlp.Model.from_csr(lp["m"], lp["n"], lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])
    prm = hprlp.Parameters(stop_tol=1e-6 if kind != "banded_large" else 1e-4, use_presolve=False)
    s = hprlp.Solver.create_dist(model, prm, rank, world, uid)
    ci = s.dist_comm_info()
    s.scale()
    lam, it = s.power_iteration()
    s.init(-1.0, lam * 1.01)
    s.iterate(steps, True)
    res = s.residuals(steps + 1, True)
    state = {k: s.get(k) for k in ("x", "y", "x_bar", "y_bar", "z_bar")}
    r = s.run()
    rec = dict(lam=lam, it=it, res=res, state=state, info=s.dist_info(), comm=ci, describe=s.describe(), off=(s.row_off, s.m_loc, s.col_off, s.n_loc),
               run=dict(status=r.status, iter=r.iter, primal_obj=r.primal_obj, x=np.array(r.x), y=np.array(r.y)))
    s.close()
    with open(f"{out}.{rank}.pkl", "wb") as f:
        pickle.dump(rec, f)
    print(f"rank {rank} ok", flush=True)


if __name__ == "__main__":
    main()
