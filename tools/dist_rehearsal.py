#!/usr/bin/env python3
"""Full-size rehearsal of the sharded solver on ONE GPU: `world` ranks as threads of this process
(hprlp_solver_create_local: device copies + host barriers in place of RCCL).  Checks, at the size the driver's
N = 2/4/8 bench runs use, that shard extraction, the halo plan, the exchange self-test and the iterations work, and
that the sharded run reaches the tolerance with the planted objective.  Timings are NOT a scaling measurement (all
ranks share one GPU); the exchange volumes are the real ones.
usage: python tools/dist_rehearsal.py [--workload c5] [--world 8] [--steps 30]"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

H = bench.H


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c5", choices=sorted(bench.WORKLOADS))
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--solve", action="store_true", help="also run to 1e-4 and compare with the planted objective")
    args = ap.parse_args()
    m, n, per_row, band = bench.WORKLOADS[args.workload]
    t0 = time.time()
    lp = bench.banded_lp(m, n, per_row, band)
    model = H.Model.from_csr(m, n, lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])
    nnz = len(lp["values"])
    for k in ("rowptr", "colind", "values"):
        lp.pop(k)
    print(f"generated {m}x{n}, nnz={nnz} in {time.time() - t0:.1f}s", flush=True)
    prm = H.Parameters(stop_tol=1e-4, use_presolve=False)
    world = args.world
    group = H.Solver.local_group(world)
    out, err = [None] * world, [None] * world

    def work(rank):
        try:
            t = time.time()
            s = H.Solver.create_local(model, prm, rank, world, group)
            t_create = time.time() - t
            t = time.time()
            s.scale()
            lam, it = s.power_iteration()
            s.init(-1.0, lam * 1.01)
            t_prep = time.time() - t
            s.iterate(5)
            t = time.time()
            s.iterate(args.steps)
            s.residuals(args.steps + 6, True)
            t_it = time.time() - t
            o = dict(rank=rank, create_s=round(t_create, 2), scale_power_s=round(t_prep, 2), power_its=it,
                     ms_per_iteration_all_ranks_on_one_gpu=round(1e3 * t_it / args.steps, 3), info=s.dist_info(), tiled=s.info()["tiled"])
            if args.solve:
                s.reset()                      # as bench.py does at N > 1: the solver of the timed iterations, back to zero iterates
                s.init(-1.0, lam * 1.01)
                t = time.time()
                r = s.run()
                o.update(status=r.status, iters=r.iter, obj=r.primal_obj, kkt=r.residuals, loop_s=round(time.time() - t, 3))
            out[rank] = o
            s.close()
        except Exception as e:  # noqa: BLE001
            err[rank] = repr(e)

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    H.Solver.free_local_group(group)
    model.free()
    if any(err):
        print("FAILED", err)
        sys.exit(1)
    for o in out:
        print(json.dumps(o))
    if args.solve:
        rel = abs(out[0]["obj"] - lp["obj_star"]) / (1 + abs(lp["obj_star"]))
        print(f"status {out[0]['status']} after {out[0]['iters']} iterations, objective off the planted one by {rel:.2e}")
        assert out[0]["status"] == "OPTIMAL" and rel < 1e-3


if __name__ == "__main__":
    main()
