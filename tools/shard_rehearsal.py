#!/usr/bin/env python3
"""Host side of `bench.py --gpus N` at full size, on CPUs only: the N ranks assemble their shards of a workload (default
config 5: 10M x 10M, 2e8 nonzeros) exactly as bench.py does -- bench.banded_lp_shard over a gloo group -- and stop BEFORE
communicator creation.  Prints wall time and peak host RSS per rank; no GPU, no RCCL.

    python tools/shard_rehearsal.py --gpus 8 [--workload c5]
"""
import argparse
import json
import os
import resource
import subprocess
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    sys.path.insert(0, ROOT)
    rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
    t_import = time.time()
    import torch.distributed as dist
    import bench as B
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t0 = time.time()
    m, n, per_row, band = B.WORKLOADS[os.environ["REHEARSAL_WORKLOAD"]]
    shard, obj_star, nnz_loc = B.banded_lp_shard(m, n, per_row, band, rank, world, dist)
    wall = time.time() - t0
    k = shard.keep
    print(json.dumps({"rank": rank, "world": world, "assembly_wall_s": round(wall, 2), "import_and_rendezvous_s": round(t0 - t_import, 2),
                      "peak_rss_gb": round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6, 2), "nnz_A_rows": int(nnz_loc),
                      "nnz_AT_rows": int(k["AT_rp"][-1]), "obj_star": obj_star, "cpus": B.host_cpu_share()}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=8)
    ap.add_argument("--workload", default="c5")
    a = ap.parse_args()
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    t0 = time.time()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   REHEARSAL_WORKLOAD=a.workload, REHEARSAL_CHILD="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=env))
    rc = [p.wait() for p in procs]
    print(f"# {a.gpus} ranks, workload {a.workload}: all done in {time.time() - t0:.1f}s wall (exit codes {rc})", flush=True)
    return max(rc)


if __name__ == "__main__":
    if os.environ.get("REHEARSAL_CHILD"):
        child()
    else:
        sys.exit(main())
