"""CPU: the N>1 path (row partition + per-half-step all-gather) on gloo with world_size 2 and 3."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_iteration_equals_single_process(world):
    port = free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out.decode())
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed:\n{out[-3000:]}"
        assert f"rank {rank} ok" in out


def test_bench_shard_generation():
    """bench.py --gpus N: per-rank generation (no rank holds the whole LP) gives the shards of the LP generated whole."""
    world = 2
    port = free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker_shard.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for rank, p in enumerate(procs):
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, f"rank {rank} failed:\n{out.decode()[-3000:]}"
        assert f"rank {rank} ok" in out.decode()


def test_partition_edge_cases():
    import ctypes as C
    from conftest import hprlp
    L = hprlp.lib()
    off = C.c_int(); cnt = C.c_int()
    # more ranks than items: trailing ranks own nothing but keep their slot in the padded buffer
    got = []
    for r in range(8):
        chunk = L.hprlp_partition(5, 8, r, C.byref(off), C.byref(cnt))
        got.append((chunk, off.value, cnt.value))
    assert got == [(1, r, 1 if r < 5 else 0) for r in range(8)]
    assert L.hprlp_partition(10, 4, 3, C.byref(off), C.byref(cnt)) == 3 and (off.value, cnt.value) == (9, 1)
    assert L.hprlp_partition(10, 0, 0, C.byref(off), C.byref(cnt)) < 0
    assert L.hprlp_partition(10, 2, 2, C.byref(off), C.byref(cnt)) < 0


def test_threaded_host_transpose_equals_sequential():
    """Matrices above 4M nonzeros are transposed by 8 threads (host_model.cpp); the result must be the
    stable counting sort of reference src/utils.cu:203-232, i.e. identical to the oracle's."""
    import ctypes as C

    import numpy as np

    import bench_helpers as bh
    from conftest import hprlp
    from dist_worker import Shard, arr
    from oracle import oracle as O
    m = n = 260_000
    rp, ci, v = bh.gen_banded(m, n, 17, 3000, seed=9)
    assert len(v) > 4_000_000
    z = np.zeros
    model = hprlp.Model.from_csr(m, n, rp, ci, v, z(m), z(m), z(n), z(n), z(n))
    L = hprlp.lib()
    sh = Shard()
    L.hprlp_extract_shard.argtypes = [C.POINTER(hprlp.CLPInfo), C.c_int, C.c_int, C.POINTER(Shard)]
    assert L.hprlp_extract_shard(model._ptr, 0, 1, C.byref(sh)) == 0
    trp, tci, tv = O.transpose(m, n, rp, ci, v)
    assert np.array_equal(arr(sh.AT_rowptr, n + 1, np.int32), trp)
    nz = int(trp[-1])
    assert np.array_equal(arr(sh.AT_col, nz, np.int32), tci) and np.array_equal(arr(sh.AT_val, nz, np.float64), tv)
    L.hprlp_free_shard.argtypes = [C.POINTER(Shard)]
    L.hprlp_free_shard(C.byref(sh))
    model.free()


ROOT = os.path.dirname(HERE)


def _bench(argv, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, timeout=240)


def test_bench_gpus2_without_a_launcher_starts_two_ranks_and_fails_loudly_without_gpus():
    """`python bench.py --gpus 2` with WORLD_SIZE unset must start two ranks itself (child processes, the parent makes no HIP
    call), both must get as far as building their RCCL communicators, and -- no GPU here -- the run must end non-zero WITHOUT
    a result line: never a line that says n_gpus 1 for --gpus 2."""
    r = _bench(["--gpus", "2", "--workload", "c5_tiny", "--steps", "2", "--warmup", "1", "--no-cpu", "--no-side", "--no-solve"])
    err = r.stderr.decode()
    assert r.returncode != 0, err[-2000:]
    assert r.stdout.decode().strip() == "", r.stdout
    assert "started ranks as child processes" in err
    assert "rank 0 assembled its shard of c5_tiny" in err and "over 2 ranks" in err
    assert "rank 0: creating the RCCL unique ids for 2 ranks" in err
    # each rank either reached communicator creation or was told by rank 0 that the id could not be made; both say FAILED
    assert err.count("FAILED at communicator") >= 1
    assert "no result line" in err


def test_bench_refuses_a_world_that_does_not_match_gpus():
    """Under a launcher whose WORLD_SIZE differs from --gpus the bench refuses (non-zero, no line) instead of measuring
    another number of GPUs than it was asked for."""
    r = _bench(["--gpus", "8", "--no-cpu", "--no-side", "--no-solve"], {"WORLD_SIZE": "1", "RANK": "0"}, drop=())
    assert r.returncode != 0
    assert r.stdout.decode().strip() == ""
    assert "--gpus 8 but WORLD_SIZE=1" in r.stderr.decode()


SHM_WORKER = r"""
import ctypes as C, os, sys
sys.path.insert(0, {tests!r})
from conftest import hprlp
L = hprlp.lib()
rank, size, rounds, hang = (int(v) for v in sys.argv[1:5])
uid = bytes.fromhex(sys.argv[5])
rc = L.hprlp_shm_transport_selftest(uid, len(uid), rank, size, rounds, hang)
print("rank", rank, "rc", rc, hprlp.last_error() if rc else "ok", flush=True)
sys.exit(0 if rc == 0 else 7)
"""


def _shm_id():
    import ctypes as C
    sys.path.insert(0, HERE)
    from conftest import hprlp
    L = hprlp.lib()
    buf = (C.c_ubyte * 128)()
    os.environ["HPRLP_DIST_TRANSPORT"] = "shm"
    try:
        assert L.hprlp_dist_unique_id(buf, 128) == 0, hprlp.last_error()   # (no RCCL, no GPU needed for this kind of id)
    finally:
        os.environ.pop("HPRLP_DIST_TRANSPORT")
    assert bytes(buf[:8]) == b"HPRLPSHM"
    return bytes(buf)


def _shm_ranks(size, rounds, hang, timeout_s="120"):
    uid = _shm_id()
    code = SHM_WORKER.format(tests=HERE)
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), str(size), str(rounds), str(hang), uid.hex()],
                              env=dict(os.environ, HPRLP_DIST_TIMEOUT_S=timeout_s), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(size)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=120)[0].decode())
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    name = uid[8:108].split(b"\0")[0].decode()
    return procs, outs, name


@pytest.mark.parametrize("size", [2, 5])
def test_shared_memory_transport_protocol_between_processes(size):
    """csrc/dist.cpp ShmComm (round 5: the staged fallback's transport of last resort): `size` PROCESSES attach to one POSIX
    shared-memory segment and run 40 rounds of all-gather, scalar all-reduce (rank-order sum: same bits everywhere) and a ragged
    neighbour exchange on host buffers; every payload is checked by its receiver.  The segment's name is gone afterwards."""
    procs, outs, name = _shm_ranks(size, 40, -1)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r} rc 0 ok" in out, out[-2000:]
    assert not os.path.exists("/dev/shm" + name)


def test_shared_memory_transport_times_out_when_a_rank_goes_missing():
    """A rank that leaves half-way (a crashed or hung peer) must not leave the others waiting for ever: after
    HPRLP_DIST_TIMEOUT_S they end with an error that names the barrier, and the segment is marked broken for everybody."""
    t0 = __import__("time").time()
    procs, outs, name = _shm_ranks(3, 20, 1, timeout_s="2")
    assert procs[1].returncode == 0
    for r in (0, 2):
        assert procs[r].returncode == 7 and ("timed out" in outs[r] or "another rank failed" in outs[r]), outs[r][-2000:]
    assert any("timed out" in o for o in outs)
    assert __import__("time").time() - t0 < 60
    assert not os.path.exists("/dev/shm" + name)


def test_bench_supervisor_ends_a_hung_tier_and_falls_back_in_fresh_processes():
    """Round 5 (bench.py: supervise / run_tier): a rank that hangs (test hook: rank 1 of tier 0 sleeps for ever in its first
    phase, so rank 0 waits for it in a collective -- the shape of a deadlocked exchange) must not cost the run its time limit:
    after HPRLP_BENCH_STALL_S seconds of silence the parent terminates exactly the PIDs it started and starts the next
    transport tier in FRESH processes (new PIDs, new rendezvous port, the tier's environment).  No GPU here, so every later
    tier ends at communicator creation and the run ends non-zero without a line, saying how each tier ended."""
    import re
    r = _bench(["--gpus", "2", "--workload", "c5_tiny", "--steps", "2", "--warmup", "1", "--no-cpu", "--no-side", "--no-solve"],
               {"HPRLP_BENCH_STALL_S": "5", "HPRLP_BENCH_TEST_HANG": "0:1:shard"})
    err = r.stderr.decode()
    assert r.returncode != 0 and r.stdout.decode().strip() == "", err[-3000:]
    assert "rank 1: TEST HOOK: hanging at phase 'shard assembly'" in err
    assert "tier 0 ended: stalled: no line from any rank for 5s (exit codes [-15, -15])" in err, err[-3000:]
    starts = re.findall(r"tier (\d): .*started ranks as child processes \[(\d+), (\d+)\] \(127.0.0.1:(\d+)\)", err)
    assert [int(t[0]) for t in starts] == [0, 1, 2, 3], starts
    pids = [p for t in starts for p in t[1:3]]
    assert len(set(pids)) == 8 and len({t[3] for t in starts}) == 4     # fresh processes, fresh ports
    assert "1 id(s)" in err and "2 id(s)" not in err.split("tier 1:")[1]   # the fallback tiers build ONE communicator
    assert "naming the shared-memory segment for 2 ranks" in err.split("tier 3:")[1]
    for t in (1, 2, 3):
        assert re.search(rf"tier {t} ended: rank \d exited with code 3", err), err[-3000:]
    tail = err.split("no result line: every transport tier failed")[1]
    assert "tier 0" in tail and "stalled" in tail and tail.count("exited with code 3") == 3


def test_bench_under_a_launcher_rank0_supervises_and_the_other_copies_wait():
    """Under `python -m torch.distributed.run --nproc-per-node N` (how the driver starts N > 1) the N copies of bench.py make
    no HIP call: copy 0 runs the tiers with fresh rank processes on their own port, the others wait for its verdict over gloo
    and leave with the same exit code."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(HPRLP_BENCH_STALL_S="20", HPRLP_BENCH_FIRST_TIER="3")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "c5_tiny",
                        "--steps", "2", "--warmup", "1", "--no-cpu", "--no-side", "--no-solve"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
    err = r.stderr.decode()
    assert r.returncode != 0 and r.stdout.decode().strip() == "", err[-3000:]
    assert err.count("started ranks as child processes") == 1            # one supervisor, not two
    assert "tier 3: host-staged shared memory" in err and "rank 1: phase shard assembly" in err
    assert "no result line: every transport tier failed" in err


def test_predicted_scaling_model_from_shard_points():
    """bench.py: predicted_scaling() -- the model the first N > 1 record is held against: 2 x the shard's y-half x 1.05 + the exposed
    part of the exchange, from the ladder points of the same run; a point that failed is reported as such, not guessed."""
    sys.path.insert(0, ROOT)
    import bench
    ladder = {"c5_shard_of_2": {"yhalf_ms": 0.36, "yhalf_frac_of_8000": 0.52, "kernels": "A: tiled, fused (k_tiled_fused, grid 512); A^T: x"},
              "c5_shard_of_4": {"yhalf_ms": 0.18, "yhalf_frac_of_8000": 0.55, "kernels": "A: tiled, piece form; A^T: x"},
              "c5_shard_of_8": {"error": "out of memory"}}
    p = bench.predicted_scaling(1.30, ladder)
    assert abs(p["P2"]["iteration_ms_model"] - (2 * 0.36 * 1.05 + 0.01)) < 1e-12 and abs(p["P2"]["speedup_model"] - 1.30 / (0.756 + 0.01)) < 1e-9
    assert abs(p["P4"]["speedup_kernels_only"] - 1.30 / 0.36) < 1e-9 and p["P4"]["shard_kernel"].startswith("A: tiled, piece form")
    assert p["P8"] == {"error": "out of memory"} and p["P1_iteration_ms"] == 1.30
    # the real shard of a middle rank: rows [rank m / P, ...) of the SAME matrix, global columns
    lp = None
    bench.WORKLOADS["_t"] = (40_000, 40_000, 6, 300)
    try:
        lp = bench.shard_rows_lp("_t", 4)
        full = bench.gen_banded(40_000, 40_000, 6, 300)
    finally:
        bench.WORKLOADS.pop("_t")
    assert (lp["m"], lp["n"]) == (10_000, 40_000)
    k0 = full[0][20_000]
    assert np.array_equal(lp["colind"], full[1][k0:k0 + len(lp["colind"])]) and np.array_equal(lp["values"], full[2][k0:k0 + len(lp["values"])])
