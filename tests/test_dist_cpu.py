"""CPU: the N>1 path (row partition + per-half-step all-gather) on gloo with world_size 2 and 3."""
import os
import socket
import subprocess
import sys

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
