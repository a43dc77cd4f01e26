"""GPU: the RCCL code path of the row-partitioned solver with a ONE-rank communicator (the box has one
GPU).  Exercises librccl loading, communicator creation, the in-place all-gathers after every
half-step and the scalar all-reduces; with one rank they must not change any result."""
import os

import numpy as np
import pytest

from conftest import hprlp, lpgen

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ids", [1, 2])
def test_one_rank_rccl_path_equals_plain_solver(gpu, ids):
    """ids = 2: the launcher hands over two unique ids and the exchange stream gets its own communicator (no communicator
    is driven from two streams); what RCCL reports about both is what bench.py prints as rccl_ranks / devices."""
    lp = lpgen.planted_lp(400, 650, 4000, 91)
    model = hprlp.Model.from_csr(lp["m"], lp["n"], lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"],
                                 lp["l"], lp["u"], lp["c"])
    prm = hprlp.Parameters(stop_tol=1e-6, use_presolve=False)
    plain = hprlp.Solver(model, prm)
    uid = hprlp.Solver.dist_unique_id(ids)
    assert len(uid) == 128 * ids and (ids == 1 or not np.array_equal(uid[:128], uid[128:]))
    dist = hprlp.Solver.create_dist(model, prm, 0, 1, uid)
    ci = dist.dist_comm_info()
    assert (ci["comm_ranks"], ci["comm_rank"], ci["comm_device"], ci["hip_device"]) == (1, 0, 0, 0), ci
    assert (ci["xcomm_ranks"], ci["xcomm_rank"], ci["xcomm_device"]) == ((1, 0, 0) if ids == 2 else (0, -1, -1)), ci
    dist.dist_loopback(1 << 18)  # ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd: rank 0 to itself, 2 MiB, verified
    out = []
    # (bit-for-bit: both take lambda_max from the regular power iteration; the plain solver's single-launch one for small LPs
    # adds its dot products in another order -- tests/test_gpu_small.py)
    os.environ["HPRLP_NO_SMALL_POWER"] = "1"
    try:
        for s in (plain, dist):
            s.scale()
            lam, it = s.power_iteration()
            s.init(-1.0, lam * 1.01)
            s.iterate(37, True)
            res = s.residuals(38, True)
            state = {k: s.get(k) for k in ("x", "y", "x_bar", "y_bar", "z_bar")}
            r = s.run()
            out.append((lam, it, res, state, r))
            s.close()
    finally:
        os.environ.pop("HPRLP_NO_SMALL_POWER", None)
    (l0, i0, r0, s0, f0), (l1, i1, r1, s1, f1) = out
    assert (l0, i0) == (l1, i1)
    for k in s0:
        assert np.array_equal(s0[k], s1[k]), k
    for k in r0:
        assert r0[k] == r1[k], k
    assert (f0.status, f0.iter) == (f1.status, f1.iter) and f0.primal_obj == f1.primal_obj
    assert f0.status == "OPTIMAL" and abs(f0.primal_obj - lp["obj_star"]) <= 1e-5 * (1 + abs(lp["obj_star"]))
    model.free()


def run_ranks(model, prm, world, steps):
    """`world` ranks of the sharded solver as threads of this process on the one GPU (hprlp_solver_create_local)."""
    import threading
    group = hprlp.Solver.local_group(world)
    out, err = [None] * world, [None] * world

    def work(rank):
        try:
            s = hprlp.Solver.create_local(model, prm, rank, world, group)
            s.scale()
            lam, it = s.power_iteration()
            s.init(-1.0, lam * 1.01)
            s.iterate(steps, True)
            res = s.residuals(steps + 1, True)
            state = {k: s.get(k) for k in ("x", "y", "x_bar", "y_bar", "z_bar")}
            r = s.run()
            out[rank] = dict(lam=lam, it=it, res=res, state=state, run=r, info=s.dist_info(),
                             off=(s.row_off, s.m_loc, s.col_off, s.n_loc))
            s.close()
        except Exception as e:  # noqa: BLE001
            err[rank] = e

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    hprlp.Solver.free_local_group(group)
    assert all(e is None for e in err), err
    return out


def single(model, prm, steps):
    s = hprlp.Solver(model, prm)
    s.scale()
    lam, it = s.power_iteration()
    s.init(-1.0, lam * 1.01)
    s.iterate(steps, True)
    res = s.residuals(steps + 1, True)
    state = {k: s.get(k) for k in ("x", "y", "x_bar", "y_bar", "z_bar")}
    r = s.run()
    s.close()
    return dict(lam=lam, it=it, res=res, state=state, run=r)


def check_against_single(ref, ranks, m, n, obj_star):
    for o in ranks:
        row_off, m_loc, col_off, n_loc = o["off"]
        assert o["it"] == ref["it"] and abs(o["lam"] - ref["lam"]) <= 1e-12 * ref["lam"]
        for k in ("x", "x_bar", "z_bar"):
            np.testing.assert_allclose(o["state"][k][:n_loc], ref["state"][k][col_off:col_off + n_loc], rtol=1e-9, atol=1e-12, err_msg=k)
        for k in ("y", "y_bar"):
            np.testing.assert_allclose(o["state"][k][:m_loc], ref["state"][k][row_off:row_off + m_loc], rtol=1e-9, atol=1e-12, err_msg=k)
        for k in ref["res"]:
            assert abs(o["res"][k] - ref["res"][k]) <= 1e-9 * (1 + abs(ref["res"][k])), k
        assert o["run"].status == ref["run"].status == "OPTIMAL"
        assert abs(o["run"].primal_obj - obj_star) <= 1e-5 * (1 + abs(obj_star))
        assert abs(o["run"].iter - ref["run"].iter) <= 0.1 * ref["run"].iter + 150
    # the slices of the ranks tile the solution
    x = np.concatenate([o["run"].x[:o["off"][3]] for o in ranks])
    y = np.concatenate([o["run"].y[:o["off"][1]] for o in ranks])
    assert len(x) == n and len(y) == m
    np.testing.assert_allclose(x, ref["run"].x, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_solver_with_allgather_equals_single_gpu(gpu, world):
    """Unstructured LP: the shards read most remote entries, so the plan keeps the all-gather."""
    lp = lpgen.planted_lp(401, 653, 4000, 92)  # sizes not divisible by the world size
    model = hprlp.Model.from_csr(lp["m"], lp["n"], lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"],
                                 lp["l"], lp["u"], lp["c"])
    prm = hprlp.Parameters(stop_tol=1e-6, use_presolve=False)
    ref = single(model, prm, 37)
    ranks = run_ranks(model, prm, world, 37)
    assert all(o["info"]["m_sparse"] == 0 and o["info"]["n_sparse"] == 0 for o in ranks)
    check_against_single(ref, ranks, lp["m"], lp["n"], lp["obj_star"])
    model.free()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_solver_with_neighbour_exchange_equals_single_gpu(gpu, world):
    """Banded LP: every rank needs a halo and a few far entries only -> pack / grouped send-recv / scatter."""
    import bench_helpers as bh
    m = n = 6001
    lp = bh.banded_lp(m, n, 8, 150)
    model = hprlp.Model.from_csr(m, n, lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])
    prm = hprlp.Parameters(stop_tol=1e-6, use_presolve=False)
    ref = single(model, prm, 23)
    ranks = run_ranks(model, prm, world, 23)
    chunk = -(-n // world)
    for o in ranks:
        assert o["info"]["m_sparse"] == 1 and o["info"]["n_sparse"] == 1
        assert 0 < o["info"]["n_received"] < 0.6 * (n - chunk) and 0 < o["info"]["n_sent"]
    assert sum(o["info"]["n_sent"] for o in ranks) == sum(o["info"]["n_received"] for o in ranks) == ranks[0]["info"]["n_requests"]
    check_against_single(ref, ranks, m, n, lp["obj_star"])
    model.free()


def test_sharded_solver_with_tiled_kernels_on_the_shards(gpu):
    """Tiled kernels forced on the shards: they stage whole column tiles of the gathered vector, including entries
    the neighbour exchange never delivers (no matrix entry reads them) -- the iterates must not notice."""
    import os
    import bench_helpers as bh
    old = {k: os.environ.get(k) for k in ("HPRLP_TILED_MIN_ROWS", "HPRLP_TILED_MIN_DENSE")}
    os.environ["HPRLP_TILED_MIN_ROWS"] = "1"
    os.environ["HPRLP_TILED_MIN_DENSE"] = "0.0"
    try:
        m = n = 20000
        lp = bh.banded_lp(m, n, 10, 300)
        model = hprlp.Model.from_csr(m, n, lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])
        prm = hprlp.Parameters(stop_tol=1e-5, use_presolve=False)
        ref = single(model, prm, 23)
        ranks = run_ranks(model, prm, 2, 23)
        assert all(o["info"]["m_sparse"] == 1 and o["info"]["n_sparse"] == 1 for o in ranks)
        check_against_single(ref, ranks, m, n, lp["obj_star"])
        model.free()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_sharded_solver_with_tiled_shards_and_long_rows(gpu):
    """Tiled shards of a matrix with a few rows and columns of thousands of entries: on a shard the long rows are kept aside (base
    vector) AND the remote-column kernel adds the local-column part through the same kind of epilogue -- the two nest.  Same
    iterates as the single-GPU solver."""
    import os
    import bench_helpers as bh
    from scipy import sparse
    old = {k: os.environ.get(k) for k in ("HPRLP_TILED_MIN_ROWS", "HPRLP_TILED_MIN_DENSE")}
    os.environ["HPRLP_TILED_MIN_ROWS"] = "1"
    os.environ["HPRLP_TILED_MIN_DENSE"] = "0.0"
    try:
        m = n = 24000
        lp = bh.banded_lp(m, n, 10, 300)
        A = sparse.csr_matrix((lp["values"], lp["colind"], lp["rowptr"]), shape=(m, n))
        rng = np.random.default_rng(4)
        rr, cc, vv = [], [], []
        for i, L in zip(rng.choice(m, 3, replace=False), (1500, 3000, 5000)):
            c = rng.choice(n, L, replace=False); rr.append(np.full(L, i)); cc.append(c); vv.append(rng.normal(size=L) * 0.02)
        for j, L in zip(rng.choice(n, 3, replace=False), (1200, 2500, 4500)):
            r = rng.choice(m, L, replace=False); rr.append(r); cc.append(np.full(L, j)); vv.append(rng.normal(size=L) * 0.02)
        A = (A + sparse.csr_matrix((np.concatenate(vv), (np.concatenate(rr), np.concatenate(cc))), shape=(m, n))).tocsr()
        A.sort_indices()
        x0 = np.abs(rng.normal(size=n))
        b = A @ x0
        c = rng.normal(size=n)
        model = hprlp.Model.from_csr(m, n, A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data, b - 1.0, b + 1.0, np.zeros(n), np.full(n, 10.0), c)
        prm = hprlp.Parameters(stop_tol=1e-4, use_presolve=False, max_iter=20000)
        ref = single(model, prm, 23)
        ranks = run_ranks(model, prm, 2, 23)
        for o in ranks:
            row_off, m_loc, col_off, n_loc = o["off"]
            assert o["it"] == ref["it"] and abs(o["lam"] - ref["lam"]) <= 1e-11 * ref["lam"]
            for k in ("x", "x_bar", "z_bar"):
                np.testing.assert_allclose(o["state"][k][:n_loc], ref["state"][k][col_off:col_off + n_loc], rtol=1e-9, atol=1e-11, err_msg=k)
            for k in ("y", "y_bar"):
                np.testing.assert_allclose(o["state"][k][:m_loc], ref["state"][k][row_off:row_off + m_loc], rtol=1e-9, atol=1e-11, err_msg=k)
            for k in ref["res"]:
                assert abs(o["res"][k] - ref["res"][k]) <= 1e-8 * (1 + abs(ref["res"][k])), k
            assert o["run"].status == ref["run"].status
        model.free()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_failed_exchange_self_test_falls_back_to_the_all_gather(gpu):
    """Set-up self-test of the exchange (Solver::verify_exchange): a neighbour exchange reported as failed is replaced
    by the all-gather on every rank and the solve goes on to the single-GPU answer."""
    import os
    import bench_helpers as bh
    os.environ["HPRLP_DIST_SELFTEST_FAIL"] = "1"
    try:
        m = n = 6001
        lp = bh.banded_lp(m, n, 8, 150)
        model = hprlp.Model.from_csr(m, n, lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])
        prm = hprlp.Parameters(stop_tol=1e-6, use_presolve=False)
        ref = single(model, prm, 23)
        ranks = run_ranks(model, prm, 2, 23)
        assert all(o["info"]["m_sparse"] == 0 and o["info"]["n_sparse"] == 0 for o in ranks)
        check_against_single(ref, ranks, m, n, lp["obj_star"])
        model.free()
    finally:
        del os.environ["HPRLP_DIST_SELFTEST_FAIL"]


@pytest.mark.parametrize("switch", ["HPRLP_NO_OVERLAP", "HPRLP_OVERLAP_COMM_FIRST"])
def test_sharded_solver_other_forms_of_the_exchange(gpu, switch):
    """HPRLP_NO_OVERLAP=1: the unsplit shards with the exchange in line on the solver stream (the default splits every
    shard by columns and runs the exchange beside the local-column part; every other test here covers that).
    HPRLP_OVERLAP_COMM_FIRST=1: the launch order used with RCCL (exchange enqueued first, local SpMV after it)."""
    import os
    import bench_helpers as bh
    os.environ[switch] = "1"
    try:
        m = n = 6001
        lp = bh.banded_lp(m, n, 8, 150)
        model = hprlp.Model.from_csr(m, n, lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])
        prm = hprlp.Parameters(stop_tol=1e-6, use_presolve=False)
        ref = single(model, prm, 23)
        ranks = run_ranks(model, prm, 2, 23)
        check_against_single(ref, ranks, m, n, lp["obj_star"])
        model.free()
    finally:
        del os.environ[switch]


def test_time_limit_is_a_collective_decision(gpu):
    """A tiny time_limit: every rank must leave the loop with TIME_LIMIT at the SAME iteration (each rank reads its
    own clock; the stop flag is all-reduced -- a rank that went on alone would hang in the next exchange)."""
    import threading
    lp = lpgen.planted_lp(401, 653, 4000, 92)
    model = hprlp.Model.from_csr(lp["m"], lp["n"], lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"],
                                 lp["l"], lp["u"], lp["c"])
    world = 3
    group = hprlp.Solver.local_group(world)
    out, err = [None] * world, [None] * world

    def work(rank):
        try:
            # rank 0 is over its limit at the first event, the others would not be for an hour
            prm = hprlp.Parameters(stop_tol=1e-14, use_presolve=False, time_limit=0.0 if rank == 0 else 3600.0)
            s = hprlp.Solver.create_local(model, prm, rank, world, group)
            s.scale()
            lam, _ = s.power_iteration()
            s.init(-1.0, lam * 1.01)
            out[rank] = s.run()
            s.close()
        except Exception as e:  # noqa: BLE001
            err[rank] = e

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    hprlp.Solver.free_local_group(group)
    assert all(e is None for e in err), err
    assert {(o.status, o.iter) for o in out} == {("TIME_LIMIT", 0)}
    model.free()


def test_solver_from_caller_built_shards(gpu):
    """hprlp_solver_create_*_from_shard: the ranks get shards assembled outside the library (bench.py --gpus N builds them from
    per-rank rows, hpr-lp-c_amd/shard.py); same iterates and optimum as the single-GPU solver."""
    import threading

    from scipy import sparse

    from conftest import shardlib
    lp = lpgen.planted_lp(401, 653, 4000, 93)
    m, n = lp["m"], lp["n"]
    model = hprlp.Model.from_csr(m, n, lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])
    prm = hprlp.Parameters(stop_tol=1e-6, use_presolve=False)
    ref = single(model, prm, 37)
    A = sparse.csr_matrix((lp["values"], lp["colind"], lp["rowptr"]), shape=(m, n))
    AT = A.T.tocsr(); AT.sort_indices()
    world = 3
    group = hprlp.Solver.local_group(world)
    out, err = [None] * world, [None] * world

    def work(rank):
        try:
            _, ro, ml = shardlib.partition(m, world, rank)
            _, co, nl = shardlib.partition(n, world, rank)
            Ar, Tr = A[ro:ro + ml], AT[co:co + nl]
            sh = shardlib.ShardArrays(hprlp, m, n, rank, world, Ar.indptr, Ar.indices, Ar.data, Tr.indptr, Tr.indices, Tr.data,
                                      lp["AL"][ro:ro + ml], lp["AU"][ro:ro + ml], lp["l"][co:co + nl], lp["u"][co:co + nl], lp["c"][co:co + nl])
            s = hprlp.Solver.create_dist_from_shard(sh, prm, rank, world, group=group)
            s.scale()
            lam, it = s.power_iteration()
            s.init(-1.0, lam * 1.01)
            s.iterate(37, True)
            res = s.residuals(38, True)
            state = {k: s.get(k) for k in ("x", "y", "x_bar", "y_bar", "z_bar")}
            r = s.run()
            out[rank] = dict(lam=lam, it=it, res=res, state=state, run=r, info=s.dist_info(), off=(s.row_off, s.m_loc, s.col_off, s.n_loc))
            s.close()
        except Exception as e:  # noqa: BLE001
            err[rank] = e

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    hprlp.Solver.free_local_group(group)
    assert all(e is None for e in err), err
    check_against_single(ref, out, m, n, lp["obj_star"])
    model.free()


def process_ranks(kind, world, steps, tmp_path):
    """`world` ranks as separate PROCESSES on the one GPU over the shared-memory transport (HPRLP_DIST_TRANSPORT=shm)."""
    import pickle
    import subprocess
    import sys
    from types import SimpleNamespace
    here = os.path.dirname(os.path.abspath(__file__))
    os.environ["HPRLP_DIST_TRANSPORT"] = "shm"
    try:
        uid = hprlp.Solver.dist_unique_id(1)
    finally:
        os.environ.pop("HPRLP_DIST_TRANSPORT")
    assert bytes(uid[:8]) == b"HPRLPSHM"
    out = str(tmp_path / "ranks")
    procs = [subprocess.Popen([sys.executable, os.path.join(here, "dist_worker_shm_gpu.py"), str(r), str(world), str(steps), bytes(uid).hex(), kind, out],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=400)[0].decode())
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for r, (p, lg) in enumerate(zip(procs, logs)):
        assert p.returncode == 0 and f"rank {r} ok" in lg, lg[-3000:]
    ranks = []
    for r in range(world):
        rec = pickle.load(open(f"{out}.{r}.pkl", "rb"))
        rec["run"] = SimpleNamespace(**rec["run"])
        ranks.append(rec)
    return ranks


@pytest.mark.parametrize("kind,world,steps", [("unstructured", 3, 37), ("banded", 4, 23)])
def test_process_ranks_over_shared_memory_equal_single_gpu(gpu, tmp_path, kind, world, steps):
    """Round 5 (csrc/dist.cpp ShmComm): the sharded solver with one PROCESS per rank -- the shape of `bench.py --gpus N` -- on a
    one-GPU box: every rank on device 0, slices travelling device -> pinned shared area -> the reader's device.  All-gather
    plan (unstructured LP) and neighbour exchange (banded LP); iterates, residuals and the whole solve against the single-GPU
    solver, as for the thread ranks."""
    from dist_worker_shm_gpu import make_lp
    lp = make_lp(kind)
    model = hprlp.Model.from_csr(lp["m"], lp["n"], lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])
    prm = hprlp.Parameters(stop_tol=1e-6, use_presolve=False)
    ref = single(model, prm, steps)
    ranks = process_ranks(kind, world, steps, tmp_path)
    for o in ranks:
        assert o["comm"]["comm_ranks"] == world and o["comm"]["xcomm_ranks"] == 0, o["comm"]
        assert o["info"]["n_sparse"] == (1 if kind == "banded" else 0)
    check_against_single(ref, ranks, lp["m"], lp["n"], lp["obj_star"])
    model.free()


def test_bench_supervisor_cascade_ends_with_a_line_on_the_one_gpu_box(gpu):
    """`python bench.py --gpus 2` end to end on hardware, as far as a one-GPU box allows (HPRLP_BENCH_ONE_DEVICE=1: both ranks on
    device 0).  RCCL refuses two ranks on one device, so transport tiers 0-2 end at communicator creation -- real failures, each in
    fresh processes -- and tier 3 (host-staged shared memory between the rank PROCESSES) must produce the line: n_gpus 2, the tier
    and the earlier tiers' ends recorded, marked as a one-device rehearsal, the sharded solve to 1e-4 OPTIMAL on the planted
    objective.  (The supervisor's watchdog and the launcher form are covered on CPU: tests/test_dist_cpu.py.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "HPRLP_TEST_HOOKS")}
    env.update(HPRLP_BENCH_ONE_DEVICE="1", HPRLP_BENCH_STALL_S="120")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "c5_tiny", "--steps", "20", "--warmup", "5"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    err = r.stderr.decode()
    assert r.returncode == 0, err[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["finite"]
    tr = d["transport"]
    assert tr["tier"] == 3 and "shared memory" in tr["name"]
    assert [a["tier"] for a in tr["attempts"]] == [0, 1, 2, 3] and tr["attempts"][3]["ended"] == "ok"
    assert all("exited with code" in a["ended"] for a in tr["attempts"][:3]), tr["attempts"]
    assert "rehearsal" in d["config"]["rccl"] and d["config"]["rccl"]["transport"].startswith("shared memory")
    t = d["time_to_tol"]
    assert t["status"] == "OPTIMAL" and t["rel_obj_err"] < 1e-3, t
