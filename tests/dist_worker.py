"""Worker of tests/test_dist_cpu.py: one rank of the row-partitioned HPR iteration on the CPU.

Exercises, over gloo with world_size>1, exactly the structure the GPU path runs over RCCL
(hpr-lp-c_amd/csrc/dist.cpp, solver.cpp): shard extraction by the library's host code
(hprlp_extract_shard), local half-steps on the shard (the oracle's kernels stand in for the HIP
kernels -- this is a test), one all-gather of the fresh slice after each half-step, and all-reduced
reduction scalars.  The sharded iterates must equal the single-process oracle's bit for bit.
"""
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from conftest import hprlp, lpgen, shardlib  # noqa: E402
from oracle import oracle as O  # noqa: E402


class Shard(C.Structure):
    _fields_ = [("m", C.c_int), ("n", C.c_int), ("row_off", C.c_int), ("m_loc", C.c_int), ("col_off", C.c_int),
                ("n_loc", C.c_int), ("A_rowptr", hprlp.c_int_p), ("A_col", hprlp.c_int_p), ("A_val", hprlp.c_dbl_p),
                ("AT_rowptr", hprlp.c_int_p), ("AT_col", hprlp.c_int_p), ("AT_val", hprlp.c_dbl_p),
                ("AL", hprlp.c_dbl_p), ("AU", hprlp.c_dbl_p), ("l", hprlp.c_dbl_p), ("u", hprlp.c_dbl_p),
                ("c", hprlp.c_dbl_p), ("obj_constant", C.c_double)]


def arr(p, n, dt):
    return np.ctypeslib.as_array(p, shape=(max(n, 1),))[:n].astype(dt).copy() if n > 0 else np.zeros(0, dt)


def gather(local, chunk, world):
    buf = np.zeros(chunk)
    buf[:len(local)] = local
    outs = [torch.zeros(chunk, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(outs, torch.from_numpy(buf))
    return torch.cat(outs).numpy()


def main():
    rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m, n = 203, 317                       # not divisible by the world size: ragged last shard
    lp = lpgen.planted_lp(m, n, 2200, 77)
    # every rank scales the full problem identically (scaling is deterministic); the shard is cut from the scaled LP
    ref = O.ScaledLP(m, n, lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])
    model = hprlp.Model.from_csr(m, n, ref.Arp, ref.Aci, ref.Av, ref.AL, ref.AU, ref.l, ref.u, ref.c)
    L = hprlp.lib()
    sh = Shard()
    L.hprlp_extract_shard.argtypes = [C.POINTER(hprlp.CLPInfo), C.c_int, C.c_int, C.POINTER(Shard)]
    assert L.hprlp_extract_shard(model._ptr, rank, world, C.byref(sh)) == 0, hprlp.last_error()
    off = C.c_int(); cnt = C.c_int()
    chunk_m = L.hprlp_partition(m, world, rank, C.byref(off), C.byref(cnt))
    assert (off.value, cnt.value) == (sh.row_off, sh.m_loc)
    chunk_n = L.hprlp_partition(n, world, rank, C.byref(off), C.byref(cnt))
    assert (off.value, cnt.value) == (sh.col_off, sh.n_loc)
    assert chunk_m == -(-m // world) and chunk_n == -(-n // world)
    # the shard is the right slice of A and of A^T
    Arp = arr(sh.A_rowptr, sh.m_loc + 1, np.int32); nzA = int(Arp[-1])
    Aci = arr(sh.A_col, nzA, np.int32); Av = arr(sh.A_val, nzA, np.float64)
    k0, k1 = ref.Arp[sh.row_off], ref.Arp[sh.row_off + sh.m_loc]
    assert np.array_equal(Aci, ref.Aci[k0:k1]) and np.array_equal(Av, ref.Av[k0:k1])
    ATrp = arr(sh.AT_rowptr, sh.n_loc + 1, np.int32); nzT = int(ATrp[-1])
    ATci = arr(sh.AT_col, nzT, np.int32); ATv = arr(sh.AT_val, nzT, np.float64)
    k0, k1 = ref.ATrp[sh.col_off], ref.ATrp[sh.col_off + sh.n_loc]
    assert np.array_equal(ATci, ref.ATci[k0:k1]) and np.array_equal(ATv, ref.ATv[k0:k1])
    # the from-shard path (bench.py --gpus N): a rank that only ever sees ITS rows of A gets its rows of A^T through one
    # all-to-all (hpr-lp-c_amd/shard.py) -- identical, entry for entry, to the slice cut out of the full transpose
    my_rp = (ref.Arp[sh.row_off:sh.row_off + sh.m_loc + 1] - ref.Arp[sh.row_off]).astype(np.int32)
    t_rp, t_ci, t_v = shardlib.transpose_rows_distributed(m, n, sh.row_off, my_rp, Aci, Av, rank, world, dist)
    assert np.array_equal(t_rp, ATrp) and np.array_equal(t_ci, ATci) and np.array_equal(t_v, ATv)
    assert shardlib.partition(m, world, rank) == (chunk_m, sh.row_off, sh.m_loc)
    assert shardlib.partition(n, world, rank) == (chunk_n, sh.col_off, sh.n_loc)
    AL = arr(sh.AL, sh.m_loc, float); AU = arr(sh.AU, sh.m_loc, float)
    l = arr(sh.l, sh.n_loc, float); u = arr(sh.u, sh.n_loc, float); c = arr(sh.c, sh.n_loc, float)
    assert np.array_equal(c, ref.c[sh.col_off:sh.col_off + sh.n_loc])

    # sharded iteration: local state slices + gathered y and x_hat
    P = lambda a: a.ctypes.data_as(hprlp.c_dbl_p)
    I = lambda a: a.ctypes.data_as(hprlp.c_int_p)
    ol = O.lib()
    nl, ml = sh.n_loc, sh.m_loc
    x = np.zeros(nl); xh = np.zeros(nl); xb = np.zeros(nl); zb = np.zeros(nl); xt = np.zeros(nl); lx = np.zeros(nl)
    y = np.zeros(ml); yb = np.zeros(ml); yo = np.zeros(ml); yt = np.zeros(ml); ly = np.zeros(ml)
    y_full = np.zeros(chunk_m * world); xh_full = np.zeros(chunk_n * world)
    sigma, lam = 0.8, 1.9
    st = ref.new_state()
    K = 40
    for k in range(K):
        chk = int(k == K - 1)
        ol.orc_x_half(nl, I(ATrp), I(ATci), P(ATv), P(y_full), P(x), P(xh), P(xb), P(zb), P(xt), P(l), P(u), P(c),
                      P(lx), C.c_double(sigma), k, chk)
        xh_full = gather(xh, chunk_n, world)                      # RCCL all-gather #1 on the GPU path
        ol.orc_y_half(ml, I(Arp), I(Aci), P(Av), P(xh_full), P(y), P(yb), P(yo), P(yt), P(AL), P(AU), P(ly),
                      C.c_double(sigma), C.c_double(lam), k, chk)
        y_full = gather(y, chunk_m, world)                        # RCCL all-gather #2
        ref.x_half(st, sigma, k, chk)
        ref.y_half(st, sigma, lam, k, chk)
    ro, co = sh.row_off, sh.col_off
    for name, loc, o, cnt_ in (("x", x, co, nl), ("x_hat", xh, co, nl), ("x_bar", xb, co, nl), ("z_bar", zb, co, nl),
                               ("x_temp", xt, co, nl), ("y", y, ro, ml), ("y_bar", yb, ro, ml), ("y_obj", yo, ro, ml),
                               ("y_temp", yt, ro, ml)):
        assert np.array_equal(loc, st[name][o:o + cnt_]), name     # bit-exact: same per-row arithmetic
    assert np.array_equal(y_full[:m], st["y"]) and np.array_equal(xh_full[:n], st["x_hat"])
    # reduction scalars: local partial sums, one all-reduce
    part = torch.tensor([c @ xb, yo @ yb, xb @ zb, xt @ xt, yt @ yt], dtype=torch.float64)
    dist.all_reduce(part)
    want = np.array([ref.c @ st["x_bar"], st["y_obj"] @ st["y_bar"], st["x_bar"] @ st["z_bar"],
                     st["x_temp"] @ st["x_temp"], st["y_temp"] @ st["y_temp"]])
    np.testing.assert_allclose(part.numpy(), want, rtol=1e-12, atol=1e-14)
    L.hprlp_free_shard.argtypes = [C.POINTER(Shard)]
    L.hprlp_free_shard(C.byref(sh))
    model.free()
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} ok")


if __name__ == "__main__":
    main()
