"""Worker of tests/test_dist_cpu.py::test_bench_shard_generation: bench.py's per-rank generation of the benchmark LP
(every rank generates its own rows, the rows of A^T arrive through one all-to-all) must give exactly the shard that
hprlp_extract_shard cuts out of the LP generated whole -- same matrix slices, same vector slices, same planted optimum."""
import ctypes as C
import os
import sys

import numpy as np
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import bench_helpers as bh  # noqa: E402
from conftest import hprlp  # noqa: E402
from dist_worker import Shard, arr  # noqa: E402


def main():
    rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m, n, per_row, band = 30011, 20003, 7, 300      # ragged sizes, rectangular
    shard, obj_star, nnz_loc = bh._b.banded_lp_shard(m, n, per_row, band, rank, world, dist)
    lp = bh.banded_lp(m, n, per_row, band)
    assert abs(obj_star - lp["obj_star"]) <= 1e-9 * (1 + abs(lp["obj_star"]))
    model = hprlp.Model.from_csr(m, n, lp["rowptr"], lp["colind"], lp["values"], lp["AL"], lp["AU"], lp["l"], lp["u"], lp["c"])
    L = hprlp.lib()
    sh = Shard()
    L.hprlp_extract_shard.argtypes = [C.POINTER(hprlp.CLPInfo), C.c_int, C.c_int, C.POINTER(Shard)]
    assert L.hprlp_extract_shard(model._ptr, rank, world, C.byref(sh)) == 0, hprlp.last_error()
    k = shard.keep
    assert (shard.row_off, shard.m_loc, shard.col_off, shard.n_loc) == (sh.row_off, sh.m_loc, sh.col_off, sh.n_loc)
    nzA = int(k["A_rp"][-1]); nzT = int(k["AT_rp"][-1])
    assert nzA == nnz_loc
    for name, got, want in (("A_rowptr", k["A_rp"], arr(sh.A_rowptr, sh.m_loc + 1, np.int32)), ("A_col", k["A_ci"], arr(sh.A_col, nzA, np.int32)),
                            ("A_val", k["A_v"], arr(sh.A_val, nzA, np.float64)), ("AT_rowptr", k["AT_rp"], arr(sh.AT_rowptr, sh.n_loc + 1, np.int32)),
                            ("AT_col", k["AT_ci"], arr(sh.AT_col, nzT, np.int32)), ("AT_val", k["AT_v"], arr(sh.AT_val, nzT, np.float64)),
                            ("l", k["l"], arr(sh.l, sh.n_loc, np.float64)), ("u", k["u"], arr(sh.u, sh.n_loc, np.float64))):
        assert np.array_equal(got, want), name
    # b = A x and c = A^T y + z are sums over a row: the shard sums its own row / column in the same order as the whole LP
    for name, got, want in (("AL", k["AL"], arr(sh.AL, sh.m_loc, np.float64)), ("AU", k["AU"], arr(sh.AU, sh.m_loc, np.float64)),
                            ("c", k["c"], arr(sh.c, sh.n_loc, np.float64))):
        fin = np.isfinite(want)
        assert np.array_equal(np.isfinite(got), fin), name
        np.testing.assert_allclose(got[fin], want[fin], rtol=1e-12, atol=1e-13, err_msg=name)
    L.hprlp_free_shard.argtypes = [C.POINTER(Shard)]
    L.hprlp_free_shard(C.byref(sh))
    model.free()
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} ok")


if __name__ == "__main__":
    main()
