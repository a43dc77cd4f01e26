"""Per-rank assembly of a row-partitioned LP (include/hprlp_amd.h: hprlp_shard) without any rank holding the whole matrix.

Every rank brings its rows of A (CSR, global column indices) and its slices of the vectors.  The rows of A^T a rank
owns are the COLUMNS [col_off, col_off + n_loc) of A, whose entries sit in every rank's rows: one all-to-all of
(column, row, value) triples over torch.distributed moves each entry to the owner of its column, a sort by (column, row)
turns what arrives into CSR -- the distributed form of the stable counting-sort transpose the reference does on one host
(reference src/utils.cu:203-232).  With world == 1 (dist None) it is a local transpose.
"""
import ctypes as C

import numpy as np


def partition(total, parts, rank):
    chunk = -(-total // parts)
    off = min(total, rank * chunk)
    return chunk, off, max(0, min(total, off + chunk) - off)


def transpose_rows_distributed(m, n, row_off, rp, ci, v, rank, world, dist=None):
    """rows [row_off, row_off + len(rp) - 1) of A  ->  rows [col_off, col_off + n_loc) of A^T (rowptr, col = global row, val)."""
    chunk_n, col_off, n_loc = partition(n, world, rank)
    rows = np.repeat(np.arange(row_off, row_off + len(rp) - 1, dtype=np.int64), np.diff(rp))
    ci = np.asarray(ci, np.int64)
    v = np.asarray(v, np.float64)
    if world == 1 or dist is None:
        cols, grow, vals = ci, rows, v
    else:
        import torch
        dest = ci // chunk_n
        order = np.argsort(dest, kind="stable")
        counts = np.bincount(dest, minlength=world).astype(np.int64)
        send_counts = torch.from_numpy(counts.copy())
        recv_counts = torch.zeros(world, dtype=torch.int64)
        dist.all_to_all_single(recv_counts, send_counts)
        rc = recv_counts.numpy()
        out = []
        for arr, dt in ((ci[order], torch.int64), (rows[order], torch.int64), (v[order], torch.float64)):
            send = torch.from_numpy(np.ascontiguousarray(arr))
            recv = torch.zeros(int(rc.sum()), dtype=dt)
            dist.all_to_all_single(recv, send, output_split_sizes=[int(k) for k in rc], input_split_sizes=[int(k) for k in counts])
            out.append(recv.numpy())
        cols, grow, vals = out
    key = (cols - col_off) * np.int64(m) + grow           # (local column, global row): rows ascending inside a column
    order = np.argsort(key, kind="stable")
    lc = (cols - col_off)[order]
    trp = np.zeros(n_loc + 1, np.int64)
    np.add.at(trp, lc + 1, 1)
    trp = np.cumsum(trp)
    return trp.astype(np.int32), grow[order].astype(np.int32), np.ascontiguousarray(vals[order])


def transposed_rows_generated(hprlp, m, n, per_row, band, seed, rank, world, threads=0):
    """Rows [col_off, col_off + n_loc) of A^T for the banded benchmark generator WITHOUT communication: the generator is a pure
    function of (seed, row), so the rank sweeps all m rows in C (csrc/gen.cpp: hprlp_gen_banded_csr_transposed) and keeps the
    entries whose column it owns.  Entry for entry what transpose_rows_distributed delivers (tests/test_dist_cpu.py), at a
    fraction of its time and memory: no 2e8-triple all-to-all over gloo, no 25 M-key argsort per rank."""
    _, col_off, n_loc = partition(n, world, rank)
    L = hprlp.lib()
    ip, dp = hprlp.c_int_p, hprlp.c_dbl_p
    L.hprlp_gen_banded_csr_transposed.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_ulonglong, C.c_int, C.c_int, ip,
                                                  C.POINTER(ip), C.POINTER(dp), C.POINTER(C.c_long), C.c_int]
    L.hprlp_host_free.argtypes = [C.c_void_p]
    L.hprlp_host_free.restype = None
    trp = np.zeros(n_loc + 1, np.int32)
    pci, pv, nnz = ip(), dp(), C.c_long(0)
    if L.hprlp_gen_banded_csr_transposed(m, n, per_row, band, seed, col_off, n_loc, trp.ctypes.data_as(ip), C.byref(pci), C.byref(pv),
                                         C.byref(nnz), threads) != 0:
        raise RuntimeError(hprlp.last_error())
    k = int(nnz.value)
    try:
        tci = np.ctypeslib.as_array(pci, shape=(max(k, 1),))[:k].copy()
        tv = np.ctypeslib.as_array(pv, shape=(max(k, 1),))[:k].copy()
    finally:
        L.hprlp_host_free(C.cast(pci, C.c_void_p))
        L.hprlp_host_free(C.cast(pv, C.c_void_p))
    return trp, tci, tv


class ShardArrays:
    """Owns the numpy arrays of one rank's shard and exposes them as a ctypes hprlp_shard."""

    def __init__(self, hprlp, m, n, rank, world, A_rp, A_ci, A_v, AT_rp, AT_ci, AT_v, AL, AU, l, u, c, obj_constant=0.0):
        _, self.row_off, self.m_loc = partition(m, world, rank)
        _, self.col_off, self.n_loc = partition(n, world, rank)
        f = lambda a: np.ascontiguousarray(a, np.float64)
        i = lambda a: np.ascontiguousarray(a, np.int32)
        self.keep = dict(A_rp=i(A_rp), A_ci=i(A_ci), A_v=f(A_v), AT_rp=i(AT_rp), AT_ci=i(AT_ci), AT_v=f(AT_v), AL=f(AL), AU=f(AU),
                         l=f(l), u=f(u), c=f(c))
        k = self.keep
        assert len(k["A_rp"]) == self.m_loc + 1 and len(k["AT_rp"]) == self.n_loc + 1
        assert len(k["AL"]) == len(k["AU"]) == self.m_loc and len(k["l"]) == len(k["u"]) == len(k["c"]) == self.n_loc
        ip, dp = hprlp.c_int_p, hprlp.c_dbl_p
        P = lambda a, t: a.ctypes.data_as(t)
        self.c_shard = hprlp.CShard(m=m, n=n, row_off=self.row_off, m_loc=self.m_loc, col_off=self.col_off, n_loc=self.n_loc,
                                    A_rowptr=P(k["A_rp"], ip), A_col=P(k["A_ci"], ip), A_val=P(k["A_v"], dp),
                                    AT_rowptr=P(k["AT_rp"], ip), AT_col=P(k["AT_ci"], ip), AT_val=P(k["AT_v"], dp),
                                    AL=P(k["AL"], dp), AU=P(k["AU"], dp), l=P(k["l"], dp), u=P(k["u"], dp), c=P(k["c"], dp),
                                    obj_constant=float(obj_constant))
        self.m, self.n = m, n
