// transpose.hip -- explicit A^T built on the device (the reference calls
// cusparseCsr2cscEx2, src/utils.cu:203-232).  Stable in row order like the host counting sort
// (csr_transpose_host): a stable radix sort of (column, entry index) pairs -- hipCUB/rocPRIM's device
// radix sort is the one library piece here; the rest is three small kernels -- followed by a gather of
// the values and of the entries' row numbers, and a binary search for the row pointers of A^T.
// Config 5 (2e8 nonzeros): ~25 ms against 1.0-1.4 s for the 8-thread host transpose.
#include <hipcub/hipcub.hpp>

#include "common.h"
#include "kernels.h"

namespace hprlp {

namespace {

__global__ void __launch_bounds__(kThreads) k_iota_keys(long nnz, const int *__restrict__ col, unsigned *__restrict__ key,
                                                       int *__restrict__ idx) {
    const long k = static_cast<long>(blockIdx.x) * kThreads + threadIdx.x;
    if (k < nnz) {
        key[k] = static_cast<unsigned>(col[k]);
        idx[k] = static_cast<int>(k);
    }
}

// entry p of A^T is entry perm[p] of A: its value, and its row (largest r with rowptr[r] <= perm[p])
__global__ void __launch_bounds__(kThreads) k_gather_transposed(long nnz, int rows, const int *__restrict__ perm,
                                                               const int *__restrict__ rowptr, const double *__restrict__ val,
                                                               int *__restrict__ tci, double *__restrict__ tv) {
    const long p = static_cast<long>(blockIdx.x) * kThreads + threadIdx.x;
    if (p >= nnz) return;
    const int k = perm[p];
    int lo = 0, hi = rows;  // invariant: rowptr[lo] <= k < rowptr[hi]
    while (hi - lo > 1) {
        const int mid = lo + ((hi - lo) >> 1);
        if (rowptr[mid] <= k) lo = mid;
        else hi = mid;
    }
    tci[p] = lo;
    tv[p] = val[k];
}

// trp[j] = first position whose (sorted) column is >= j
__global__ void __launch_bounds__(kThreads) k_row_starts(int cols, long nnz, const unsigned *__restrict__ sorted_key,
                                                        int *__restrict__ trp) {
    const int j = blockIdx.x * kThreads + threadIdx.x;
    if (j > cols) return;
    long lo = 0, hi = nnz;  // first p in [0, nnz] with key[p] >= j
    while (lo < hi) {
        const long mid = lo + ((hi - lo) >> 1);
        if (sorted_key[mid] < static_cast<unsigned>(j)) lo = mid + 1;
        else hi = mid;
    }
    trp[j] = static_cast<int>(lo);
}

}  // namespace

void device_transpose(int rows, int cols, long nnz, const int *rowptr, const int *col, const double *val, int *trp,
                      int *tci, double *tv, hipStream_t s) {
    if (nnz <= 0) {
        HIP_CHECK(hipMemsetAsync(trp, 0, sizeof(int) * (static_cast<size_t>(cols) + 1), s));
        return;
    }
    DBuf<unsigned> key_in(static_cast<size_t>(nnz)), key_out(static_cast<size_t>(nnz));
    DBuf<int> idx_in(static_cast<size_t>(nnz)), idx_out(static_cast<size_t>(nnz));
    const unsigned grid = static_cast<unsigned>((nnz + kThreads - 1) / kThreads);
    hipLaunchKernelGGL(k_iota_keys, dim3(grid), dim3(kThreads), 0, s, nnz, col, key_in.p, idx_in.p);
    int bits = 1;
    while (bits < 32 && (1L << bits) < static_cast<long>(cols)) ++bits;
    size_t tmp_bytes = 0;
    HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, key_in.p, key_out.p, idx_in.p, idx_out.p,
                                                 static_cast<int>(nnz), 0, bits, s));
    DBuf<char> tmp(tmp_bytes + 16);
    HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, key_in.p, key_out.p, idx_in.p, idx_out.p,
                                                 static_cast<int>(nnz), 0, bits, s));
    hipLaunchKernelGGL(k_gather_transposed, dim3(grid), dim3(kThreads), 0, s, nnz, rows, idx_out.p, rowptr, val, tci, tv);
    hipLaunchKernelGGL(k_row_starts, dim3(static_cast<unsigned>((cols + 1 + kThreads - 1) / kThreads)), dim3(kThreads), 0, s,
                       cols, nnz, key_out.p, trp);
    HIP_CHECK(hipStreamSynchronize(s));  // the temporaries are released on return
}

// ------------------------------------------------------------------------------------------------
// Column split of a device CSR matrix (multi-GPU overlap, solver.cpp prepare_overlap): the entries whose column lies
// in [lo, hi) -- the part of the gathered vector this rank owns -- go to one CSR matrix, the others to a second one;
// both keep all rows and the CSR order inside a row.
// ------------------------------------------------------------------------------------------------
namespace {

__global__ void __launch_bounds__(kThreads) k_count_local(int rows, const int *__restrict__ rowptr, const int *__restrict__ col, int lo,
                                                         int hi, int *__restrict__ cnt) {
    const int r = blockIdx.x * kThreads + threadIdx.x;
    if (r > rows) return;
    int c = 0;
    if (r < rows)
        for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) c += (col[k] >= lo && col[k] < hi);
    cnt[r] = c;  // cnt[rows] = 0: the scan's last element is the total
}

__global__ void __launch_bounds__(kThreads) k_split_rows(int rows, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                        const double *__restrict__ val, int lo, int hi, const int *__restrict__ rp_loc,
                                                        int *__restrict__ rp_rem, int *__restrict__ col_loc, double *__restrict__ val_loc,
                                                        int *__restrict__ col_rem, double *__restrict__ val_rem) {
    const int r = blockIdx.x * kThreads + threadIdx.x;
    if (r > rows) return;
    int a = rp_loc[r], b = rowptr[r] - a;
    rp_rem[r] = b;
    if (r == rows) return;
    for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) {
        const int c = col[k];
        if (c >= lo && c < hi) {
            col_loc[a] = c;
            val_loc[a++] = val[k];
        } else {
            col_rem[b] = c;
            val_rem[b++] = val[k];
        }
    }
}

}  // namespace

void device_split_columns(int rows, long nnz, const int *rowptr, const int *col, const double *val, int lo, int hi, DBuf<int> &rp_loc,
                          DBuf<int> &col_loc, DBuf<double> &val_loc, DBuf<int> &rp_rem, DBuf<int> &col_rem, DBuf<double> &val_rem,
                          hipStream_t s) {
    const unsigned grid = static_cast<unsigned>((static_cast<long>(rows) + 1 + kThreads - 1) / kThreads);
    DBuf<int> cnt(static_cast<size_t>(rows) + 1);
    rp_loc.alloc(static_cast<size_t>(rows) + 1);
    rp_rem.alloc(static_cast<size_t>(rows) + 1);
    hipLaunchKernelGGL(k_count_local, dim3(grid), dim3(kThreads), 0, s, rows, rowptr, col, lo, hi, cnt.p);
    size_t tmp_bytes = 0;
    HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, cnt.p, rp_loc.p, rows + 1, s));
    DBuf<char> tmp(tmp_bytes);
    HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(tmp.p, tmp_bytes, cnt.p, rp_loc.p, rows + 1, s));
    int n_loc = 0;
    HIP_CHECK(hipMemcpyAsync(&n_loc, rp_loc.p + rows, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    const long n_rem = nnz - n_loc;
    col_loc.alloc(static_cast<size_t>(n_loc));
    val_loc.alloc(static_cast<size_t>(n_loc));
    col_rem.alloc(static_cast<size_t>(n_rem));
    val_rem.alloc(static_cast<size_t>(n_rem));
    hipLaunchKernelGGL(k_split_rows, dim3(grid), dim3(kThreads), 0, s, rows, rowptr, col, val, lo, hi, rp_loc.p, rp_rem.p, col_loc.p,
                       val_loc.p, col_rem.p, val_rem.p);
    HIP_CHECK(hipStreamSynchronize(s));
}

// warm-up (abi.cpp: hprlp_warmup): an attribute query makes the runtime load this translation unit's code object now instead
// of at the first launch of one of its kernels
void warm_transpose_tu() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&k_iota_keys));
}

}  // namespace hprlp
