// reorder_dev.hip -- the data-parallel parts of the set-up time locality ordering (reorder.cpp) on the device: the
// median sweeps (200M random gathers per sweep on config 5: ~10 s of host time, milliseconds here), the rank
// normalisation and the final argsorts (radix sorts), and the permuted copy P A Q of the matrix.  The clustering and
// the spectral ordering of the small cluster graph stay on the host (reorder.cpp, cluster_positions).
#include <hipcub/hipcub.hpp>

#include "common.h"
#include "kernels.h"
#include "reorder.h"

namespace hprlp {

namespace {

inline unsigned grid_for(long n) { return static_cast<unsigned>((n + kThreads - 1) / kThreads); }

// out[i] = robust centre of src over the neighbours of i (at most 32, evenly sampled): mean of the middle 60 % of the sorted
// sample (reorder.cpp, centre_sweep: same rule); nodes without neighbours keep their value
__global__ void __launch_bounds__(kThreads) k_centre_sweep(int rows, const int *__restrict__ rp, const int *__restrict__ ci,
                                                          const double *__restrict__ src, double *__restrict__ out) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= rows) return;
    const int k0 = rp[i], len = rp[i + 1] - k0;
    if (len <= 0) return;
    const int take = min(len, 32);
    double buf[32];
    for (int q = 0; q < take; ++q) {
        const double v = src[ci[k0 + static_cast<int>(static_cast<long>(q) * len / take)]];
        int p = q;  // insertion sort
        while (p > 0 && buf[p - 1] > v) {
            buf[p] = buf[p - 1];
            --p;
        }
        buf[p] = v;
    }
    const int lo = take / 5, hi = take - take / 5;
    double sum = 0.0;
    for (int q = lo; q < hi; ++q) sum += buf[q];
    out[i] = sum / static_cast<double>(hi - lo);
}

__global__ void __launch_bounds__(kThreads) k_pos_keys(int n, const double *__restrict__ pos, unsigned long long *__restrict__ key,
                                                      int *__restrict__ idx) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    key[i] = static_cast<unsigned long long>(__double_as_longlong(pos[i]));  // positions are positive: the bit pattern sorts like the value
    idx[i] = i;
}

__global__ void __launch_bounds__(kThreads) k_assign_ranks(int n, const int *__restrict__ sorted_idx, double *__restrict__ pos) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    pos[sorted_idx[i]] = (static_cast<double>(i) + 0.5) / static_cast<double>(n);
}

__global__ void __launch_bounds__(kThreads) k_invert(int n, const int *__restrict__ new2old, int *__restrict__ old2new) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i < n) old2new[new2old[i]] = i;
}

__global__ void __launch_bounds__(kThreads) k_new_lengths(int m, const int *__restrict__ rp, const int *__restrict__ new2old, int *__restrict__ len) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i > m) return;
    if (i == m) {
        len[m] = 0;
        return;
    }
    const int o = new2old[i];
    len[i] = rp[o + 1] - rp[o];
}

__global__ void __launch_bounds__(kThreads) k_entry_keys(long nnz, int rows, const int *__restrict__ rp, const int *__restrict__ ci,
                                                        const int *__restrict__ r_old2new, const int *__restrict__ c_old2new,
                                                        unsigned long long *__restrict__ key, int *__restrict__ idx) {
    const long k = static_cast<long>(blockIdx.x) * kThreads + threadIdx.x;
    if (k >= nnz) return;
    int lo = 0, hi = rows;  // rp[lo] <= k < rp[hi]
    while (hi - lo > 1) {
        const int mid = lo + ((hi - lo) >> 1);
        if (rp[mid] <= k) lo = mid;
        else hi = mid;
    }
    key[k] = (static_cast<unsigned long long>(r_old2new[lo]) << 32) | static_cast<unsigned long long>(c_old2new[ci[k]]);
    idx[k] = static_cast<int>(k);
}

__global__ void __launch_bounds__(kThreads) k_permuted_entries(long nnz, const unsigned long long *__restrict__ skey, const int *__restrict__ sidx,
                                                              const double *__restrict__ val, int *__restrict__ ci_out, double *__restrict__ val_out) {
    const long p = static_cast<long>(blockIdx.x) * kThreads + threadIdx.x;
    if (p >= nnz) return;
    ci_out[p] = static_cast<int>(skey[p] & 0xffffffffull);
    val_out[p] = val[sidx[p]];
}

// stable argsort of the positions into sorted_idx, then pos <- (rank + 0.5) / n
void rank_normalise(int n, double *pos, DBuf<unsigned long long> &kin, DBuf<unsigned long long> &kout, DBuf<int> &vin, int *sorted_idx,
                    hipStream_t s) {
    hipLaunchKernelGGL(k_pos_keys, dim3(grid_for(n)), dim3(kThreads), 0, s, n, pos, kin.p, vin.p);
    size_t bytes = 0;
    HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, kin.p, kout.p, vin.p, sorted_idx, n, 0, 64, s));
    DBuf<char> tmp(bytes + 16);
    HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(tmp.p, bytes, kin.p, kout.p, vin.p, sorted_idx, n, 0, 64, s));
    hipLaunchKernelGGL(k_assign_ranks, dim3(grid_for(n)), dim3(kThreads), 0, s, n, sorted_idx, pos);
    HIP_CHECK(hipStreamSynchronize(s));
}

}  // namespace

void device_refine_order(int m, int n, const int *rp, const int *ci, const int *trp, const int *tci, double *pos_r, double *pos_c,
                         int sweeps, int *row_new2old, int *col_new2old, hipStream_t s) {
    const int big = std::max(m, n);
    DBuf<unsigned long long> kin(static_cast<size_t>(big)), kout(static_cast<size_t>(big));
    DBuf<int> vin(static_cast<size_t>(big));
    rank_normalise(m, pos_r, kin, kout, vin, row_new2old, s);
    for (int sw = 0; sw < sweeps; ++sw) {
        hipLaunchKernelGGL(k_centre_sweep, dim3(grid_for(n)), dim3(kThreads), 0, s, n, trp, tci, pos_r, pos_c);
        rank_normalise(n, pos_c, kin, kout, vin, col_new2old, s);
        hipLaunchKernelGGL(k_centre_sweep, dim3(grid_for(m)), dim3(kThreads), 0, s, m, rp, ci, pos_c, pos_r);
        rank_normalise(m, pos_r, kin, kout, vin, row_new2old, s);
    }
    if (sweeps <= 0) rank_normalise(n, pos_c, kin, kout, vin, col_new2old, s);
    HIP_CHECK(hipStreamSynchronize(s));
}

void device_permute_csr(int m, int n, long nnz, const int *rp, const int *ci, const double *val, const int *row_new2old,
                        const int *col_new2old, int *rp_out, int *ci_out, double *val_out, hipStream_t s) {
    DBuf<int> r_old2new(static_cast<size_t>(m)), c_old2new(static_cast<size_t>(n)), len(static_cast<size_t>(m) + 1);
    hipLaunchKernelGGL(k_invert, dim3(grid_for(m)), dim3(kThreads), 0, s, m, row_new2old, r_old2new.p);
    hipLaunchKernelGGL(k_invert, dim3(grid_for(n)), dim3(kThreads), 0, s, n, col_new2old, c_old2new.p);
    hipLaunchKernelGGL(k_new_lengths, dim3(grid_for(m + 1)), dim3(kThreads), 0, s, m, rp, row_new2old, len.p);
    {
        size_t bytes = 0;
        HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, len.p, rp_out, m + 1, s));
        DBuf<char> tmp(bytes + 16);
        HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(tmp.p, bytes, len.p, rp_out, m + 1, s));
        HIP_CHECK(hipStreamSynchronize(s));
    }
    DBuf<unsigned long long> kin(static_cast<size_t>(nnz)), kout(static_cast<size_t>(nnz));
    DBuf<int> vin(static_cast<size_t>(nnz)), vout(static_cast<size_t>(nnz));
    hipLaunchKernelGGL(k_entry_keys, dim3(grid_for(nnz)), dim3(kThreads), 0, s, nnz, m, rp, ci, r_old2new.p, c_old2new.p, kin.p, vin.p);
    int row_bits = 1;
    while ((1L << row_bits) < m) ++row_bits;
    {
        size_t bytes = 0;
        HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, kin.p, kout.p, vin.p, vout.p, static_cast<int>(nnz), 0, 32 + row_bits, s));
        DBuf<char> tmp(bytes + 16);
        HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(tmp.p, bytes, kin.p, kout.p, vin.p, vout.p, static_cast<int>(nnz), 0, 32 + row_bits, s));
        HIP_CHECK(hipStreamSynchronize(s));
    }
    hipLaunchKernelGGL(k_permuted_entries, dim3(grid_for(nnz)), dim3(kThreads), 0, s, nnz, kout.p, vout.p, val, ci_out, val_out);
    HIP_CHECK(hipStreamSynchronize(s));
}

}  // namespace hprlp
