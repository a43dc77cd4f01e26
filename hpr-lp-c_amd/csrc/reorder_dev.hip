// reorder_dev.hip -- the data-parallel parts of the set-up time locality ordering (reorder.cpp) on the device: the
// median sweeps (200M random gathers per sweep on config 5: ~10 s of host time, milliseconds here), the rank
// normalisation and the final argsorts (radix sorts), the permuted copy P A Q of the matrix, the tiling test, and the
// graph work of the clustering (Voronoi BFS, majority relabelling, cluster-to-cluster edge counts).  Only the spectral
// ordering of the small cluster graph stays on the host (reorder.cpp, cluster_order).
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "env.h"
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


// ---- clustering (reorder.cpp, cluster_positions: the same rules, level-synchronous)

__global__ void __launch_bounds__(kThreads) k_fill_int(long n, int *__restrict__ a, int v) {
    const long i = static_cast<long>(blockIdx.x) * kThreads + threadIdx.x;
    if (i < n) a[i] = v;
}

__global__ void __launch_bounds__(kThreads) k_seed_rows(int K, int m, int *__restrict__ lab_r) {
    const int k = blockIdx.x * kThreads + threadIdx.x;
    if (k < K) lab_r[static_cast<long>(k) * m / K] = k;
}

// an unlabelled node takes the label of its first labelled neighbour (adjacency order); src is not written by this launch
__global__ void __launch_bounds__(kThreads) k_pull_labels(int nodes, const int *__restrict__ xp, const int *__restrict__ xi,
                                                         const int *__restrict__ src, int *__restrict__ dst, int *__restrict__ changed) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= nodes || dst[i] >= 0) return;
    for (int k = xp[i], e = xp[i + 1]; k < e; ++k) {
        const int l = src[xi[k]];
        if (l >= 0) {
            dst[i] = l;
            atomicAdd(changed, 1);
            return;
        }
    }
}

// dst[i] = the most frequent label among (at most 32, evenly sampled) neighbours; ties: the smallest label; no labelled
// neighbour: dst keeps what it holds (the caller seeds it with the node's own label)
__global__ void __launch_bounds__(kThreads) k_majority(int nodes, const int *__restrict__ xp, const int *__restrict__ xi,
                                                      const int *__restrict__ src, int *__restrict__ dst) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= nodes) return;
    const int k0 = xp[i], len = xp[i + 1] - k0;
    const int want = min(len, 32);
    int buf[32];
    int take = 0;
    for (int q = 0; q < want; ++q) {
        const int l = src[xi[k0 + static_cast<int>(static_cast<long>(q) * len / want)]];
        if (l < 0) continue;
        int p = take++;  // insertion sort
        while (p > 0 && buf[p - 1] > l) {
            buf[p] = buf[p - 1];
            --p;
        }
        buf[p] = l;
    }
    if (take == 0) return;
    int best = buf[0], best_n = 0, run = 0;
    for (int q = 0; q < take; ++q) {
        run = (q > 0 && buf[q] == buf[q - 1]) ? run + 1 : 1;
        if (run > best_n) {
            best_n = run;
            best = buf[q];
        }
    }
    dst[i] = best;
}

// key of entry k = (cluster of its row) << 16 | (cluster of its column); entries inside one cluster or with an unlabelled
// end get the all-ones key (a == b == 65535 is such an entry itself, so the marker collides with nothing that counts)
__global__ void __launch_bounds__(kThreads) k_cluster_edge_keys(long nnz, int rows, const int *__restrict__ rp, const int *__restrict__ ci,
                                                               const int *__restrict__ lab_r, const int *__restrict__ lab_c,
                                                               unsigned *__restrict__ key) {
    const long k = static_cast<long>(blockIdx.x) * kThreads + threadIdx.x;
    if (k >= nnz) return;
    int lo = 0, hi = rows;  // rp[lo] <= k < rp[hi]
    while (hi - lo > 1) {
        const int mid = lo + ((hi - lo) >> 1);
        if (rp[mid] <= k) lo = mid;
        else hi = mid;
    }
    const int a = lab_r[lo], b = lab_c[ci[k]];
    key[k] = (a < 0 || b < 0 || a == b) ? 0xffffffffu : (static_cast<unsigned>(a) << 16) | static_cast<unsigned>(b);
}

__global__ void __launch_bounds__(kThreads) k_cluster_pos(int nodes, const int *__restrict__ lab, const double *__restrict__ cpos,
                                                         double *__restrict__ pos) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= nodes) return;
    const int l = lab[i];
    pos[i] = l >= 0 ? cpos[l] : (static_cast<double>(i) + 0.5) / static_cast<double>(nodes);
}

// ---- tiling test

__global__ void __launch_bounds__(kThreads) k_tile_keys(long nnz, int rows, const int *__restrict__ rp, const int *__restrict__ ci,
                                                       const int *__restrict__ r_old2new, const int *__restrict__ c_old2new,
                                                       unsigned long long *__restrict__ key) {
    const long k = static_cast<long>(blockIdx.x) * kThreads + threadIdx.x;
    if (k >= nnz) return;
    int lo = 0, hi = rows;
    while (hi - lo > 1) {
        const int mid = lo + ((hi - lo) >> 1);
        if (rp[mid] <= k) lo = mid;
        else hi = mid;
    }
    const int r = r_old2new ? r_old2new[lo] : lo, c = c_old2new ? c_old2new[ci[k]] : ci[k];
    key[k] = (static_cast<unsigned long long>(r / kTileRows) << 32) | static_cast<unsigned long long>(c / kTileCols);
}

__global__ void __launch_bounds__(kThreads) k_sum_dense_runs(const int *__restrict__ n_runs, const int *__restrict__ counts,
                                                            unsigned long long *__restrict__ total) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    unsigned long long v = (i < *n_runs && counts[i] >= kTileDenseMin) ? static_cast<unsigned long long>(counts[i]) : 0ull;
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(total, v);
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

double device_tiling_dense_fraction(int m, int n, long nnz, const int *rp, const int *ci, const int *row_new2old,
                                    const int *col_new2old, hipStream_t s) {
    if (nnz <= 0) return 0.0;
    DBuf<int> r_old2new, c_old2new;
    if (row_new2old) {
        r_old2new.alloc(static_cast<size_t>(m));
        hipLaunchKernelGGL(k_invert, dim3(grid_for(m)), dim3(kThreads), 0, s, m, row_new2old, r_old2new.p);
    }
    if (col_new2old) {
        c_old2new.alloc(static_cast<size_t>(n));
        hipLaunchKernelGGL(k_invert, dim3(grid_for(n)), dim3(kThreads), 0, s, n, col_new2old, c_old2new.p);
    }
    DBuf<unsigned long long> kin(static_cast<size_t>(nnz)), kout(static_cast<size_t>(nnz));
    hipLaunchKernelGGL(k_tile_keys, dim3(grid_for(nnz)), dim3(kThreads), 0, s, nnz, m, rp, ci, r_old2new.p, c_old2new.p, kin.p);
    int sb_bits = 1;
    while ((1L << sb_bits) < (m + kTileRows - 1) / kTileRows) ++sb_bits;
    {
        size_t bytes = 0;
        HIP_CHECK(hipcub::DeviceRadixSort::SortKeys(nullptr, bytes, kin.p, kout.p, static_cast<int>(nnz), 0, 32 + sb_bits, s));
        DBuf<char> tmp(bytes + 16);
        HIP_CHECK(hipcub::DeviceRadixSort::SortKeys(tmp.p, bytes, kin.p, kout.p, static_cast<int>(nnz), 0, 32 + sb_bits, s));
        HIP_CHECK(hipStreamSynchronize(s));
    }
    DBuf<int> counts(static_cast<size_t>(nnz)), n_runs(1);
    DBuf<unsigned long long> total(1);
    HIP_CHECK(hipMemsetAsync(total.p, 0, sizeof(unsigned long long), s));
    {
        size_t bytes = 0;  // the unique keys land in kin (not needed afterwards)
        HIP_CHECK(hipcub::DeviceRunLengthEncode::Encode(nullptr, bytes, kout.p, kin.p, counts.p, n_runs.p, static_cast<int>(nnz), s));
        DBuf<char> tmp(bytes + 16);
        HIP_CHECK(hipcub::DeviceRunLengthEncode::Encode(tmp.p, bytes, kout.p, kin.p, counts.p, n_runs.p, static_cast<int>(nnz), s));
        HIP_CHECK(hipStreamSynchronize(s));
    }
    hipLaunchKernelGGL(k_sum_dense_runs, dim3(grid_for(nnz)), dim3(kThreads), 0, s, n_runs.p, counts.p, total.p);
    unsigned long long h = 0;
    HIP_CHECK(hipMemcpyAsync(&h, total.p, sizeof(h), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    return static_cast<double>(h) / static_cast<double>(nnz);
}

void device_cluster_positions(int m, int n, long nnz, const int *rp, const int *ci, const int *trp, const int *tci, double *pos_r,
                              double *pos_c, ReorderStats *st, hipStream_t s) {
    ReorderStats local;
    ReorderStats &S = st ? *st : local;
    const long N = static_cast<long>(m) + n;
    long per_cluster = 16384;  // reorder.cpp, cluster_positions: same sizing
    if (const char *e = env_get("HPRLP_REORDER_CLUSTER")) per_cluster = std::max(64L, std::atol(e));
    const int K = static_cast<int>(std::max<long>(2, std::min<long>(65536, std::min<long>(m, N / per_cluster + 1))));
    S.clusters = K;
    DBuf<int> lab_r(static_cast<size_t>(m)), lab_c(static_cast<size_t>(n)), changed(1);
    hipLaunchKernelGGL(k_fill_int, dim3(grid_for(m)), dim3(kThreads), 0, s, static_cast<long>(m), lab_r.p, -1);
    hipLaunchKernelGGL(k_fill_int, dim3(grid_for(n)), dim3(kThreads), 0, s, static_cast<long>(n), lab_c.p, -1);
    hipLaunchKernelGGL(k_seed_rows, dim3(grid_for(K)), dim3(kThreads), 0, s, K, m, lab_r.p);
    // ---- 1. Voronoi clusters: level-synchronous multi-source BFS until nothing changes (a grid of 1e7 nodes needs a few
    // hundred levels; each is two launches over the nodes).  Nodes of components without a seed stay unlabelled.
    const int max_levels = 1 << 16;
    for (int level = 0; level < max_levels; ++level) {
        HIP_CHECK(hipMemsetAsync(changed.p, 0, sizeof(int), s));
        hipLaunchKernelGGL(k_pull_labels, dim3(grid_for(n)), dim3(kThreads), 0, s, n, trp, tci, lab_r.p, lab_c.p, changed.p);
        hipLaunchKernelGGL(k_pull_labels, dim3(grid_for(m)), dim3(kThreads), 0, s, m, rp, ci, lab_c.p, lab_r.p, changed.p);
        int ch = 0;
        HIP_CHECK(hipMemcpyAsync(&ch, changed.p, sizeof(int), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        S.bfs_levels = level + 1;
        if (ch == 0) break;
    }
    // two rounds of majority relabelling (dissolves the satellite blobs of seed rows with a far entry)
    {
        DBuf<int> nr(static_cast<size_t>(m)), nc(static_cast<size_t>(n));
        for (int round = 0; round < 2; ++round) {
            HIP_CHECK(hipMemcpyAsync(nc.p, lab_c.p, sizeof(int) * static_cast<size_t>(n), hipMemcpyDeviceToDevice, s));
            hipLaunchKernelGGL(k_majority, dim3(grid_for(n)), dim3(kThreads), 0, s, n, trp, tci, lab_r.p, nc.p);
            std::swap(lab_c.p, nc.p);
            HIP_CHECK(hipMemcpyAsync(nr.p, lab_r.p, sizeof(int) * static_cast<size_t>(m), hipMemcpyDeviceToDevice, s));
            hipLaunchKernelGGL(k_majority, dim3(grid_for(m)), dim3(kThreads), 0, s, m, rp, ci, lab_c.p, nr.p);
            std::swap(lab_r.p, nr.p);
        }
        HIP_CHECK(hipStreamSynchronize(s));
    }
    // ---- 2. cluster graph: one 32-bit key per entry, sorted, run lengths = directed edge counts
    std::vector<std::vector<std::pair<int, float>>> W(static_cast<size_t>(K));
    {
        DBuf<unsigned> kin(static_cast<size_t>(nnz)), kout(static_cast<size_t>(nnz));
        hipLaunchKernelGGL(k_cluster_edge_keys, dim3(grid_for(nnz)), dim3(kThreads), 0, s, nnz, m, rp, ci, lab_r.p, lab_c.p, kin.p);
        {
            size_t bytes = 0;
            HIP_CHECK(hipcub::DeviceRadixSort::SortKeys(nullptr, bytes, kin.p, kout.p, static_cast<int>(nnz), 0, 32, s));
            DBuf<char> tmp(bytes + 16);
            HIP_CHECK(hipcub::DeviceRadixSort::SortKeys(tmp.p, bytes, kin.p, kout.p, static_cast<int>(nnz), 0, 32, s));
            HIP_CHECK(hipStreamSynchronize(s));
        }
        DBuf<int> counts(static_cast<size_t>(nnz)), n_runs(1);
        {
            size_t bytes = 0;  // unique keys into kin
            HIP_CHECK(hipcub::DeviceRunLengthEncode::Encode(nullptr, bytes, kout.p, kin.p, counts.p, n_runs.p, static_cast<int>(nnz), s));
            DBuf<char> tmp(bytes + 16);
            HIP_CHECK(hipcub::DeviceRunLengthEncode::Encode(tmp.p, bytes, kout.p, kin.p, counts.p, n_runs.p, static_cast<int>(nnz), s));
            HIP_CHECK(hipStreamSynchronize(s));
        }
        int runs = 0;
        n_runs.download(&runs, 1);
        std::vector<unsigned> hk(static_cast<size_t>(runs));
        std::vector<int> hc(static_cast<size_t>(runs));
        if (runs > 0) {
            kin.download(hk.data(), hk.size());
            counts.download(hc.data(), hc.size());
        }
        for (int q = 0; q < runs; ++q) {
            if (hk[q] == 0xffffffffu) continue;
            const int a = static_cast<int>(hk[q] >> 16), b = static_cast<int>(hk[q] & 0xffffu);
            if (a < K && b < K) W[static_cast<size_t>(a)].emplace_back(b, static_cast<float>(hc[q]));
        }
    }
    // ---- 3. spectral order of the clusters (host: K nodes), 4. positions
    const std::vector<double> cpos = cluster_order(K, W, &S);
    DBuf<double> dcpos(static_cast<size_t>(K));
    dcpos.upload(cpos.data(), cpos.size());
    hipLaunchKernelGGL(k_cluster_pos, dim3(grid_for(m)), dim3(kThreads), 0, s, m, lab_r.p, dcpos.p, pos_r);
    hipLaunchKernelGGL(k_cluster_pos, dim3(grid_for(n)), dim3(kThreads), 0, s, n, lab_c.p, dcpos.p, pos_c);
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

// warm-up (abi.cpp: hprlp_warmup): an attribute query makes the runtime load this translation unit's code object now instead
// of at the first launch of one of its kernels
void warm_reorder_tu() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&k_assign_ranks));
}

}  // namespace hprlp
