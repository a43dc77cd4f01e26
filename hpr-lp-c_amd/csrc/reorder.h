// reorder.h -- set-up time locality ordering of an LP matrix (reorder.cpp; private).
#pragma once

#include <hip/hip_runtime.h>

#include <utility>
#include <vector>

namespace hprlp {

constexpr int kReorderSweeps = 3;  // refinement sweeps of the fine positions (host and device paths)

struct ReorderStats {
    double fraction_before = 0.0;  // share of the entries in tiles dense enough to be staged (the tiling test), given order
    double fraction_after = 0.0;   // same, permuted
    int clusters = 0, components = 0, bfs_levels = 0, spectral_iterations = 0;
    double seconds = 0.0;
};

// Share of the nonzeros that land in (super-block, tile) pairs with at least kTileDenseMin entries -- what the tiled
// builders accept -- for the pattern as given (null permutations) or permuted (row_new2old[m], col_old2new[n]).
double tiling_dense_fraction(int m, int n, const int *rp, const int *ci, const int *row_new2old, const int *col_old2new);

// Finds permutations that make the pattern of the m x n CSR matrix band-like.  Returns false (permutations empty) when
// the given order already passes the tiling test at accept_fraction, or when no ordering that passes it was found.
bool locality_ordering(int m, int n, const int *rp, const int *ci, std::vector<int> *row_new2old, std::vector<int> *col_new2old,
                       ReorderStats *stats, double accept_fraction = 0.5);

// Steps 1-3 of the method (reorder.cpp): Voronoi clusters, cluster graph, spectral order; pos_r / pos_c = the rank of a
// node's cluster in (0, 1) (nodes no cluster reached keep their own relative index).  trp / tci: pattern of A^T.
void cluster_positions(int m, int n, const int *rp, const int *ci, const int *trp, const int *tci, std::vector<double> *pos_r,
                       std::vector<double> *pos_c, ReorderStats *stats);

// Steps 2b-3 alone: directed cluster-to-cluster edge counts W[a] = {(b, count)}, sorted by b -> rank of every cluster in (0, 1).
std::vector<double> cluster_order(int K, const std::vector<std::vector<std::pair<int, float>>> &W, ReorderStats *stats);

// cluster_positions with the graph work on the device (level-synchronous Voronoi BFS, majority relabelling, edge counts by
// radix sort + run lengths); only the K x K cluster graph comes back for cluster_order.  pos_r / pos_c: device arrays.
void device_cluster_positions(int m, int n, long nnz, const int *rp, const int *ci, const int *trp, const int *tci, double *pos_r,
                              double *pos_c, ReorderStats *stats, hipStream_t s);
// the tiling test (tiling_dense_fraction) for device patterns; null permutations = the given order
double device_tiling_dense_fraction(int m, int n, long nnz, const int *rp, const int *ci, const int *row_new2old,
                                    const int *col_new2old, hipStream_t s);

// Device parts (reorder_dev.hip; all pointers device memory): the median sweeps + rank normalisation + argsorts, and P A Q.
void device_refine_order(int m, int n, const int *rp, const int *ci, const int *trp, const int *tci, double *pos_r, double *pos_c,
                         int sweeps, int *row_new2old, int *col_new2old, hipStream_t s);
void device_permute_csr(int m, int n, long nnz, const int *rp, const int *ci, const double *val, const int *row_new2old,
                        const int *col_new2old, int *rp_out, int *ci_out, double *val_out, hipStream_t s);

}  // namespace hprlp
