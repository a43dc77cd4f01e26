// reorder.cpp -- set-up time locality ordering of an LP matrix: the method on the host (what hprlp_locality_ordering and the
// CPU tests run, and the form the device path is checked against) and cluster_order, the one step the solver keeps on the
// host; the solver runs steps 1, 2, 4, 5 on the device (reorder_dev.hip, Solver::try_reorder).
//
// Why: the column-tiled kernels (tiled.h) need column locality -- the rows of a super-block must read a narrow window
// of the gathered vector.  An LP handed over in an arbitrary row/column order (or a structured one whose structure its
// numbering hides) fails the tiling test and falls back to the stream kernel at a quarter of the HBM roofline.  This
// file finds row and column permutations P, Q such that P A Q is band-like where the sparsity graph allows it.  The
// solver then works on the permuted problem (all per-row / per-column vectors permuted alike) and un-permutes the
// solution in collect_solution (reference src/utils.cu:143-200 returns x, y, z in the caller's numbering); the iterates
// are the reference's up to the order of floating-point sums.
//
// Method (bipartite graph: m row nodes + n column nodes, one edge per nonzero; robust against a few percent of
// "far" entries, which defeat plain BFS / Cuthill-McKee orderings because they make the graph a small world):
//   1. Voronoi clusters: K evenly spaced seed rows, multi-source BFS, every node takes the label of the first
//      labelled neighbour (a few hops: clusters are local wherever most edges are local);
//   2. cluster graph: edge counts between clusters; a pair linked only by stray far edges is two orders of magnitude
//      lighter than a true neighbour pair and is dropped (below 15 % of the lighter end's heaviest link);
//   3. spectral ordering of the (small) cluster graph per connected component: the Fiedler vector (second eigenvector of
//      the normalised adjacency) by Lanczos from a BFS-level start vector;
//   4. fine positions: a node starts at its cluster's rank; kReorderSweeps sweeps "column = robust centre (trimmed mean) of
//      its rows' positions, row = robust centre of its columns' positions" -- the trimming ignores the far neighbours --,
//      rank-normalised after each half-sweep;
//   5. the permutations are the argsorts; accepted only if the permuted pattern passes the tiling test that the
//      given order failed.
// Everything is deterministic (fixed seeds, no races: the parallel loops write disjoint outputs from read-only inputs).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <numeric>
#include <thread>
#include <vector>

#include "common.h"
#include "env.h"
#include "reorder.h"
#include "tiled.h"

namespace hprlp {

namespace {

int worker_count(long work) {
    if (work < (1L << 20)) return 1;
    unsigned hw = std::thread::hardware_concurrency();
    if (hw == 0) hw = 1;
    return static_cast<int>(std::min<unsigned>(hw, 16u));
}

// fn(thread index, begin, end) over [0, n) in contiguous chunks
template <class F>
void parallel_chunks(long n, int T, F fn) {
    if (T <= 1 || n < T) {
        fn(0, 0L, n);
        return;
    }
    std::vector<std::thread> th;
    th.reserve(T - 1);
    for (int t = 1; t < T; ++t) th.emplace_back([=]() { fn(t, n * t / T, n * (t + 1) / T); });
    fn(0, 0L, n / T);
    for (auto &x : th) x.join();
}

// pattern transpose (counting sort, stable in row order)
void transpose_pattern(int m, int n, const int *rp, const int *ci, std::vector<int> &trp, std::vector<int> &tci) {
    const long nnz = rp[m];
    trp.assign(static_cast<size_t>(n) + 1, 0);
    for (long k = 0; k < nnz; ++k) ++trp[ci[k] + 1];
    for (int j = 0; j < n; ++j) trp[j + 1] += trp[j];
    tci.resize(static_cast<size_t>(nnz));
    std::vector<int> next(trp.begin(), trp.end() - 1);
    for (int i = 0; i < m; ++i)
        for (int k = rp[i]; k < rp[i + 1]; ++k) tci[next[ci[k]]++] = i;
}

// ranks / count of `pos` (stable: ties by index), in place
void rank_normalise(std::vector<double> &pos, int T) {
    const long n = static_cast<long>(pos.size());
    if (n == 0) return;
    std::vector<int> ord(static_cast<size_t>(n));
    std::iota(ord.begin(), ord.end(), 0);
    std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return pos[a] < pos[b]; });
    const double inv = 1.0 / static_cast<double>(n);
    parallel_chunks(n, T, [&](int, long b, long e) {
        for (long i = b; i < e; ++i) pos[ord[i]] = (static_cast<double>(i) + 0.5) * inv;
    });
}

// out[i] = robust centre of src over the neighbours of i (at most 32 of them, evenly sampled): the mean of the middle 60 % of
// the sorted sample.  The trimming drops the few far neighbours like a median does, the averaging has a third of a
// median's variance on the (roughly uniform) spread of the near neighbours.  Nodes without neighbours keep their value.
// Same rule as k_centre_sweep (reorder_dev.hip).
void centre_sweep(int rows, const int *rp, const int *ci, const std::vector<double> &src, std::vector<double> &out, int T) {
    parallel_chunks(rows, T, [&](int, long b, long e) {
        double buf[32];
        for (long i = b; i < e; ++i) {
            const int k0 = rp[i], len = rp[i + 1] - k0;
            if (len <= 0) continue;
            const int take = std::min(len, 32);
            for (int q = 0; q < take; ++q) buf[q] = src[ci[k0 + static_cast<int>(static_cast<long>(q) * len / take)]];
            std::sort(buf, buf + take);
            static const int trim_env = env_get("HPRLP_REORDER_TRIM") ? std::atoi(env_get("HPRLP_REORDER_TRIM")) : 20;  // percent per side
            int lo = take * trim_env / 100, hi = take - lo;
            if (trim_env >= 50 || hi <= lo) { lo = (take - 1) / 2; hi = take / 2 + 1; }  // median
            double sum = 0.0;
            for (int q = lo; q < hi; ++q) sum += buf[q];
            out[i] = sum / static_cast<double>(hi - lo);
        }
    });
}

}  // namespace

double tiling_dense_fraction(int m, int n, const int *rp, const int *ci, const int *row_new2old, const int *col_old2new) {
    const long nnz = rp[m];
    if (nnz <= 0 || m <= 0) return 0.0;
    const int nsb = (m + kTileRows - 1) / kTileRows, ntile = (n + kTileCols - 1) / kTileCols;
    const int T = worker_count(nnz);
    std::vector<long> dense(static_cast<size_t>(T), 0);
    parallel_chunks(nsb, T, [&](int t, long b, long e) {
        std::vector<int> cnt(static_cast<size_t>(ntile), 0), touched;
        long d = 0;
        for (long sb = b; sb < e; ++sb) {
            touched.clear();
            const int r0 = static_cast<int>(sb) * kTileRows, r1 = std::min(m, r0 + kTileRows);
            for (int i = r0; i < r1; ++i) {
                const int old = row_new2old ? row_new2old[i] : i;
                for (int k = rp[old]; k < rp[old + 1]; ++k) {
                    const int c = col_old2new ? col_old2new[ci[k]] : ci[k];
                    const int tl = c / kTileCols;
                    if (cnt[tl]++ == 0) touched.push_back(tl);
                }
            }
            for (int tl : touched) {
                if (cnt[tl] >= kTileDenseMin) d += cnt[tl];
                cnt[tl] = 0;
            }
        }
        dense[t] += d;
    });
    long d = 0;
    for (long v : dense) d += v;
    return static_cast<double>(d) / static_cast<double>(nnz);
}

// Steps 2b-3: from the directed cluster-to-cluster edge counts W[a] = {(b, count)} (sorted by b) to the position (rank in
// (0, 1)) of every cluster: symmetrise, drop the weak links, order every component of what is left spectrally.
std::vector<double> cluster_order(int K, const std::vector<std::vector<std::pair<int, float>>> &W, ReorderStats *st) {
    ReorderStats local;
    ReorderStats &S = st ? *st : local;
    // symmetric weights S = W + W^T, then the threshold (relative to the heaviest link of either end)
    std::vector<std::vector<std::pair<int, float>>> G(static_cast<size_t>(K));
    {
        std::vector<std::vector<std::pair<int, float>>> WT(static_cast<size_t>(K));
        for (int a = 0; a < K; ++a)
            for (auto &pr : W[a]) WT[pr.first].emplace_back(a, pr.second);
        std::vector<float> heavy(static_cast<size_t>(K), 0.0f);
        for (int a = 0; a < K; ++a) {
            // merge the two sorted lists
            auto &out = G[a];
            size_t i = 0, j = 0;
            const auto &x = W[a], &y = WT[a];
            while (i < x.size() || j < y.size()) {
                if (j >= y.size() || (i < x.size() && x[i].first < y[j].first)) out.push_back(x[i++]);
                else if (i >= x.size() || y[j].first < x[i].first) out.push_back(y[j++]);
                else {
                    out.emplace_back(x[i].first, x[i].second + y[j].second);
                    ++i;
                    ++j;
                }
            }
            for (auto &pr : out) heavy[a] = std::max(heavy[a], pr.second);
        }
        for (int a = 0; a < K; ++a) {
            auto &g = G[a];
            size_t w = 0;
            for (auto &pr : g)
                if (pr.second >= 0.15f * std::min(heavy[a], heavy[pr.first]) && pr.second >= 2.0f) g[w++] = pr;
            g.resize(w);
        }
    }

    // ---- 3. order the clusters: components of the thresholded graph, spectral order inside each
    std::vector<double> cpos(static_cast<size_t>(K), 0.0);  // position (rank) of a cluster
    {
        std::vector<int> comp(static_cast<size_t>(K), -1), order;
        order.reserve(K);
        int ncomp = 0;
        std::vector<int> queue, level(static_cast<size_t>(K), 0);
        for (int s0 = 0; s0 < K; ++s0) {
            if (comp[s0] >= 0) continue;
            // component by BFS
            queue.assign(1, s0);
            comp[s0] = ncomp;
            for (size_t h = 0; h < queue.size(); ++h)
                for (auto &pr : G[queue[h]])
                    if (comp[pr.first] < 0) {
                        comp[pr.first] = ncomp;
                        queue.push_back(pr.first);
                    }
            std::vector<int> nodes = queue;
            const int nc = static_cast<int>(nodes.size());
            if (nc <= 2) {
                for (int v : nodes) order.push_back(v);
                ++ncomp;
                continue;
            }
            // pseudo-peripheral start: the last node of a BFS from the last node of the first BFS
            auto bfs_levels = [&](int root) {
                for (int v : nodes) level[v] = -1;
                queue.assign(1, root);
                level[root] = 0;
                for (size_t h = 0; h < queue.size(); ++h)
                    for (auto &pr : G[queue[h]])
                        if (level[pr.first] < 0) {
                            level[pr.first] = level[queue[h]] + 1;
                            queue.push_back(pr.first);
                        }
                return queue.back();
            };
            const int far1 = bfs_levels(nodes[0]);
            bfs_levels(far1);
            // Second eigenvector of the normalised adjacency N = D^-1/2 S D^-1/2 (the Fiedler vector of the component) by
            // Lanczos with full reorthogonalisation, started from the BFS levels; x = D^-1/2 v orders the clusters.  (A
            // power iteration needs ~(chain length)^2 steps; stopped early it leaves a share of the third eigenvector --
            // a full cosine period, i.e. a FOLD in the order that no local refinement repairs.)
            std::vector<int> loc(static_cast<size_t>(K), -1);
            for (int q = 0; q < nc; ++q) loc[nodes[q]] = q;
            std::vector<double> x(static_cast<size_t>(nc)), dsq(static_cast<size_t>(nc), 0.0);
            for (int q = 0; q < nc; ++q) {
                double d = 0.0;
                for (auto &pr : G[nodes[q]]) d += pr.second;
                dsq[q] = std::sqrt(std::max(d, 1e-300));
                x[q] = static_cast<double>(level[nodes[q]]) * dsq[q];
            }
            std::vector<double> v1(dsq);  // top eigenvector D^1/2 1, normalised
            {
                double nrm = 0.0;
                for (double d : v1) nrm += d * d;
                nrm = std::sqrt(nrm);
                for (double &d : v1) d /= nrm;
            }
            auto apply_N = [&](const std::vector<double> &in, std::vector<double> &out) {
                for (int q = 0; q < nc; ++q) {
                    double sacc = 0.0;
                    for (auto &pr : G[nodes[q]]) {
                        const int r = loc[pr.first];
                        sacc += pr.second * in[r] / dsq[r];
                    }
                    out[q] = sacc / dsq[q];
                }
            };
            auto dot = [&](const std::vector<double> &a2, const std::vector<double> &b2) {
                double sacc = 0.0;
                for (int q = 0; q < nc; ++q) sacc += a2[q] * b2[q];
                return sacc;
            };
            const int max_steps = std::min(nc - 1, 400);
            std::vector<std::vector<double>> Q;
            std::vector<double> alpha, beta, w(static_cast<size_t>(nc));
            {
                const double c1 = dot(x, v1);
                for (int q = 0; q < nc; ++q) x[q] -= c1 * v1[q];
                const double nrm = std::sqrt(dot(x, x));
                if (nrm < 1e-300) {  // degenerate start (all levels equal): any vector orthogonal to v1
                    for (int q = 0; q < nc; ++q) x[q] = (q % 2 ? 1.0 : -1.0);
                    const double c2 = dot(x, v1);
                    for (int q = 0; q < nc; ++q) x[q] -= c2 * v1[q];
                }
                const double n2 = std::sqrt(dot(x, x));
                for (double &d : x) d /= n2;
            }
            Q.push_back(x);
            std::vector<double> ritz;  // coefficients of the wanted Ritz vector in the Lanczos basis
            int it = 0;
            for (; it < max_steps; ++it) {
                apply_N(Q[it], w);
                const double al = dot(w, Q[it]);
                alpha.push_back(al);
                // full reorthogonalisation (twice) against v1 and every Lanczos vector so far
                for (int pass = 0; pass < 2; ++pass) {
                    const double c1 = dot(w, v1);
                    for (int q = 0; q < nc; ++q) w[q] -= c1 * v1[q];
                    for (const auto &qv : Q) {
                        const double cq = dot(w, qv);
                        for (int q = 0; q < nc; ++q) w[q] -= cq * qv[q];
                    }
                }
                const double be = std::sqrt(dot(w, w));
                const int j = static_cast<int>(alpha.size());
                if ((j % 20 == 0) || be < 1e-12 || it + 1 == max_steps) {
                    // largest eigenpair of the tridiagonal T_j: bisection on the Sturm count, then inverse iteration
                    double lo = -2.0, hi = 2.0;
                    auto count_below = [&](double mu) {  // eigenvalues of T_j smaller than mu
                        int cnt = 0;
                        double d = 1.0;
                        for (int i = 0; i < j; ++i) {
                            d = alpha[i] - mu - (i > 0 ? beta[i - 1] * beta[i - 1] / d : 0.0);
                            if (std::abs(d) < 1e-300) d = -1e-300;
                            if (d < 0) ++cnt;
                        }
                        return cnt;
                    };
                    for (int bis = 0; bis < 100; ++bis) {
                        const double mid = 0.5 * (lo + hi);
                        if (count_below(mid) >= j) hi = mid; else lo = mid;  // all j below mid: the top one is below mid
                    }
                    const double theta = 0.5 * (lo + hi);
                    std::vector<double> sv(static_cast<size_t>(j), 1.0), dd(static_cast<size_t>(j)), rhs(static_cast<size_t>(j));
                    for (int rep = 0; rep < 3; ++rep) {  // (T - (theta + eps)) sv_new = sv  by the Thomas algorithm
                        const double mu = theta + 1e-10 * std::max(1.0, std::abs(theta));
                        rhs = sv;
                        dd[0] = alpha[0] - mu;
                        for (int i = 1; i < j; ++i) {
                            if (std::abs(dd[i - 1]) < 1e-300) dd[i - 1] = 1e-300;
                            const double f = beta[i - 1] / dd[i - 1];
                            dd[i] = alpha[i] - mu - f * beta[i - 1];
                            rhs[i] -= f * rhs[i - 1];
                        }
                        if (std::abs(dd[j - 1]) < 1e-300) dd[j - 1] = 1e-300;
                        sv[j - 1] = rhs[j - 1] / dd[j - 1];
                        for (int i = j - 2; i >= 0; --i) sv[i] = (rhs[i] - beta[i] * sv[i + 1]) / dd[i];
                        double nrm = 0.0;
                        for (double d : sv) nrm += d * d;
                        nrm = std::sqrt(std::max(nrm, 1e-300));
                        for (double &d : sv) d /= nrm;
                    }
                    ritz = sv;
                    const double resid = std::abs(be * sv[j - 1]);  // |N y - theta y| of the Ritz pair
                    if (resid < 1e-6 || be < 1e-12) {
                        ++it;
                        break;
                    }
                }
                beta.push_back(be);
                for (double &d : w) d /= be;
                Q.push_back(w);
            }
            std::fill(x.begin(), x.end(), 0.0);
            for (size_t i = 0; i < ritz.size(); ++i)
                for (int q = 0; q < nc; ++q) x[q] += ritz[i] * Q[i][q];
            for (int q = 0; q < nc; ++q) x[q] /= dsq[q];
            std::vector<int> ord(static_cast<size_t>(nc));
            S.spectral_iterations += it;
            std::iota(ord.begin(), ord.end(), 0);
            std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return x[a] < x[b]; });
            for (int q : ord) order.push_back(nodes[q]);
            ++ncomp;
        }
        S.components = ncomp;
        for (int q = 0; q < K; ++q) cpos[order[q]] = (static_cast<double>(q) + 0.5) / K;
    }

    return cpos;
}

void cluster_positions(int m, int n, const int *rp, const int *ci, const int *trp_, const int *tci_, std::vector<double> *pos_r,
                       std::vector<double> *pos_c, ReorderStats *st) {
    ReorderStats local;
    ReorderStats &S = st ? *st : local;
    auto tphase = time_now();
    const bool timing = env_get("HPRLP_TIMING") != nullptr;
    auto tick = [&](const char *what) {
        if (timing) std::fprintf(stderr, "[timing]   reorder %-28s %.2f s\n", what, time_since(tphase));
        tphase = time_now();
    };
    const long nnz = rp[m];
    const int T = worker_count(nnz);
    const int *trp = trp_, *tci = tci_;
    // ---- 1. Voronoi clusters
    const long N = static_cast<long>(m) + n;
    // about 16k nodes per cluster: a BFS ball of that size is as wide as the matrix' natural window anyway, and the
    // spectral ordering of the cluster graph (dense-ish: every cluster links to all that overlap it) stays cheap
    long per_cluster = 16384;
    if (const char *e = env_get("HPRLP_REORDER_CLUSTER")) per_cluster = std::max(64L, std::atol(e));
    const int K = static_cast<int>(std::max<long>(2, std::min<long>(65536, std::min<long>(m, N / per_cluster + 1))));
    S.clusters = K;
    std::vector<int> lab_r(static_cast<size_t>(m), -1), lab_c(static_cast<size_t>(n), -1);
    for (int k = 0; k < K; ++k) lab_r[static_cast<size_t>(static_cast<long>(k) * m / K)] = k;
    for (int level = 0; level < 12; ++level) {
        // columns pull from rows, then rows pull from the columns just labelled (one hop each)
        std::vector<long> changed(static_cast<size_t>(T), 0);
        parallel_chunks(n, T, [&](int t, long b, long e) {
            long ch = 0;
            for (long j = b; j < e; ++j) {
                if (lab_c[j] >= 0) continue;
                for (int k = trp[j]; k < trp[j + 1]; ++k) {
                    const int l = lab_r[tci[k]];
                    if (l >= 0) {
                        lab_c[j] = l;
                        ++ch;
                        break;
                    }
                }
            }
            changed[t] = ch;
        });
        // (a row reads column labels only and writes its own: no races, the result does not depend on the thread count)
        parallel_chunks(m, T, [&](int t, long b, long e) {
            long ch = 0;
            for (long i = b; i < e; ++i) {
                if (lab_r[i] >= 0) continue;
                for (int k = rp[i]; k < rp[i + 1]; ++k) {
                    const int l = lab_c[ci[k]];
                    if (l >= 0) {
                        lab_r[i] = l;
                        ++ch;
                        break;
                    }
                }
            }
            changed[t] += ch;
        });
        long ch = 0;
        for (long v : changed) ch += v;
        S.bfs_levels = level + 1;
        if (ch == 0) break;
    }
    // Graphs of large diameter (grids, chains: a BFS ball grows polynomially, not by a factor per hop) are not covered after
    // those levels: finish with a queue-based multi-source BFS from the labelled nodes, O(edges) in all, sequential and in
    // index order (deterministic).  Nodes of components without a seed stay unlabelled (they keep their own relative index).
    {
        std::vector<int> queue;  // node ids: rows 0..m-1, columns m..m+n-1
        for (int i = 0; i < m; ++i)
            if (lab_r[i] >= 0) queue.push_back(i);
        for (int j = 0; j < n; ++j)
            if (lab_c[j] >= 0) queue.push_back(m + j);
        if (static_cast<long>(queue.size()) < N) {
            for (size_t h = 0; h < queue.size(); ++h) {
                const int v = queue[h];
                if (v < m) {
                    const int lab = lab_r[v];
                    for (int k = rp[v]; k < rp[v + 1]; ++k)
                        if (lab_c[ci[k]] < 0) {
                            lab_c[ci[k]] = lab;
                            queue.push_back(m + ci[k]);
                        }
                } else {
                    const int j = v - m, lab = lab_c[j];
                    for (int k = trp[j]; k < trp[j + 1]; ++k)
                        if (lab_r[tci[k]] < 0) {
                            lab_r[tci[k]] = lab;
                            queue.push_back(tci[k]);
                        }
                }
            }
        }
    }

    // A seed row with a far entry grows a satellite blob around that entry's column (a few percent of the cluster, far
    // away); such a cluster links two distant places of the cluster graph and folds the spectral order.  Two rounds of
    // "take the label most of your neighbours have" dissolve the satellites into the clusters around them.
    {
        auto majority = [&](int cnt_nodes, const int *xp, const int *xi, const std::vector<int> &src, std::vector<int> &dst) {
            parallel_chunks(cnt_nodes, T, [&](int, long b, long e) {
                int buf[32];
                for (long i = b; i < e; ++i) {
                    const int k0 = xp[i], len = xp[i + 1] - k0;
                    int take = 0;
                    const int want = std::min(len, 32);
                    for (int q = 0; q < want; ++q) {
                        const int l = src[xi[k0 + static_cast<int>(static_cast<long>(q) * len / want)]];
                        if (l >= 0) buf[take++] = l;
                    }
                    if (take == 0) continue;
                    std::sort(buf, buf + take);
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
            });
        };
        for (int round = 0; round < 2; ++round) {
            std::vector<int> nc_lab(lab_c), nr_lab(lab_r);
            majority(n, trp, tci, lab_r, nc_lab);
            lab_c.swap(nc_lab);
            majority(m, rp, ci, lab_c, nr_lab);
            lab_r.swap(nr_lab);
        }
    }
    tick("Voronoi clusters");
    // ---- 2. cluster graph from the row side: W[a] = {(b, #edges between rows of a and columns of b)}
    std::vector<int> cl_ptr(static_cast<size_t>(K) + 1, 0), cl_rows;
    {
        for (int i = 0; i < m; ++i)
            if (lab_r[i] >= 0) ++cl_ptr[lab_r[i] + 1];
        for (int k = 0; k < K; ++k) cl_ptr[k + 1] += cl_ptr[k];
        cl_rows.resize(static_cast<size_t>(cl_ptr[K]));
        std::vector<int> next(cl_ptr.begin(), cl_ptr.end() - 1);
        for (int i = 0; i < m; ++i)
            if (lab_r[i] >= 0) cl_rows[next[lab_r[i]]++] = i;
    }
    std::vector<std::vector<std::pair<int, float>>> W(static_cast<size_t>(K));
    parallel_chunks(K, T, [&](int, long b, long e) {
        std::vector<int> cnt(static_cast<size_t>(K), 0), touched;
        for (long a = b; a < e; ++a) {
            touched.clear();
            for (int q = cl_ptr[a]; q < cl_ptr[a + 1]; ++q) {
                const int i = cl_rows[q];
                for (int k = rp[i]; k < rp[i + 1]; ++k) {
                    const int l = lab_c[ci[k]];
                    if (l < 0 || l == a) continue;
                    if (cnt[l]++ == 0) touched.push_back(l);
                }
            }
            std::sort(touched.begin(), touched.end());
            W[a].reserve(touched.size());
            for (int l : touched) {
                W[a].emplace_back(l, static_cast<float>(cnt[l]));
                cnt[l] = 0;
            }
        }
    });
    tick("cluster graph");
    const std::vector<double> cpos = cluster_order(K, W, &S);
    tick("spectral order of the clusters");
    // ---- 4. fine positions
    std::vector<double> pr_(static_cast<size_t>(m)), pc_(static_cast<size_t>(n));
    for (int i = 0; i < m; ++i) pr_[i] = lab_r[i] >= 0 ? cpos[lab_r[i]] : (static_cast<double>(i) + 0.5) / m;
    for (int j = 0; j < n; ++j) pc_[j] = lab_c[j] >= 0 ? cpos[lab_c[j]] : (static_cast<double>(j) + 0.5) / n;
    *pos_r = std::move(pr_);
    *pos_c = std::move(pc_);
}

bool locality_ordering(int m, int n, const int *rp, const int *ci, std::vector<int> *row_new2old, std::vector<int> *col_new2old,
                       ReorderStats *st, double accept_fraction) {
    ReorderStats local;
    ReorderStats &S = st ? *st : local;
    S = ReorderStats();
    const auto t0 = time_now();
    auto tphase = time_now();
    const bool timing = env_get("HPRLP_TIMING") != nullptr;
    auto tick = [&](const char *what) {
        if (timing) std::fprintf(stderr, "[timing]   reorder %-28s %.2f s\n", what, time_since(tphase));
        tphase = time_now();
    };
    const long nnz = (m > 0) ? rp[m] : 0;
    if (m <= 0 || n <= 0 || nnz <= 0) return false;
    const int T = worker_count(nnz);
    S.fraction_before = tiling_dense_fraction(m, n, rp, ci, nullptr, nullptr);
    if (S.fraction_before >= accept_fraction) {
        S.seconds = time_since(t0);
        return false;  // the given order is already local
    }
    tick("tiling test (given order)");
    std::vector<int> trp, tci;
    transpose_pattern(m, n, rp, ci, trp, tci);
    tick("pattern transpose");
    std::vector<double> pr_, pc_;
    cluster_positions(m, n, rp, ci, trp.data(), tci.data(), &pr_, &pc_, &S);
    tphase = time_now();
    rank_normalise(pr_, T);
    const int nsweeps = env_get("HPRLP_REORDER_SWEEPS") ? std::atoi(env_get("HPRLP_REORDER_SWEEPS")) : kReorderSweeps;
    for (int sweep = 0; sweep < nsweeps; ++sweep) {
        centre_sweep(n, trp.data(), tci.data(), pr_, pc_, T);
        rank_normalise(pc_, T);
        centre_sweep(m, rp, ci, pc_, pr_, T);
        rank_normalise(pr_, T);
    }

    tick("median sweeps");
    // ---- 5. permutations, accepted only if they make the pattern tileable
    row_new2old->resize(static_cast<size_t>(m));
    col_new2old->resize(static_cast<size_t>(n));
    std::iota(row_new2old->begin(), row_new2old->end(), 0);
    std::iota(col_new2old->begin(), col_new2old->end(), 0);
    std::stable_sort(row_new2old->begin(), row_new2old->end(), [&](int a, int b) { return pr_[a] < pr_[b]; });
    std::stable_sort(col_new2old->begin(), col_new2old->end(), [&](int a, int b) { return pc_[a] < pc_[b]; });
    std::vector<int> col_old2new(static_cast<size_t>(n));
    for (int j = 0; j < n; ++j) col_old2new[(*col_new2old)[j]] = j;
    S.fraction_after = tiling_dense_fraction(m, n, rp, ci, row_new2old->data(), col_old2new.data());
    tick("sorts + tiling test (permuted)");
    S.seconds = time_since(t0);
    if (S.fraction_after < accept_fraction || S.fraction_after <= S.fraction_before) {
        row_new2old->clear();
        col_new2old->clear();
        return false;
    }
    return true;
}

}  // namespace hprlp
