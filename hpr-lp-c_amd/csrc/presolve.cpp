// presolve.cpp -- in-process LP presolve / postsolve (design and scope: presolve.h).
#include "presolve.h"
#include "env.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <string>
#include <thread>

#include "HPRLP.h"
#include "common.h"

namespace hprlp {
namespace {

constexpr double kFeasTol = 1e-9;  // a crossing of bounds beyond this (relative) is left to the solver
constexpr int kMaxPasses = 50;
constexpr double kSlackPivot = 0.5;  // a costed slack column is substituted only if |a_ij| >= this * max_k |a_ik|
constexpr size_t kMaxParallelProbe = 64;  // partners tried per row / column inside a group of equal sparsity pattern

inline bool fin(double v) { return std::isfinite(v); }
inline double rel(double v) { return kFeasTol * (1.0 + std::abs(v)); }
// per-index term of the order-independent pattern hash of the parallel-row / parallel-column scans
inline unsigned long long pattern_hash(int idx) {
    unsigned long long z = static_cast<unsigned long long>(idx) + 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

}  // namespace

ReduceStage::~ReduceStage() {
    if (reduced_) free_model(reduced_);
}

// Large models (more than kScanFirstNnz nonzeros): before paying for the transpose and the working copies, one
// threaded scan counts what the first pass of the loop below would remove (fixed and empty columns, empty,
// singleton and redundant rows).  Less than 0.1 % of the rows + columns: not worth a reduced copy of the model.
constexpr long kScanFirstNnz = 10000000;

bool ReduceStage::worth_it(const LP_info_cpu *model) const {
    const int m = model->m, n = model->n;
    const int *rp = model->A->rowPtr, *ci = model->A->colIndex;
    const double *av = model->A->value, *l = model->l, *u = model->u, *AL = model->AL, *AU = model->AU;
    const int T = static_cast<int>(std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency())));
    const int chunk = (m + T - 1) / T;
    std::vector<std::vector<int>> hist(static_cast<size_t>(T));
    std::vector<long> rows_hit(static_cast<size_t>(T), 0);
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
        th.emplace_back([&, t]() {
            std::vector<int> &h = hist[t];
            h.assign(static_cast<size_t>(n), 0);
            long hit = 0;
            const int r0 = std::min(m, t * chunk), r1 = std::min(m, r0 + chunk);
            for (int i = r0; i < r1; ++i) {
                int cnt = 0;
                double lo_act = 0.0, up_act = 0.0;
                bool lo_inf = false, up_inf = false;
                for (int k = rp[i]; k < rp[i + 1]; ++k) {
                    const double a = av[k];
                    if (a == 0.0) continue;
                    const int j = ci[k];
                    ++cnt;
                    ++h[j];
                    const double bl = a > 0 ? l[j] : u[j], bu = a > 0 ? u[j] : l[j];
                    if (fin(bl)) lo_act += a * bl;
                    else lo_inf = true;
                    if (fin(bu)) up_act += a * bu;
                    else up_inf = true;
                }
                const bool redundant = (!fin(AL[i]) || (!lo_inf && lo_act >= AL[i])) && (!fin(AU[i]) || (!up_inf && up_act <= AU[i]));
                if (cnt <= 1 || redundant) ++hit;
            }
            rows_hit[t] = hit;
        });
    for (auto &x : th) x.join();
    long hit = 0;
    for (long v : rows_hit) hit += v;
    for (int j = 0; j < n; ++j) {
        int c = 0;
        for (int t = 0; t < T; ++t) c += hist[t][j];
        if (c == 0 || (fin(l[j]) && l[j] == u[j])) ++hit;
    }
    return static_cast<double>(hit) >= 1e-3 * (static_cast<double>(m) + static_cast<double>(n));
}

bool ReduceStage::run(const LP_info_cpu *model) {
    const auto t0 = std::chrono::steady_clock::now();
    if (!model || !model->A || model->m <= 0 || model->n <= 0) return false;
    org_ = model;
    m_ = model->m;
    n_ = model->n;
    const int m = m_, n = n_;
    const int *rp = model->A->rowPtr, *ci = model->A->colIndex;
    const double *av = model->A->value;
    const long nnz = rp[m];
    if (nnz > kScanFirstNnz && !worth_it(model)) {
        stats_.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return false;
    }
    csr_transpose_host(m, n, nnz, rp, ci, av, trp_, tci_, tv_);

    std::vector<double> AL(model->AL, model->AL + m), AU(model->AU, model->AU + m);
    std::vector<double> l(model->l, model->l + n), u(model->u, model->u + n);
    std::vector<double> cost(model->c, model->c + n);  // slack substitution moves cost between columns
    std::vector<char> row_alive(m, 1), col_alive(n, 1);
    std::vector<int> row_cnt(m), col_cnt(n);
    for (int i = 0; i < m; ++i) row_cnt[i] = rp[i + 1] - rp[i];
    for (int j = 0; j < n; ++j) col_cnt[j] = trp_[j + 1] - trp_[j];
    // explicit zeros do not count as structure
    for (int i = 0; i < m; ++i)
        for (int k = rp[i]; k < rp[i + 1]; ++k)
            if (av[k] == 0.0) {
                --row_cnt[i];
                --col_cnt[ci[k]];
            }
    double offset = 0.0;
    bool give_up = false;
    const bool timing = env_get("HPRLP_TIMING") != nullptr;  // section times of this stage on stderr
    double t_sec[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto t_mark = std::chrono::steady_clock::now();
    auto lap = [&](int k) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        t_sec[k] += std::chrono::duration<double>(now - t_mark).count();
        t_mark = now;
    };
    lap(0);  // transpose + working copies
    // HPRLP_PRESOLVE_OFF=slack,dualfix,parallel,forcing switches reductions off (diagnostics)
    const char *off_env = env_get("HPRLP_PRESOLVE_OFF");
    const std::string off = off_env ? off_env : "";
    const bool use_slack = off.find("slack") == std::string::npos, use_dualfix = off.find("dualfix") == std::string::npos;
    const bool use_parallel = off.find("parallel") == std::string::npos, use_forcing = off.find("forcing") == std::string::npos;
    const char *sp_env = env_get("HPRLP_SLACK_PIVOT");
    const double slack_pivot = sp_env ? std::atof(sp_env) : kSlackPivot;

    auto fix_column = [&](int j, double v, Kind kind) {
        for (int k = trp_[j]; k < trp_[j + 1]; ++k) {
            const int i = tci_[k];
            if (!row_alive[i] || tv_[k] == 0.0) continue;
            const double s = tv_[k] * v;
            if (fin(AL[i])) AL[i] -= s;
            if (fin(AU[i])) AU[i] -= s;
            --row_cnt[i];
        }
        offset += cost[j] * v;
        col_alive[j] = 0;
        stack_.push_back(Record{kind, -1, j, 0.0, v, l[j], u[j], v, v, cost[j]});
        if (kind == FixedCol) ++stats_.fixed_cols;
        else if (kind == DualFixCol) ++stats_.dual_fixed_cols;
        else ++stats_.empty_cols;
    };
    auto drop_row = [&](int i, Kind kind) {
        for (int k = rp[i]; k < rp[i + 1]; ++k)
            if (col_alive[ci[k]] && av[k] != 0.0) --col_cnt[ci[k]];
        row_alive[i] = 0;
        if (kind != SingletonRow) stack_.push_back(Record{kind, i, -1, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0});
    };

    bool changed = true;
    while (changed && !give_up && stats_.passes < kMaxPasses) {
        changed = false;
        ++stats_.passes;
        // ---- columns whose bounds coincide
        for (int j = 0; j < n && !give_up; ++j) {
            if (!col_alive[j]) continue;
            if (l[j] > u[j]) {
                if (l[j] - u[j] > rel(l[j])) give_up = true;  // infeasible bounds: the solver reports it
                else l[j] = u[j] = 0.5 * (l[j] + u[j]);
            }
            if (!give_up && fin(l[j]) && l[j] == u[j]) {
                fix_column(j, l[j], FixedCol);
                changed = true;
            }
        }
        lap(1);  // fixed columns
        // ---- rows
        for (int i = 0; i < m && !give_up; ++i) {
            if (!row_alive[i]) continue;
            if (row_cnt[i] == 0) {
                if (AL[i] > rel(AL[i]) || AU[i] < -rel(AU[i])) {
                    give_up = true;
                    break;
                }
                drop_row(i, EmptyRow);
                ++stats_.empty_rows;
                changed = true;
                continue;
            }
            if (row_cnt[i] == 1) {
                int j = -1;
                double a = 0.0;
                for (int k = rp[i]; k < rp[i + 1]; ++k)
                    if (col_alive[ci[k]] && av[k] != 0.0) {
                        j = ci[k];
                        a = av[k];
                        break;
                    }
                double lo = a > 0 ? AL[i] / a : AU[i] / a;
                double up = a > 0 ? AU[i] / a : AL[i] / a;
                if (std::isnan(lo) || std::isnan(up)) continue;  // 0 * inf style input: leave the row alone
                const double l_new = std::max(l[j], lo), u_new = std::min(u[j], up);
                if (l_new > u_new && l_new - u_new > rel(l_new)) {
                    give_up = true;
                    break;
                }
                stack_.push_back(Record{SingletonRow, i, j, a, 0.0, l[j], u[j], l_new, u_new});
                l[j] = l_new;
                u[j] = std::max(u_new, l_new);
                stack_.back().u_new = u[j];
                drop_row(i, SingletonRow);
                ++stats_.singleton_rows;
                changed = true;
                continue;
            }
            // activity bounds: a row that no point of the box can violate carries no information
            if (!fin(AL[i]) && !fin(AU[i])) {
                drop_row(i, RedundantRow);
                ++stats_.redundant_rows;
                changed = true;
                continue;
            }
            double lo_act = 0.0, up_act = 0.0;
            bool lo_inf = false, up_inf = false;
            for (int k = rp[i]; k < rp[i + 1] && !(lo_inf && up_inf); ++k) {
                const int j = ci[k];
                const double a = av[k];
                if (!col_alive[j] || a == 0.0) continue;
                const double bl = a > 0 ? l[j] : u[j], bu = a > 0 ? u[j] : l[j];
                if (fin(bl)) lo_act += a * bl;
                else lo_inf = true;
                if (fin(bu)) up_act += a * bu;
                else up_inf = true;
            }
            const bool low_ok = !fin(AL[i]) || (!lo_inf && lo_act >= AL[i]);
            const bool upp_ok = !fin(AU[i]) || (!up_inf && up_act <= AU[i]);
            if (low_ok && upp_ok) {
                drop_row(i, RedundantRow);
                ++stats_.redundant_rows;
                changed = true;
                continue;
            }
            // forcing row: the smallest activity the box allows already sits on the upper side (or the largest on the
            // lower side) -- every column of the row is pinned to the bound that realises it
            const bool force_min = fin(AU[i]) && !lo_inf && std::abs(lo_act - AU[i]) <= rel(AU[i]);
            const bool force_max = fin(AL[i]) && !up_inf && std::abs(up_act - AL[i]) <= rel(AL[i]);
            if (use_forcing && (force_min || force_max)) {
                std::vector<int> live;
                double amin = INFINITY, amax = 0.0;
                for (int k = rp[i]; k < rp[i + 1]; ++k)
                    if (col_alive[ci[k]] && av[k] != 0.0) {
                        live.push_back(ci[k]);
                        amin = std::min(amin, std::abs(av[k]));
                        amax = std::max(amax, std::abs(av[k]));
                    }
                // the test above holds to a tolerance: a column whose whole range moves the activity by less than that
                // tolerance is not pinned by the row at all -- rows with such entries are left to the solver
                if (amin < 1e-6 * amax) continue;
                std::sort(live.begin(), live.end());
                if (std::adjacent_find(live.begin(), live.end()) != live.end()) continue;  // repeated column index: leave it
                const int cnt = static_cast<int>(live.size());
                drop_row(i, SingletonRow);  // (no record of its own from drop_row)
                stack_.push_back(Record{ForcingRow, i, cnt, force_min ? -1.0 : 1.0, 0.0, 0.0, 0.0, 0.0, 0.0});
                for (int k = rp[i]; k < rp[i + 1]; ++k) {
                    const int j = ci[k];
                    if (!col_alive[j] || av[k] == 0.0) continue;
                    const bool at_lower = (av[k] > 0) == force_min;
                    fix_column(j, at_lower ? l[j] : u[j], FixedCol);
                }
                ++stats_.forcing_rows;
                changed = true;
            }
        }
        lap(2);  // rows: empty / singleton / redundant / forcing
        // ---- parallel rows (PSLP: Parallel_rows): row i2 = lambda * row i1 over the live columns.  Row i2 goes, row i1
        // keeps the intersection of its own sides and row i2's sides divided by lambda.  (The two parallel scans hash
        // every live row and column: they run in the first pass and then in every fourth one.)
        const bool scan_parallel = use_parallel && stats_.passes % 4 == 1;
        if (scan_parallel) {
            // rows grouped by an order-independent hash of their live column pattern
            std::vector<std::pair<unsigned long long, int>> keys;
            std::vector<std::pair<int, double>> e1, e2;
            auto live_entries = [&](int i, std::vector<std::pair<int, double>> &out) {
                out.clear();
                for (int k = rp[i]; k < rp[i + 1]; ++k)
                    if (col_alive[ci[k]] && av[k] != 0.0) out.emplace_back(ci[k], av[k]);
                std::sort(out.begin(), out.end());
                for (size_t q = 1; q < out.size(); ++q)
                    if (out[q].first == out[q - 1].first) return false;  // repeated column index: leave the row alone
                return true;
            };
            for (int i = 0; i < m; ++i) {
                if (!row_alive[i] || row_cnt[i] < 2) continue;
                unsigned long long h = static_cast<unsigned long long>(row_cnt[i]);
                for (int k = rp[i]; k < rp[i + 1]; ++k)
                    if (col_alive[ci[k]] && av[k] != 0.0) h += pattern_hash(ci[k]);
                keys.emplace_back(h, i);
            }
            std::sort(keys.begin(), keys.end());
            std::vector<int> rows;
            for (size_t g0 = 0; g0 < keys.size() && !give_up;) {
                size_t g1 = g0 + 1;
                while (g1 < keys.size() && keys[g1].first == keys[g0].first) ++g1;
                rows.clear();
                for (size_t q = g0; q < g1; ++q) rows.push_back(keys[q].second);  // ascending row numbers
                g0 = g1;
                if (rows.size() < 2) continue;
                for (size_t p1 = 0; p1 < rows.size() && !give_up; ++p1) {
                    const int i1 = rows[p1];
                    if (!row_alive[i1] || !live_entries(i1, e1)) continue;
                    for (size_t p2 = p1 + 1; p2 < rows.size() && p2 - p1 <= kMaxParallelProbe && !give_up; ++p2) {
                        const int i2 = rows[p2];
                        if (!row_alive[i2] || !live_entries(i2, e2) || e2.size() != e1.size()) continue;
                        const double lambda = e2[0].second / e1[0].second;
                        bool par = std::isfinite(lambda) && lambda != 0.0;
                        for (size_t q = 0; q < e1.size() && par; ++q)
                            par = e1[q].first == e2[q].first &&
                                  std::abs(e2[q].second - lambda * e1[q].second) <= 1e-12 * std::abs(e2[q].second);
                        if (!par) continue;
                        const double lo2 = lambda > 0 ? AL[i2] / lambda : AU[i2] / lambda;
                        const double up2 = lambda > 0 ? AU[i2] / lambda : AL[i2] / lambda;
                        const double lo = std::max(AL[i1], lo2), up = std::min(AU[i1], up2);
                        if (lo > up && lo - up > rel(lo)) {
                            give_up = true;  // the two rows contradict each other: the solver reports it
                            break;
                        }
                        stack_.push_back(Record{ParallelRow, i2, i1, lambda, 0.0, AL[i1], AU[i1], lo2, up2});
                        AL[i1] = lo;
                        AU[i1] = std::max(up, lo);
                        drop_row(i2, SingletonRow);  // (SingletonRow: drop_row pushes no record of its own)
                        ++stats_.parallel_rows;
                        changed = true;
                    }
                }
            }
        }
        // ---- parallel columns (PSLP: Parallel_cols): column j2 = lambda * column j1 over the live rows and c_j2 =
        // lambda * c_j1.  The pair acts through x_j1 + lambda x_j2 only: column j2 goes, column j1 stands for the sum
        // with the sum's range as its bounds.
        if (scan_parallel) {
            std::vector<std::pair<unsigned long long, int>> keys;
            std::vector<std::pair<int, double>> e1, e2;
            auto live_entries = [&](int j, std::vector<std::pair<int, double>> &out) {
                out.clear();
                for (int k = trp_[j]; k < trp_[j + 1]; ++k)
                    if (row_alive[tci_[k]] && tv_[k] != 0.0) out.emplace_back(tci_[k], tv_[k]);
                std::sort(out.begin(), out.end());
                for (size_t q = 1; q < out.size(); ++q)
                    if (out[q].first == out[q - 1].first) return false;
                return true;
            };
            for (int j = 0; j < n; ++j) {
                if (!col_alive[j] || col_cnt[j] < 2) continue;
                unsigned long long h = static_cast<unsigned long long>(col_cnt[j]);
                for (int k = trp_[j]; k < trp_[j + 1]; ++k)
                    if (row_alive[tci_[k]] && tv_[k] != 0.0) h += pattern_hash(tci_[k]);
                keys.emplace_back(h, j);
            }
            std::sort(keys.begin(), keys.end());
            std::vector<int> cols;
            for (size_t g0 = 0; g0 < keys.size();) {
                size_t g1 = g0 + 1;
                while (g1 < keys.size() && keys[g1].first == keys[g0].first) ++g1;
                cols.clear();
                for (size_t q = g0; q < g1; ++q) cols.push_back(keys[q].second);
                g0 = g1;
                if (cols.size() < 2) continue;
                for (size_t p1 = 0; p1 < cols.size(); ++p1) {
                    const int j1 = cols[p1];
                    if (!col_alive[j1] || !live_entries(j1, e1)) continue;
                    for (size_t p2 = p1 + 1; p2 < cols.size() && p2 - p1 <= kMaxParallelProbe; ++p2) {
                        const int j2 = cols[p2];
                        if (!col_alive[j2] || !live_entries(j2, e2) || e2.size() != e1.size()) continue;
                        const double lambda = e2[0].second / e1[0].second;
                        bool par = std::isfinite(lambda) && lambda != 0.0 &&
                                   std::abs(cost[j2] - lambda * cost[j1]) <= 1e-12 * std::abs(cost[j2]);
                        for (size_t q = 0; q < e1.size() && par; ++q)
                            par = e1[q].first == e2[q].first &&
                                  std::abs(e2[q].second - lambda * e1[q].second) <= 1e-12 * std::abs(e2[q].second);
                        if (!par) continue;
                        const double t_lo = std::min(lambda * l[j2], lambda * u[j2]), t_up = std::max(lambda * l[j2], lambda * u[j2]);
                        if (std::isnan(t_lo) || std::isnan(t_up)) continue;
                        stack_.push_back(Record{ParallelCol, j2, j1, lambda, 0.0, l[j1], u[j1], l[j2], u[j2]});
                        l[j1] += t_lo;  // (-inf stays -inf; the lower parts are never +inf)
                        u[j1] += t_up;
                        for (const auto &e : e2) --row_cnt[e.first];
                        col_alive[j2] = 0;
                        ++stats_.parallel_cols;
                        changed = true;
                    }
                }
            }
        }
        lap(3);  // parallel rows and columns
        // ---- slack columns (PSLP: StonCols): column j appears only in row i,
        //   AL <= a x_j + sum_k a_ik x_k <= AU,  l_j <= x_j <= u_j,
        // and either the row is an equality (any cost) or c_j = 0.  x_j is eliminated: the row becomes
        //   AL - a u_j <= sum_k a_ik x_k <= AU - a l_j   (a > 0; bounds swapped for a < 0)
        // and, in the equality case, c_j x_j = c_j (b - sum_k a_ik x_k) / a moves onto the other columns' costs.
        for (int j = 0; j < n && !give_up && use_slack; ++j) {
            if (!col_alive[j] || col_cnt[j] != 1) continue;
            int i = -1;
            double a = 0.0;
            for (int k = trp_[j]; k < trp_[j + 1]; ++k)
                if (row_alive[tci_[k]] && tv_[k] != 0.0) {
                    i = tci_[k];
                    a = tv_[k];
                    break;
                }
            if (i < 0 || row_cnt[i] < 2) continue;
            bool equality = fin(AL[i]) && AL[i] == AU[i];
            const double cj = cost[j];
            const double ratio = cj / a;
            if (!fin(ratio)) continue;
            double row_max = 0.0, mn = 0.0, mx = 0.0;  // largest entry of the row; activity range of its OTHER columns
            bool mn_inf = false, mx_inf = false;
            for (int k = rp[i]; k < rp[i + 1]; ++k) {
                const int c2 = ci[k];
                if (!col_alive[c2] || av[k] == 0.0) continue;
                row_max = std::max(row_max, std::abs(av[k]));
                if (c2 == j) continue;
                const double bl = av[k] > 0 ? l[c2] : u[c2], bu = av[k] > 0 ? u[c2] : l[c2];
                if (fin(bl)) mn += av[k] * bl; else mn_inf = true;
                if (fin(bu)) mx += av[k] * bu; else mx_inf = true;
            }
            // the cost moves onto the row's other columns multiplied by a_ik / a: only behind a pivot that is not small
            // for its row, or |c| of the reduced model (and with it the meaning of the relative tolerance) blows up
            if (cj != 0.0 && std::abs(a) < slack_pivot * row_max) continue;
            // implied-free column: the row and the other columns' bounds already keep x_j inside [l_j, u_j] -- it
            // counts as free (the bounds stay in the record; they only guard rounding at postsolve)
            double lj = l[j], uj = u[j];
            if (fin(lj) || fin(uj)) {
                const double t_lo = (fin(AL[i]) && !mx_inf) ? AL[i] - mx : -INFINITY, t_up = (fin(AU[i]) && !mn_inf) ? AU[i] - mn : INFINITY;
                const double x_lo = a > 0 ? t_lo / a : t_up / a, x_up = a > 0 ? t_up / a : t_lo / a;
                if ((!fin(lj) || x_lo >= lj - rel(lj)) && (!fin(uj) || x_up <= uj + rel(uj))) {
                    lj = -INFINITY;
                    uj = INFINITY;
                }
            }
            if (!equality && cj != 0.0) {
                // a FREE column with a cost: its reduced cost must vanish, so y_i = c_j / a is known and not zero -- the
                // row is active on the matching side in every optimal solution and may be treated as that equality
                if (fin(lj) || fin(uj)) continue;
                const double side = (ratio > 0.0) ? AL[i] : AU[i];
                if (!fin(side)) continue;  // unbounded direction: the solver sees it
                AL[i] = AU[i] = side;
                equality = true;
            }
            const double lo = a > 0 ? AL[i] - a * uj : AL[i] - a * lj;  // -inf when that bound of x_j is infinite
            const double up = a > 0 ? AU[i] - a * lj : AU[i] - a * uj;
            if (std::isnan(lo) || std::isnan(up)) continue;
            if (cj != 0.0) {
                for (int k = rp[i]; k < rp[i + 1]; ++k)
                    if (col_alive[ci[k]] && ci[k] != j && av[k] != 0.0) cost[ci[k]] -= ratio * av[k];
                offset += ratio * AL[i];
            }
            stack_.push_back(Record{SlackCol, i, j, a, 0.0, l[j], u[j], AL[i], AU[i], cj});  // l_new / u_new: the row's sides
            AL[i] = lo;
            AU[i] = up;
            --row_cnt[i];
            col_alive[j] = 0;
            ++stats_.slack_cols;
            changed = true;
        }
        lap(4);  // slack columns
        // ---- dual fixing (PSLP: Simple_dual_fix): a column whose cost and whose rows all push it the same way sits at
        // that bound in some optimal solution.  Down: c_j >= 0 and lowering x_j can violate no row (positive entries
        // only in rows without a lower side, negative entries only in rows without an upper side); up: mirrored.
        for (int j = 0; j < n && !give_up && use_dualfix; ++j) {
            if (!col_alive[j] || col_cnt[j] == 0) continue;
            const double c = cost[j];
            bool down_ok = c >= 0.0 && fin(l[j]), up_ok = c <= 0.0 && fin(u[j]);
            for (int k = trp_[j]; k < trp_[j + 1] && (down_ok || up_ok); ++k) {
                const int i = tci_[k];
                const double a = tv_[k];
                if (!row_alive[i] || a == 0.0) continue;
                const bool has_lo = fin(AL[i]), has_up = fin(AU[i]);
                if (a > 0 ? has_lo : has_up) down_ok = false;  // lowering x_j lowers (a > 0) / raises (a < 0) the activity
                if (a > 0 ? has_up : has_lo) up_ok = false;
            }
            if (down_ok || up_ok) {
                fix_column(j, down_ok ? l[j] : u[j], DualFixCol);
                changed = true;
            }
        }
        lap(5);  // dual fixing
        // ---- columns that no row uses any more
        for (int j = 0; j < n && !give_up; ++j) {
            if (!col_alive[j] || col_cnt[j] != 0) continue;
            const double c = cost[j];
            double v;
            if (c > 0) v = l[j];
            else if (c < 0) v = u[j];
            else v = fin(l[j]) ? l[j] : (fin(u[j]) ? u[j] : 0.0);
            if (!fin(v)) continue;  // unbounded direction: keep the column, the solver sees it
            fix_column(j, v, EmptyCol);
            changed = true;
        }
    }
    lap(6);  // empty columns (last pass)
    if (give_up) return false;

    // ---- assemble the reduced model
    std::vector<int> new_col(n, -1);
    for (int j = 0; j < n; ++j)
        if (col_alive[j]) {
            new_col[j] = static_cast<int>(col_of_.size());
            col_of_.push_back(j);
        }
    for (int i = 0; i < m; ++i)
        if (row_alive[i]) row_of_.push_back(i);
    const int rm = static_cast<int>(row_of_.size()), rn = static_cast<int>(col_of_.size());
    stats_.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (rn == 0) {  // every column was removed (and with them every row): the records alone give the optimum
        solved_ = rm == 0;
        return false;
    }
    if (rm == 0) return false;                  // columns without rows that no rule could place: the solver sees them
    if (rm == m && rn == n) return false;       // nothing removed
    std::vector<int> rrp(rm + 1, 0), rci;
    std::vector<double> rv, rAL(rm), rAU(rm), rl(rn), ru(rn), rc(rn);
    rci.reserve(nnz);
    rv.reserve(nnz);
    for (int r = 0; r < rm; ++r) {
        const int i = row_of_[r];
        for (int k = rp[i]; k < rp[i + 1]; ++k)
            if (col_alive[ci[k]]) {  // explicit zeros of live columns are kept: same sparsity pattern
                rci.push_back(new_col[ci[k]]);
                rv.push_back(av[k]);
            }
        rrp[r + 1] = static_cast<int>(rci.size());
        rAL[r] = AL[i];
        rAU[r] = AU[i];
    }
    for (int q = 0; q < rn; ++q) {
        rl[q] = l[col_of_[q]];
        ru[q] = u[col_of_[q]];
        rc[q] = cost[col_of_[q]];
    }
    reduced_ = model_from_csr(rm, rn, static_cast<long>(rci.size()), rrp.data(), rci.data(), rv.data(), rAL.data(),
                              rAU.data(), rl.data(), ru.data(), rc.data(), model->obj_constant + offset);
    stats_.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    lap(7);  // assembling the reduced model
    if (timing)
        std::fprintf(stderr, "[timing] presolve reductions (%d passes): set-up %.4f, fixed cols %.4f, rows %.4f, parallel %.4f, slack %.4f, dual fix %.4f, empty cols %.4f, assemble %.4f s\n",
                     stats_.passes, t_sec[0], t_sec[1], t_sec[2], t_sec[3], t_sec[4], t_sec[5], t_sec[6], t_sec[7]);
    return reduced_ != nullptr;
}

void ReduceStage::postsolve(const double *xr, const double *yr, const double *zr, double *x, double *y, double *z) const {
    std::fill(x, x + n_, 0.0);
    std::fill(y, y + m_, 0.0);
    std::fill(z, z + n_, 0.0);
    for (size_t q = 0; q < col_of_.size(); ++q) {
        x[col_of_[q]] = xr[q];
        z[col_of_[q]] = zr[q];
    }
    for (size_t r = 0; r < row_of_.size(); ++r) y[row_of_[r]] = yr[r];
    std::vector<char> have_x(static_cast<size_t>(n_), 0);  // columns whose x is final at this point of the undo sequence
    for (size_t q = 0; q < col_of_.size(); ++q) have_x[col_of_[q]] = 1;
    const int *rp = org_->A->rowPtr, *ci = org_->A->colIndex;
    const double *av = org_->A->value;
    // Undo the reductions last-in first-out.  At every point (x, y, z) is a primal-dual pair of the problem as it was
    // when the record on top was applied: rows not yet restored carry y = 0, a column's reduced cost is formed with its
    // cost at that time, whenever a row dual is set the reduced cost of its column is updated, and undoing a slack
    // substitution shifts the row's dual by what it had moved into the other columns' costs (their reduced costs do
    // not change).  So z = c - A^T y holds for the original model at the end.
    for (size_t s = stack_.size(); s-- > 0;) {
        const Record &r = stack_[s];
        switch (r.kind) {
            case FixedCol:
            case DualFixCol:
            case EmptyCol: {
                double red = r.cost;
                for (int k = trp_[r.j]; k < trp_[r.j + 1]; ++k) red -= tv_[k] * y[tci_[k]];
                x[r.j] = r.v;
                z[r.j] = red;
                have_x[r.j] = 1;
                break;
            }
            case SlackCol: {
                // r.l_new / r.u_new = the row's sides when the column was eliminated; the columns of the row that were
                // alive at that time are exactly the ones restored so far.  a x_j has to land in
                // [AL - act, AU - act] and in a * [l_j, u_j]; an active side of the reduced row (y != 0) pins it to the
                // matching end, which is also the bound of x_j that the reduced cost -a y then leans on.
                double act = 0.0;
                for (int k = rp[r.i]; k < rp[r.i + 1]; ++k)
                    if (ci[k] != r.j && have_x[ci[k]]) act += av[k] * x[ci[k]];
                const double b1 = r.a * r.l_old, b2 = r.a * r.u_old;
                const double tlo = std::max(r.l_new - act, std::min(b1, b2)), thi = std::min(r.u_new - act, std::max(b1, b2));
                double t;
                if (y[r.i] > 0.0) t = tlo;
                else if (y[r.i] < 0.0) t = thi;
                else t = fin(tlo) ? tlo : (fin(thi) ? thi : 0.0);
                double xj = t / r.a;
                xj = std::min(std::max(xj, r.l_old), r.u_old);  // rounding only
                x[r.j] = xj;
                z[r.j] = -r.a * y[r.i];    // which side of the ranged row is active = which bound x_j sits on
                y[r.i] += r.cost / r.a;    // the cost that moved onto the other columns belongs to the row's multiplier
                have_x[r.j] = 1;
                break;
            }
            case SingletonRow: {
                // the row became the bound l_new / u_new of column j: if that bound is the one the
                // reduced cost leans on and the row (not the old bound) supplied it, the multiplier
                // belongs to the row
                const double zj = z[r.j];
                double yi = 0.0;
                if (zj > 0.0 && r.l_new > r.l_old) yi = zj / r.a;
                else if (zj < 0.0 && r.u_new < r.u_old) yi = zj / r.a;
                if (yi != 0.0) {
                    y[r.i] = yi;
                    z[r.j] = 0.0;
                }
                break;
            }
            case ParallelCol: {
                // x[r.j] is the sum x_j1 + lambda x_j2 (r.i = j2, r.a = lambda; r.l_old / r.u_old: bounds of j1, r.l_new /
                // r.u_new: bounds of j2): split it inside both boxes.  At a bound of the sum both parts sit at their own
                // bounds, so the reduced costs z_j1 = z, z_j2 = lambda z keep their meaning.
                const double xm = x[r.j], zm = z[r.j];
                const double t_lo = std::min(r.a * r.l_new, r.a * r.u_new), t_up = std::max(r.a * r.l_new, r.a * r.u_new);
                const double t_pref = std::min(std::max(0.0, t_lo), t_up);
                double x1 = std::min(std::max(xm - t_pref, r.l_old), r.u_old);
                const double t2 = std::min(std::max(xm - x1, t_lo), t_up);
                x1 = xm - t2;
                x[r.j] = std::min(std::max(x1, r.l_old), r.u_old);  // rounding only
                x[r.i] = std::min(std::max(t2 / r.a, r.l_new), r.u_new);
                z[r.i] = r.a * zm;
                have_x[r.i] = 1;
                break;
            }
            case ParallelRow: {
                // row r.i = r.a * row r.j was folded into row r.j: the multiplier belongs to whichever of the two rows
                // supplied the active side (r.l_old / r.u_old: row r.j's own sides, r.l_new / r.u_new: row r.i's sides
                // divided by r.a); a_{r.j} y = a_{r.i} (y / r.a), so no reduced cost changes
                const double yk = y[r.j];
                const bool from_i = (yk > 0.0 && r.l_new > r.l_old) || (yk < 0.0 && r.u_new < r.u_old);
                if (from_i) {
                    y[r.i] = yk / r.a;
                    y[r.j] = 0.0;
                } else {
                    y[r.i] = 0.0;
                }
                break;
            }
            case ForcingRow: {
                // the r.j records above this one (already undone) are the columns the row pinned; their reduced costs were
                // formed with y_i = 0.  r.a = -1: row on its upper side, columns at the bound of least activity -- the
                // multiplier is the largest y <= 0 that leaves every one of them the sign its bound needs; r.a = +1: mirrored.
                double yi = 0.0;
                for (int q = 1; q <= r.j; ++q) {
                    const Record &c = stack_[s + q];
                    double a = 0.0;
                    for (int k = trp_[c.j]; k < trp_[c.j + 1]; ++k)
                        if (tci_[k] == r.i) a += tv_[k];
                    if (a == 0.0) continue;
                    const double cand = z[c.j] / a;
                    yi = r.a < 0.0 ? std::min(yi, cand) : std::max(yi, cand);
                }
                y[r.i] = yi;
                if (yi != 0.0)
                    for (int q = 1; q <= r.j; ++q) {
                        const Record &c = stack_[s + q];
                        for (int k = trp_[c.j]; k < trp_[c.j + 1]; ++k)
                            if (tci_[k] == r.i) z[c.j] -= tv_[k] * yi;
                    }
                break;
            }
            case EmptyRow:
            case RedundantRow:
                y[r.i] = 0.0;
                break;
        }
    }
}

OriginalKkt original_kkt(const LP_info_cpu *model, const double *x, const double *y, const double *z) {
    // same definitions as the reference's check on the original model (src/pslp_integration.cpp:458-580):
    // duals are first projected onto the sign their finite bounds allow
    const int m = model->m, n = model->n;
    const int *rp = model->A->rowPtr, *ci = model->A->colIndex;
    const double *av = model->A->value;
    auto project = [](double v, double lo, double hi) {
        const bool lo_inf = std::isinf(lo) && lo < 0, hi_inf = std::isinf(hi) && hi > 0;
        if (lo_inf && hi_inf) return 0.0;
        if (hi_inf) return std::max(v, 0.0);
        if (lo_inf) return std::min(v, 0.0);
        return v;
    };
    std::vector<double> aty(n, 0.0);
    double rhs2 = 0.0, row_viol2 = 0.0, d_lin = 0.0;
    for (int i = 0; i < m; ++i) {
        const double yp = project(y[i], model->AL[i], model->AU[i]);
        double ax = 0.0;
        for (int k = rp[i]; k < rp[i + 1]; ++k) {
            ax += av[k] * x[ci[k]];
            aty[ci[k]] += av[k] * yp;
        }
        const double lo = fin(model->AL[i]) ? std::abs(model->AL[i]) : 0.0, hi = fin(model->AU[i]) ? std::abs(model->AU[i]) : 0.0;
        rhs2 += std::max(lo, hi) * std::max(lo, hi);
        double viol = 0.0;
        if (fin(model->AL[i]) && ax < model->AL[i]) viol = std::max(viol, model->AL[i] - ax);
        if (fin(model->AU[i]) && ax > model->AU[i]) viol = std::max(viol, ax - model->AU[i]);
        row_viol2 += viol * viol;
        const double support = yp >= 0.0 ? (fin(model->AL[i]) ? model->AL[i] : 0.0) : (fin(model->AU[i]) ? model->AU[i] : 0.0);
        d_lin += yp * support;
    }
    double c2 = 0.0, box_viol2 = 0.0, dual2 = 0.0, p_lin = 0.0;
    for (int j = 0; j < n; ++j) {
        const double zp = project(z[j], model->l[j], model->u[j]);
        c2 += model->c[j] * model->c[j];
        double viol = 0.0;
        if (fin(model->l[j]) && x[j] < model->l[j]) viol = std::max(viol, model->l[j] - x[j]);
        if (fin(model->u[j]) && x[j] > model->u[j]) viol = std::max(viol, x[j] - model->u[j]);
        box_viol2 += viol * viol;
        const double res = model->c[j] - aty[j] - zp;
        dual2 += res * res;
        p_lin += model->c[j] * x[j];
        const double support = zp >= 0.0 ? (fin(model->l[j]) ? model->l[j] : 0.0) : (fin(model->u[j]) ? model->u[j] : 0.0);
        d_lin += zp * support;
    }
    OriginalKkt k;
    k.primal_feas = std::max(std::sqrt(row_viol2), std::sqrt(box_viol2)) / (1.0 + std::sqrt(rhs2));
    k.dual_feas = std::sqrt(dual2) / (1.0 + std::sqrt(c2));
    k.gap = std::abs(d_lin - p_lin) / (1.0 + std::abs(d_lin) + std::abs(p_lin));
    k.primal_obj = p_lin + model->obj_constant;
    k.dual_obj = d_lin + model->obj_constant;
    return k;
}

}  // namespace hprlp
