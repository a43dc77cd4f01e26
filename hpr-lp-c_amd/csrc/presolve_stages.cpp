// presolve_stages.cpp -- the two presolve stages that change the matrix or the box (doubleton equations, primal bound
// propagation) and the chain that solve() runs (presolve.h).  Host code; the counterpart in the reference is the vendored
// PSLP presolver (third_party/PSLP/src/explorers/DtonsEq.c, Primal_propagation.c), run there in a forked child
// (src/pslp_integration.cpp:219-339); written from the mathematics of the reductions, with our own data structures.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <string>

#include "HPRLP.h"
#include "env.h"
#include "common.h"
#include "presolve.h"

namespace hprlp {
namespace {

constexpr double kFeasTol = 1e-9;        // as in presolve.cpp: a crossing of bounds beyond this (relative) is left to the solver
constexpr double kMaxPivotRatio = 10.0;  // |a_k / a_j| of a doubleton: the factor a substitution multiplies with (PSLP allows 1e3)
constexpr int kMaxSubstColumn = 256;     // longest column that is substituted (fill-in and postsolve storage stay small)
constexpr double kCancel = 1e-12;        // a merged coefficient this small relative to its parts counts as cancelled
constexpr double kGray = 1e-6;           // ... and one between kCancel and this makes the substitution too ill-conditioned to do
constexpr double kTinyEntry = 1e-6;      // a substitution must not leave an entry below this times the largest other entry of its row
constexpr double kBoundMargin = 1e-6;    // an implied bound is loosened by this (relative to 1 + |bound|): it stays redundant
constexpr double kHugeBound = 1e8;       // implied bounds beyond this are not worth having
constexpr int kBoundSweeps = 3;
constexpr long kStageMaxNnz = 50000000;  // beyond this the dynamic row lists / the model copy cost more than they can save
constexpr int kMaxRounds = 4;

inline bool fin(double v) { return std::isfinite(v); }
inline double rel(double v) { return kFeasTol * (1.0 + std::abs(v)); }

using Entry = std::pair<int, double>;  // (column, value), rows sorted by column

// position of column c in a sorted row, or -1
inline int find_col(const std::vector<Entry> &row, int c) {
    auto it = std::lower_bound(row.begin(), row.end(), c, [](const Entry &e, int col) { return e.first < col; });
    return (it != row.end() && it->first == c) ? static_cast<int>(it - row.begin()) : -1;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// doubleton equations
// ------------------------------------------------------------------------------------------------
DoubletonStage::~DoubletonStage() {
    if (reduced_) free_model(reduced_);
}

bool DoubletonStage::run(const LP_info_cpu *model) {
    if (!model || !model->A || model->m <= 0 || model->n <= 0) return false;
    m_ = model->m;
    n_ = model->n;
    const int m = m_, n = n_;
    const int *rp = model->A->rowPtr, *ci = model->A->colIndex;
    const double *av = model->A->value;
    const long nnz = rp[m];
    if (nnz > kStageMaxNnz) return false;
    // cheap test first: is there any equality row with two entries at all?
    std::vector<int> queue;
    for (int i = 0; i < m; ++i) {
        if (!(fin(model->AL[i]) && model->AL[i] == model->AU[i])) continue;
        int cnt = 0;
        for (int k = rp[i]; k < rp[i + 1] && cnt <= 2; ++k) cnt += av[k] != 0.0;
        if (cnt == 2) queue.push_back(i);
    }
    if (queue.empty()) return false;

    // working copy: rows as sorted (column, value) lists without explicit zeros (duplicates summed), column -> rows lists
    // (a column's list may name a row that no longer holds it: every use looks the entry up)
    std::vector<std::vector<Entry>> R(static_cast<size_t>(m));
    std::vector<std::vector<int>> C(static_cast<size_t>(n));
    std::vector<int> col_cnt(static_cast<size_t>(n), 0);
    for (int i = 0; i < m; ++i) {
        auto &row = R[i];
        row.reserve(static_cast<size_t>(rp[i + 1] - rp[i]));
        for (int k = rp[i]; k < rp[i + 1]; ++k)
            if (av[k] != 0.0) row.emplace_back(ci[k], av[k]);
        std::sort(row.begin(), row.end(), [](const Entry &a, const Entry &b) { return a.first < b.first; });
        size_t w = 0;
        for (size_t q = 0; q < row.size(); ++q) {
            if (w > 0 && row[w - 1].first == row[q].first) row[w - 1].second += row[q].second;
            else row[w++] = row[q];
        }
        row.resize(w);
        for (const Entry &e : row) {
            C[e.first].push_back(i);
            ++col_cnt[e.first];
        }
    }
    std::vector<double> AL(model->AL, model->AL + m), AU(model->AU, model->AU + m);
    std::vector<double> l(model->l, model->l + n), u(model->u, model->u + n), cost(model->c, model->c + n);
    std::vector<char> row_alive(static_cast<size_t>(m), 1), col_alive(static_cast<size_t>(n), 1);
    double offset = 0.0;

    const int max_elim = env_get("HPRLP_DTON_MAX") ? std::atoi(env_get("HPRLP_DTON_MAX")) : 1 << 30;  // (debugging)
    for (size_t h = 0; h < queue.size() && static_cast<int>(recs_.size()) < max_elim; ++h) {
        const int i = queue[h];
        if (!row_alive[i] || R[i].size() != 2 || !(fin(AL[i]) && AL[i] == AU[i])) continue;
        const Entry e0 = R[i][0], e1 = R[i][1];
        if (!col_alive[e0.first] || !col_alive[e1.first]) continue;
        // Substitute the shorter column (less fill-in) unless its coefficient is the small one of the two: x_j = (b - a_k x_k) / a_j
        // multiplies everything it touches by a_k / a_j, and chains of doubletons multiply those factors (three links at 500
        // each left a reduced model that was infeasible at 1e-7) -- so |a_k / a_j| stays below kMaxPivotRatio, by taking the
        // other column if need be.
        bool subst0 = col_cnt[e0.first] <= col_cnt[e1.first];
        auto ratio_of = [&](bool s0) { return s0 ? std::abs(e1.second / e0.second) : std::abs(e0.second / e1.second); };
        auto len_of = [&](bool s0) { return col_cnt[s0 ? e0.first : e1.first]; };
        if (!(ratio_of(subst0) <= kMaxPivotRatio) || len_of(subst0) > kMaxSubstColumn) subst0 = !subst0;
        if (!(ratio_of(subst0) <= kMaxPivotRatio) || len_of(subst0) > kMaxSubstColumn) continue;
        const int j = subst0 ? e0.first : e1.first, k = subst0 ? e1.first : e0.first;
        const double aj = subst0 ? e0.second : e1.second, ak = subst0 ? e1.second : e0.second;
        const double b = AL[i];
        // bounds of x_j as bounds of x_k:  a_k x_k = b - a_j x_j  in  [b - max(a_j l_j, a_j u_j), b - min(...)]
        const double t1 = aj * l[j], t2 = aj * u[j];
        const double s_lo = b - std::max(t1, t2), s_up = b - std::min(t1, t2);  // (no NaN: a_j != 0, b finite)
        const double k_lo = ak > 0 ? s_lo / ak : s_up / ak, k_up = ak > 0 ? s_up / ak : s_lo / ak;
        const double l_new = std::max(l[k], k_lo), u_new = std::min(u[k], k_up);
        if (l_new > u_new && l_new - u_new > rel(l_new)) return false;  // the two boxes contradict the row: the solver reports it
        Rec rec{i, j, k, aj, ak, b, l[k], u[k], l_new, std::max(u_new, l_new), cost[j], static_cast<int>(ents_.size()), 0};
        if (env_get("HPRLP_DTON_TRACE"))
            std::fprintf(stderr, "[dton %zu] row %d: %.17g x%d + %.17g x%d = %.17g; x%d in [%g, %g] (len %d), x%d in [%g, %g] (len %d) -> [%.17g, %.17g]\n",
                         recs_.size() + 1, i, aj, j, ak, k, b, j, l[j], u[j], col_cnt[j], k, l[k], u[k], col_cnt[k], rec.lk_new, rec.uk_new);
        // A merged coefficient that nearly cancels (|a_rk - (a_rj / a_j) a_k| tiny against its parts, but not rounding noise)
        // would stay in the model as an entry of size 1e-10 whose column other reductions then take at face value (a forcing
        // row pinned such a column to a bound 11 away from its value): leave this doubleton to the solver.
        // The same for an entry that is tiny against the rest of its row without any cancellation: a chain of substitutions
        // with |a_k / a_j| well below one multiplies those factors into the fill-in (1e-3 * 0.1^7 = 1e-10 seen), the column
        // scaling of the solver then blows such a column up, and the reduced model took 294 300 iterations at 1e-4 where the
        // model as given took 1 500 (tests/test_presolve.py, seed 45; oracle on both).  kTinyEntry bounds the dynamic range a
        // substitution may add to a row.
        {
            bool gray = false;
            for (int r : C[j]) {
                if (r == i || !row_alive[r]) continue;
                const auto &row = R[r];
                const int pj = find_col(row, j);
                if (pj < 0) continue;
                const int pk = find_col(row, k);
                const double old = pk >= 0 ? row[pk].second : 0.0, delta = -(row[pj].second / aj) * ak, now = old + delta;
                const double scale = std::max(std::abs(old), std::abs(delta));
                if (pk >= 0 && std::abs(now) > kCancel * scale && std::abs(now) < kGray * scale) {
                    gray = true;
                    break;
                }
                if (pk < 0 || std::abs(now) > kCancel * scale) {  // the entry stays: how small is it in its row?
                    double row_max = 0.0;
                    for (const Entry &e : row)
                        if (e.first != j && e.first != k) row_max = std::max(row_max, std::abs(e.second));
                    if (std::abs(now) < kTinyEntry * row_max) {
                        gray = true;
                        break;
                    }
                }
            }
            if (gray) continue;
        }
        // every other row that holds x_j:  a_rj x_j = (a_rj / a_j) (b - a_k x_k)
        for (int r : C[j]) {
            if (r == i || !row_alive[r]) continue;
            auto &row = R[r];
            const int pj = find_col(row, j);
            if (pj < 0) continue;  // stale
            const double arj = row[pj].second;
            ents_.emplace_back(r, arj);
            row.erase(row.begin() + pj);
            const double f = arj / aj, delta = -f * ak;
            const int pk = find_col(row, k);
            if (pk >= 0) {
                const double old = row[pk].second, now = old + delta;
                if (std::abs(now) <= kCancel * std::max(std::abs(old), std::abs(delta))) {
                    row.erase(row.begin() + pk);
                    --col_cnt[k];
                } else {
                    row[pk].second = now;
                }
            } else {
                row.insert(std::lower_bound(row.begin(), row.end(), k, [](const Entry &e, int col) { return e.first < col; }),
                           Entry(k, delta));
                C[k].push_back(r);
                ++col_cnt[k];
            }
            if (fin(AL[r])) AL[r] -= f * b;
            if (fin(AU[r])) AU[r] -= f * b;
            if (row.size() == 2 && fin(AL[r]) && AL[r] == AU[r]) queue.push_back(r);
        }
        rec.e1 = static_cast<int>(ents_.size());
        cost[k] -= cost[j] * ak / aj;
        offset += cost[j] * b / aj;
        l[k] = rec.lk_new;
        u[k] = rec.uk_new;
        row_alive[i] = 0;
        col_alive[j] = 0;
        col_cnt[j] = 0;
        --col_cnt[k];
        R[i].clear();
        recs_.push_back(rec);
    }
    if (recs_.empty()) return false;

    std::vector<int> new_col(static_cast<size_t>(n), -1);
    for (int j = 0; j < n; ++j)
        if (col_alive[j]) {
            new_col[j] = static_cast<int>(col_of_.size());
            col_of_.push_back(j);
        }
    for (int i = 0; i < m; ++i)
        if (row_alive[i]) row_of_.push_back(i);
    const int rm = static_cast<int>(row_of_.size()), rn = static_cast<int>(col_of_.size());
    if (rm == 0 || rn == 0) {  // (a chain of doubletons can use up every row: leave such models to the other stage)
        recs_.clear();
        ents_.clear();
        row_of_.clear();
        col_of_.clear();
        return false;
    }
    std::vector<int> rrp(static_cast<size_t>(rm) + 1, 0), rci;
    std::vector<double> rv, rAL(static_cast<size_t>(rm)), rAU(static_cast<size_t>(rm)), rl(static_cast<size_t>(rn)),
        ru(static_cast<size_t>(rn)), rc(static_cast<size_t>(rn));
    for (int r = 0; r < rm; ++r) {
        const int i = row_of_[r];
        for (const Entry &e : R[i]) {
            rci.push_back(new_col[e.first]);
            rv.push_back(e.second);
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
    reduced_ = model_from_csr(rm, rn, static_cast<long>(rci.size()), rrp.data(), rci.data(), rv.data(), rAL.data(), rAU.data(),
                              rl.data(), ru.data(), rc.data(), model->obj_constant + offset);
    return reduced_ != nullptr;
}

void DoubletonStage::postsolve(const double *xr, const double *yr, const double *zr, double *x, double *y, double *z) const {
    std::fill(x, x + n_, 0.0);
    std::fill(y, y + m_, 0.0);
    std::fill(z, z + n_, 0.0);
    for (size_t q = 0; q < col_of_.size(); ++q) {
        x[col_of_[q]] = xr[q];
        z[col_of_[q]] = zr[q];
    }
    for (size_t r = 0; r < row_of_.size(); ++r) y[row_of_[r]] = yr[r];
    // Last substitution first.  With rho_c = c_c - sum over the OTHER rows of a_rc y_r (costs and entries as they were
    // when the row was eliminated) the reduced cost of the kept column in the model after the substitution is
    //   z_k' = rho_k - (a_k / a_j) rho_j ,   and before it   z_j = rho_j - a_j y_i ,  z_k = rho_k - a_k y_i .
    // y_i = rho_j / a_j gives z_j = 0, z_k = z_k' : right whenever x_k is inside its box or on a bound of its own.  If
    // x_k sits on a bound that x_j's box implied, it is x_j that is on its bound: y_i = rho_k / a_k, z_k = 0 and
    // z_j = -(a_j / a_k) z_k' carries the multiplier.
    for (size_t s = recs_.size(); s-- > 0;) {
        const Rec &r = recs_[s];
        x[r.j] = (r.b - r.ak * x[r.k]) / r.aj;
        double rho_j = r.cj;
        for (int e = r.e0; e < r.e1; ++e) rho_j -= ents_[e].second * y[ents_[e].first];
        const double zk = z[r.k];
        const double rho_k = zk + (r.ak / r.aj) * rho_j;
        const bool from_j = (zk > 0.0 && r.lk_new > r.lk_old) || (zk < 0.0 && r.uk_new < r.uk_old);
        if (from_j) {
            const double yi = rho_k / r.ak;
            y[r.i] = yi;
            z[r.k] = 0.0;
            z[r.j] = rho_j - r.aj * yi;
        } else {
            y[r.i] = rho_j / r.aj;
            z[r.j] = 0.0;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// primal bound propagation (infinite bounds only)
// ------------------------------------------------------------------------------------------------
BoundStage::~BoundStage() {
    if (reduced_) free_model(reduced_);
}

bool BoundStage::run(const LP_info_cpu *model) {
    if (!model || !model->A || model->m <= 0 || model->n <= 0) return false;
    org_ = model;
    m_ = model->m;
    n_ = model->n;
    const int m = m_, n = n_;
    const int *rp = model->A->rowPtr, *ci = model->A->colIndex;
    const double *av = model->A->value;
    if (static_cast<long>(rp[m]) > kStageMaxNnz) return false;
    std::vector<double> l(model->l, model->l + n), u(model->u, model->u + n);
    bool any_inf = false;
    for (int j = 0; j < n && !any_inf; ++j) any_inf = !fin(l[j]) || !fin(u[j]);
    if (!any_inf) return false;
    for (int sweep = 0; sweep < kBoundSweeps; ++sweep) {
        const size_t before = recs_.size();
        for (int i = 0; i < m; ++i) {
            const double lo_side = model->AL[i], up_side = model->AU[i];
            if (!fin(lo_side) && !fin(up_side)) continue;
            // activity range of the row over the current box; an infinite end is counted, not added
            double mn = 0.0, mx = 0.0;
            int mn_inf = 0, mx_inf = 0, cnt = 0;
            bool repeated = false;
            for (int k = rp[i]; k < rp[i + 1]; ++k) {
                const double a = av[k];
                if (a == 0.0) continue;
                const int j = ci[k];
                if (k > rp[i] && ci[k - 1] >= j) repeated = true;  // unsorted or repeated column index: leave the row alone
                ++cnt;
                const double bl = a > 0 ? l[j] : u[j], bu = a > 0 ? u[j] : l[j];
                if (fin(bl)) mn += a * bl; else ++mn_inf;
                if (fin(bu)) mx += a * bu; else ++mx_inf;
            }
            if (cnt < 2 || repeated) continue;  // (singleton rows become bounds in the other stage)
            const bool use_up = fin(up_side) && mn_inf <= 1, use_lo = fin(lo_side) && mx_inf <= 1;
            if (!use_up && !use_lo) continue;
            for (int k = rp[i]; k < rp[i + 1]; ++k) {
                const double a = av[k];
                if (a == 0.0) continue;
                const int j = ci[k];
                const double bl = a > 0 ? l[j] : u[j], bu = a > 0 ? u[j] : l[j];
                // from  sum <= AU :  a x_j <= AU - (least activity of the others)
                if (use_up) {
                    const bool own_inf = !fin(bl);
                    if (mn_inf == 0 || (mn_inf == 1 && own_inf)) {
                        const double rest = own_inf ? mn : mn - a * bl;
                        const double v = (up_side - rest) / a;
                        if (a > 0 && !fin(u[j]) && std::abs(v) < kHugeBound) {
                            u[j] = v + kBoundMargin * (1.0 + std::abs(v));
                            recs_.push_back(Rec{i, j, a, false});
                        } else if (a < 0 && !fin(l[j]) && std::abs(v) < kHugeBound) {
                            l[j] = v - kBoundMargin * (1.0 + std::abs(v));
                            recs_.push_back(Rec{i, j, a, true});
                        }
                    }
                }
                // from  sum >= AL :  a x_j >= AL - (largest activity of the others)
                if (use_lo) {
                    const bool own_inf = !fin(bu);
                    if (mx_inf == 0 || (mx_inf == 1 && own_inf)) {
                        const double rest = own_inf ? mx : mx - a * bu;
                        const double v = (lo_side - rest) / a;
                        if (a > 0 && !fin(l[j]) && std::abs(v) < kHugeBound) {
                            l[j] = v - kBoundMargin * (1.0 + std::abs(v));
                            recs_.push_back(Rec{i, j, a, true});
                        } else if (a < 0 && !fin(u[j]) && std::abs(v) < kHugeBound) {
                            u[j] = v + kBoundMargin * (1.0 + std::abs(v));
                            recs_.push_back(Rec{i, j, a, false});
                        }
                    }
                }
            }
            // (the activities above were formed before this row's own tightenings: each column's bound used only the
            // OTHER columns' ends, so nothing derived here depends on a bound derived here)
        }
        if (recs_.size() == before) break;
    }
    if (recs_.empty()) return false;
    for (int j = 0; j < n; ++j)
        if (l[j] > u[j]) {  // an implied bound beyond the opposite bound: infeasible-looking, leave it to the solver
            recs_.clear();
            return false;
        }
    reduced_ = model_from_csr(m, n, static_cast<long>(rp[m]), rp, ci, av, model->AL, model->AU, l.data(), u.data(), model->c,
                              model->obj_constant);
    return reduced_ != nullptr;
}

void BoundStage::postsolve(const double *xr, const double *yr, const double *zr, double *x, double *y, double *z) const {
    std::copy(xr, xr + n_, x);
    std::copy(yr, yr + m_, y);
    std::copy(zr, zr + n_, z);
    // the stage's own copy has the input's rows, entries and indices (only bounds differ): no pointer into the caller's
    // model is followed after run()
    const int *rp = reduced_->A->rowPtr, *ci = reduced_->A->colIndex;
    const double *av = reduced_->A->value;
    // a reduced cost leaning on a bound that only this stage gave the column belongs to the row that implied the bound
    for (size_t s = recs_.size(); s-- > 0;) {
        const Rec &r = recs_[s];
        const double zj = z[r.j];
        if ((r.lower && zj > 0.0) || (!r.lower && zj < 0.0)) {
            const double dy = zj / r.a;
            y[r.i] += dy;
            for (int k = rp[r.i]; k < rp[r.i + 1]; ++k) z[ci[k]] -= av[k] * dy;
            z[r.j] = 0.0;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// the chain
// ------------------------------------------------------------------------------------------------
bool Presolve::run(const LP_info_cpu *model) {
    const auto t0 = std::chrono::steady_clock::now();
    if (!model || !model->A || model->m <= 0 || model->n <= 0) return false;
    m_ = model->m;
    n_ = model->n;
    // HPRLP_PRESOLVE_OFF=doubleton,bounds switches the two stages off (the other names: presolve.cpp)
    const char *off_env = env_get("HPRLP_PRESOLVE_OFF");
    const std::string off = off_env ? off_env : "";
    const bool use_dton = off.find("doubleton") == std::string::npos, use_bounds = off.find("bounds") == std::string::npos;
    // HPRLP_PRESOLVE_ONLY=doubleton|bounds: that stage alone on the model as given (unit tests of the stages)
    if (const char *only = env_get("HPRLP_PRESOLVE_ONLY")) {
        const std::string which = only;
        if (which == "doubleton") {
            auto st = std::make_unique<DoubletonStage>();
            if (!st->run(model)) return false;
            stats_.doubleton_rows = st->eliminated();
            reduced_ = st->reduced();
            chain_.push_back(std::move(st));
            return true;
        }
        if (which == "bounds") {
            auto st = std::make_unique<BoundStage>();
            if (!st->run(model)) return false;
            stats_.tightened_bounds = st->tightened();
            reduced_ = st->reduced();
            chain_.push_back(std::move(st));
            return true;
        }
    }
    const LP_info_cpu *cur = model;
    auto add = [&](const PresolveStats &s) {
        stats_.fixed_cols += s.fixed_cols; stats_.empty_cols += s.empty_cols; stats_.empty_rows += s.empty_rows;
        stats_.singleton_rows += s.singleton_rows; stats_.redundant_rows += s.redundant_rows; stats_.passes += s.passes;
        stats_.dual_fixed_cols += s.dual_fixed_cols; stats_.slack_cols += s.slack_cols; stats_.parallel_rows += s.parallel_rows;
        stats_.parallel_cols += s.parallel_cols; stats_.forcing_rows += s.forcing_rows;
    };
    // The bound stage removes nothing by itself: its box is kept only if the round after it finds something to remove with
    // it (PSLP likewise drops the implied bounds that stayed redundant at the end, Primal_propagation.c:786).
    bool bounds_done = false, pending = false;
    const LP_info_cpu *before_bounds = nullptr;
    const int max_links = env_get("HPRLP_PRESOLVE_MAX_LINKS") ? std::atoi(env_get("HPRLP_PRESOLVE_MAX_LINKS")) : 1 << 20;  // (debugging)
    for (int round = 0; round < kMaxRounds && !solved_ && static_cast<int>(chain_.size()) < max_links; ++round) {
        bool progress = false;
        ++stats_.rounds;
        {
            auto st = std::make_unique<ReduceStage>();
            const bool ok = st->run(cur);
            add(st->stats());
            if (ok) {
                cur = st->reduced();
                chain_.push_back(std::move(st));
                progress = true;
            } else if (st->solved()) {
                chain_.push_back(std::move(st));
                solved_ = true;
                pending = false;
                break;
            }
        }
        if (use_dton && static_cast<int>(chain_.size()) < max_links) {
            auto st = std::make_unique<DoubletonStage>();
            if (st->run(cur)) {
                stats_.doubleton_rows += st->eliminated();
                cur = st->reduced();
                chain_.push_back(std::move(st));
                progress = true;
            }
        }
        if (pending) {
            pending = false;
            if (!progress) {
                chain_.pop_back();
                cur = before_bounds;
                stats_.tightened_bounds = 0;
                break;
            }
        }
        bool added = false;
        if (use_bounds && !bounds_done) {
            bounds_done = true;
            auto st = std::make_unique<BoundStage>();
            if (st->run(cur)) {
                stats_.tightened_bounds += st->tightened();
                before_bounds = cur;
                cur = st->reduced();
                chain_.push_back(std::move(st));
                pending = added = true;
            }
        }
        if (!progress && !added) break;
    }
    if (pending) {  // out of rounds right behind the bound stage
        chain_.pop_back();
        cur = before_bounds;
        stats_.tightened_bounds = 0;
    }
    stats_.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (solved_) return false;
    if (chain_.empty()) return false;
    reduced_ = cur;
    return true;
}

void Presolve::postsolve(const double *xr, const double *yr, const double *zr, double *x, double *y, double *z) const {
    if (chain_.empty()) return;
    std::vector<double> cx, cy, cz, nx, ny, nz;
    const double *px = xr, *py = yr, *pz = zr;
    for (size_t s = chain_.size(); s-- > 0;) {
        const PresolveLink &lk = *chain_[s];
        const int lm = lk.input_m(), ln = lk.input_n();
        if (s == 0) {
            lk.postsolve(px, py, pz, x, y, z);
        } else {
            nx.assign(static_cast<size_t>(ln), 0.0);
            ny.assign(static_cast<size_t>(lm), 0.0);
            nz.assign(static_cast<size_t>(ln), 0.0);
            lk.postsolve(px, py, pz, nx.data(), ny.data(), nz.data());
            cx.swap(nx);
            cy.swap(ny);
            cz.swap(nz);
            px = cx.data();
            py = cy.data();
            pz = cz.data();
        }
    }
}

}  // namespace hprlp
