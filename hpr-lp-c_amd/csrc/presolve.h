// presolve.h -- host-side LP presolve / postsolve wrapped around solve().
//
// The reference runs the vendored PSLP presolver in a forked child and talks to it over pipes
// (reference src/pslp_integration.cpp:219-339,628-759; wired into solve() at src/HPRLP.cu:504-521).
// This is our own presolver, run in-process (no fork next to a live HIP context): a fixed-point loop
// over the reductions whose postsolve is exact for a primal-dual pair --
//   fixed columns (l == u), empty rows, singleton rows (turned into column bounds), redundant rows
//   (activity bounds inside [AL, AU]), empty columns (moved to the bound the cost prefers), dual fixing (columns
//   whose cost and rows all push them to one bound) and slack columns (a column that appears in one row only is
//   eliminated when the row is an equality -- its cost moves onto the row's other columns -- or when its cost is
//   zero: the row's sides widen by the column's range), parallel rows (a row that is a multiple of another one over
//   the live columns is folded into it: intersection of the sides) and parallel columns (a column that is a multiple
//   of another one, cost included, is folded into it: the kept column stands for the weighted sum), forcing rows (the
//   box allows only one activity inside the row's sides: every column of the row is pinned to the bound realising it).
// (A free or implied-free singleton column with a cost turns its row into the equality its multiplier c_j / a demands
// first; implied free: the row and the other columns' bounds already keep it inside its own bounds.)
// PSLP applies more (doubleton equations, bound propagation);
// tests/test_presolve.py compares both on the same LPs.  Convention (as the solver and PSLP):
//   min c.x  s.t.  AL <= A x <= AU,  l <= x <= u,   z = c - A^T y,  y_i > 0 <=> row at AL.
// Any doubt (infeasible or unbounded-looking input, nothing left to solve) makes run() return false
// and solve() falls back to the original model, like the reference when its worker fails.
#pragma once

#include <vector>

#include "structs.h"

namespace hprlp {

LP_info_cpu *model_from_csr(int m, int n, long nnz, const int *rp, const int *ci, const double *v, const double *AL,
                            const double *AU, const double *l, const double *u, const double *c, double obj_constant);

class Presolve {
   public:
    struct Stats {
        int fixed_cols = 0, empty_cols = 0, empty_rows = 0, singleton_rows = 0, redundant_rows = 0, passes = 0;
        int dual_fixed_cols = 0, slack_cols = 0, parallel_rows = 0, parallel_cols = 0, forcing_rows = 0;
        double seconds = 0.0;
    };
    Presolve() = default;
    ~Presolve();
    Presolve(const Presolve &) = delete;
    Presolve &operator=(const Presolve &) = delete;

    // Returns true when a smaller, non-empty model was produced (reduced() is then valid).
    bool run(const LP_info_cpu *model);
    const LP_info_cpu *reduced() const { return reduced_; }
    // run() returned false because NOTHING was left: postsolve(nullptr, nullptr, nullptr, ...) yields the optimum
    bool solved() const { return solved_; }
    const Stats &stats() const { return stats_; }
    int original_m() const { return m_; }
    int original_n() const { return n_; }
    // Maps a primal-dual solution of the reduced model back to the original dimensions.
    void postsolve(const double *xr, const double *yr, const double *zr, double *x, double *y, double *z) const;

   private:
    bool worth_it(const LP_info_cpu *model) const;  // large models: is there enough to remove?
    enum Kind : int { FixedCol, EmptyCol, EmptyRow, SingletonRow, RedundantRow, DualFixCol, SlackCol, ParallelRow, ParallelCol, ForcingRow };
    struct Record {
        Kind kind;
        int i, j;
        double a, v;
        double l_old, u_old, l_new, u_new;
        double cost = 0.0;  // column records: the column's cost when it was removed (slack substitution moves cost)
    };
    int m_ = 0, n_ = 0;
    const LP_info_cpu *org_ = nullptr;
    std::vector<int> trp_, tci_;  // CSC of the original matrix (postsolve of fixed columns)
    std::vector<double> tv_;
    std::vector<int> row_of_, col_of_;  // reduced index -> original index
    std::vector<Record> stack_;
    LP_info_cpu *reduced_ = nullptr;
    bool solved_ = false;
    Stats stats_;
};

// original-model KKT metrics of a postsolved solution (reference compute_original_kkt_metrics,
// src/pslp_integration.cpp:499-580): relative primal / dual infeasibility and objective gap
struct OriginalKkt {
    double primal_feas = 0, dual_feas = 0, gap = 0, primal_obj = 0, dual_obj = 0;
};
OriginalKkt original_kkt(const LP_info_cpu *model, const double *x, const double *y, const double *z);

}  // namespace hprlp
