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
// Two more stages (classes below) change the matrix or the box: doubleton equations and primal bound propagation.
// tests/test_presolve.py compares with PSLP on the same LPs.  Convention (as the solver and PSLP):
//   min c.x  s.t.  AL <= A x <= AU,  l <= x <= u,   z = c - A^T y,  y_i > 0 <=> row at AL.
// Any doubt (infeasible or unbounded-looking input, nothing left to solve) makes run() return false
// and solve() falls back to the original model, like the reference when its worker fails.
#pragma once

#include <memory>
#include <utility>
#include <vector>

#include "structs.h"

namespace hprlp {

LP_info_cpu *model_from_csr(int m, int n, long nnz, const int *rp, const int *ci, const double *v, const double *AL,
                            const double *AU, const double *l, const double *u, const double *c, double obj_constant);

struct PresolveStats {
    int fixed_cols = 0, empty_cols = 0, empty_rows = 0, singleton_rows = 0, redundant_rows = 0, passes = 0;
    int dual_fixed_cols = 0, slack_cols = 0, parallel_rows = 0, parallel_cols = 0, forcing_rows = 0;
    int doubleton_rows = 0, tightened_bounds = 0, rounds = 0;
    double seconds = 0.0;
};

// One link of the presolve chain: maps a model to a smaller / tighter one and a primal-dual solution of that one back.
class PresolveLink {
   public:
    virtual ~PresolveLink() = default;
    virtual const LP_info_cpu *reduced() const = 0;
    virtual int input_m() const = 0;
    virtual int input_n() const = 0;
    // (xr, yr, zr) of reduced() -> (x, y, z) of the link's input model; all three null when the link left nothing
    virtual void postsolve(const double *xr, const double *yr, const double *zr, double *x, double *y, double *z) const = 0;
};

// The fixed-point loop over the reductions that leave the matrix entries alone (list at the top of this file).
class ReduceStage : public PresolveLink {
   public:
    using Stats = PresolveStats;
    ReduceStage() = default;
    ~ReduceStage() override;
    ReduceStage(const ReduceStage &) = delete;
    ReduceStage &operator=(const ReduceStage &) = delete;

    // Returns true when a smaller, non-empty model was produced (reduced() is then valid).
    bool run(const LP_info_cpu *model);
    const LP_info_cpu *reduced() const override { return reduced_; }
    // run() returned false because NOTHING was left: postsolve(nullptr, nullptr, nullptr, ...) yields the optimum
    bool solved() const { return solved_; }
    const Stats &stats() const { return stats_; }
    int input_m() const override { return m_; }
    int input_n() const override { return n_; }
    void postsolve(const double *xr, const double *yr, const double *zr, double *x, double *y, double *z) const override;

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

// Doubleton equations (PSLP: DtonsEq): a row  a_j x_j + a_k x_k = b  defines x_j = (b - a_k x_k) / a_j.  Column j is
// substituted out of every other row and of the cost, its bounds become bounds of x_k, the row goes.  The matrix entries
// change (column j's entries are merged into column k's), so this stage works on its own row lists and hands a new
// model on.  Postsolve: x_j from the row; the row's multiplier is the one that zeroes the reduced cost of whichever of
// the two columns is NOT sitting on the bound that decides (x_k on a bound x_j's box implied: the reduced cost belongs
// to x_j).
class DoubletonStage : public PresolveLink {
   public:
    DoubletonStage() = default;
    ~DoubletonStage() override;
    DoubletonStage(const DoubletonStage &) = delete;
    DoubletonStage &operator=(const DoubletonStage &) = delete;
    bool run(const LP_info_cpu *model);  // true: at least one row eliminated, reduced() valid
    const LP_info_cpu *reduced() const override { return reduced_; }
    int input_m() const override { return m_; }
    int input_n() const override { return n_; }
    int eliminated() const { return static_cast<int>(recs_.size()); }
    void postsolve(const double *xr, const double *yr, const double *zr, double *x, double *y, double *z) const override;

   private:
    struct Rec {
        int i, j, k;          // row, substituted column, kept column
        double aj, ak, b;     // a_j x_j + a_k x_k = b
        double lk_old, uk_old, lk_new, uk_new;
        double cj;            // cost of column j when it was substituted
        int e0, e1;           // ents_[e0, e1): (row, a_rj) of column j's other rows at that time
    };
    int m_ = 0, n_ = 0;
    std::vector<Rec> recs_;
    std::vector<std::pair<int, double>> ents_;
    std::vector<int> row_of_, col_of_;
    LP_info_cpu *reduced_ = nullptr;
};

// Primal bound propagation (PSLP: Primal_propagation, infinite bounds only): a row whose other columns are boxed implies
// a bound on each of its columns; an INFINITE bound of a column is replaced by the implied one, loosened by a margin so
// that it stays redundant (same feasible set, same optimal set).  No row or column goes: the gain is a boxed model for
// the solver and finite activities for the reductions of the next round.  Postsolve: x, y unchanged; a reduced cost that
// leans on a bound the original model does not have is moved onto the row that implied the bound.
class BoundStage : public PresolveLink {
   public:
    BoundStage() = default;
    ~BoundStage() override;
    BoundStage(const BoundStage &) = delete;
    BoundStage &operator=(const BoundStage &) = delete;
    bool run(const LP_info_cpu *model);  // true: at least one bound tightened, reduced() valid
    const LP_info_cpu *reduced() const override { return reduced_; }
    int input_m() const override { return m_; }
    int input_n() const override { return n_; }
    int tightened() const { return static_cast<int>(recs_.size()); }
    void postsolve(const double *xr, const double *yr, const double *zr, double *x, double *y, double *z) const override;

   private:
    struct Rec {
        int i, j;      // the row that implied the bound, the column
        double a;      // a_ij
        bool lower;    // which bound of column j was infinite and is now finite
    };
    int m_ = 0, n_ = 0;
    const LP_info_cpu *org_ = nullptr;
    std::vector<Rec> recs_;
    LP_info_cpu *reduced_ = nullptr;
};

// The presolver solve() uses: rounds of ReduceStage -> DoubletonStage (-> BoundStage in the first round) until nothing
// changes; postsolve walks the chain backwards.
class Presolve {
   public:
    using Stats = PresolveStats;
    Presolve() = default;
    Presolve(const Presolve &) = delete;
    Presolve &operator=(const Presolve &) = delete;

    // Returns true when a different, non-empty model was produced (reduced() is then valid).
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
    int m_ = 0, n_ = 0;
    std::vector<std::unique_ptr<PresolveLink>> chain_;
    const LP_info_cpu *reduced_ = nullptr;  // owned by the last link
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
