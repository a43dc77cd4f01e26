// solver.h -- device-resident single-LP HPR solver state (private).
// Replaces the reference's HPRLP_workspace_gpu / LP_info_gpu / Scaling_info / HPRLP_restart
// (reference include/structs.h:127-277) without vendor-library handles.
#pragma once

#include <string>

#include <functional>
#include <future>
#include <map>
#include <memory>

#include "common.h"
#include "env.h"
#include "dist.h"
#include "kernels.h"

namespace hprlp {

struct Residuals {  // reference HPRLP_residuals, include/structs.h:255-263
    double err_Rp = 0, err_Rd = 0, primal_obj = 0, dual_obj = 0, rel_gap = 0;
    double kkt = std::numeric_limits<double>::infinity();
};

struct RestartState {  // reference HPRLP_restart, include/structs.h:215-228
    int flag = 0;
    bool first = true;
    double last_gap = std::numeric_limits<double>::infinity();
    double current_gap = std::numeric_limits<double>::infinity();
    double save_gap = std::numeric_limits<double>::infinity();
    double best_gap = std::numeric_limits<double>::infinity();
    double best_sigma = 1.0;
    int inner = 0, sufficient = 0, necessary = 0, long_ = 0, times = 0;
};

struct TraceRow {  // same layout as hprlp_trace_row in include/hprlp_amd.h
    int iter, restart_flag;
    double err_Rp, err_Rd, primal_obj, dual_obj, gap, kkt, sigma, current_gap, lambda_max;
};

// One row-partitioned CSR matrix resident on the device together with its row-block descriptors.
struct DeviceMatrix {
    DBuf<int> rowptr, col;
    DBuf<double> val;
    DBuf<int4> blk, longrows;
    DBuf<double> long_partial;
    DeviceTiled tiled;  // optional column-tiled copy for large matrices with column locality (tiled.h)
    // shape of the tiled copy, set by the solver before upload() / describe(): rows per super-block (tiled.h: 8192 unless
    // lowered for a mid-size matrix) and columns per source group of the remainder lists (= sb_rows of the OTHER matrix)
    int sb_rows = kTileRows, far_group = kTileRows;
    int rem_cap = kTileRemCap;  // entries per remainder step (tiled.h: kPbRemCap for the all-remainder form)
    int tile_cols = kTileCols;  // columns per tile of the copy (tiled.h: kTileCols, or kTileColsNarrow for narrow bands)
    CsrDev view;
    // The tiled copy is built on the host (about a second per 2e8 nonzeros) by a background job started in
    // upload(); until finish_tiling() has run every launch on this matrix uses the stream kernel.  `keep` is held
    // by the job (the A^T arrays, which nobody else owns); rp / ci must stay valid until finish_tiling().
    std::future<std::shared_ptr<TiledHost>> tiling;
    int planned_grid = 0;  // grid of the tiled kernel if the build succeeds (sizes the reduction partials)
    void upload(int rows, int cols, const int *rp, const int *ci, const double *v, std::shared_ptr<void> keep = nullptr);
    // after the arrays are on the device.  min_dense >= 0 overrides the share of the entries the tiled build wants in
    // staged tiles (0: accept any pattern -- everything outside dense tiles goes through the propagation-blocking remainder)
    void describe(int rows, int cols, const int *rp, const int *ci, std::shared_ptr<void> keep, double min_dense_override = -1.0);
    // the same with the host row pointers delivered later and the values possibly still being uploaded (solver.cpp)
    void describe_when(int rows, int cols, long nnz, std::shared_future<const int *> rp_ready, const int *ci, std::shared_ptr<void> keep,
                       double min_dense_override, std::future<void> *values_ready);
    void build_tiled_copy(int rows, int cols, int nnz, const std::function<const int *()> &host_rp, const int *ci, std::shared_ptr<void> keep,
                          double min_dense_override, const std::function<void()> &join_values, struct PhaseTimer &pt);
    int longest_row = 0;  // entries of the longest row (set by describe)
    bool declined_sparse = false;  // the last tiled build was declined for lack of dense tiles (not for size)
    bool declined_shape = false;   // ... not attempted: too few columns for staging to pay, or rows too long for the remainder list
    bool declined_l2 = false;      // ... not attempted (a case of declined_shape): piece form against a stream kernel whose gathers stay in one L2
    bool declined_coalesced = false;  // ... not attempted (a case of declined_shape): neighbouring rows gather from the same lines
    bool declined_skew = false;       // ... not attempted (a case of declined_shape): too many of the entries in long rows
    bool declined_imbalance = false;  // ... not attempted (a case of declined_shape): one block of sb_rows rows holds several times the mean
    bool declined_thin = false;       // a PIECE-form copy was built and dropped: fewer than kPiecesThinRows entries per row (stream kernel instead)
    bool declined_popular = false;    // a copy was built and dropped: its remainder gathers from a few popular columns and the rest from one L2's window
    bool declined_long_rows = false;  // ... not attempted for the length of its rows alone (Solver::pb_fallback_wanted)
    bool declined_few_rows = false;   // ... not attempted for the number of rows alone (fewer than a super-block per CU); Solver::pb_fallback_wanted
    double long_row_share = 0.0;      // share of the entries in rows of more than kSkewRow entries (describe_when; 0 for small matrices)
    double line_density = 1.0;     // distinct 64-byte lines of the gathered vector per entry (kernels.hip: launch_line_density)
    double xcd_gather_bytes = 0.0; // estimate by Solver::choose_sb_rows: bytes of the gathered vector an XCD's eighth of the rows reads (0: unknown)
    void finish_tiling(hipStream_t s);  // wait for the job, upload the copy, fill its values from the CSR values
    void refresh_tiled(hipStream_t s);  // re-gather the tiled values from the CSR values (after scaling)
};

std::vector<int4> build_row_blocks(int rows, const int *rowptr, std::vector<int4> *longrows);
// host only: the stream kernel's block list of a pattern, built and checked as describe_when() does (solver.cpp)
void row_block_plan_host(int rows, int cols, const int *rp, const int *ci, bool with_cuts, long out[6]);

// How one kind of gathered vector (length-m: read through the columns of the A^T shard; length-n:
// through the columns of the A shard) reaches this rank after a half-step.  Dense coupling: one
// in-place all-gather.  Sparse coupling (the shards name less than half of the remote entries, e.g.
// banded or block-angular LPs): every rank packs exactly the entries each peer's column indices
// name, one grouped send/recv moves them, a scatter kernel drops them into the full-length vector.
// The choice is made from the all-gathered request counts, so every rank takes the same branch.
struct HaloPlan {
    bool sparse = false;
    int nsend = 0, nrecv = 0;
    long total_requests = 0;  // over all ranks (info)
    DBuf<int> send_idx, recv_idx;  // positions in the gathered vector, grouped by peer, ascending
    DBuf<double> sendbuf, recvbuf;
    std::vector<P2P> ops;
    void build(Comm *comm, const int *cols, long nnz, int total, int chunk, hipStream_t s);
};

struct Solver {
    HPRLP_parameters prm;
    int m = 0, n = 0;          // global sizes
    int m_loc = 0, n_loc = 0;  // rows of A / rows of A^T owned by this rank
    int row_off = 0, col_off = 0;
    int m_pad = 0, n_pad = 0;  // gathered-vector lengths (multiple of the chunk size)
    double obj_constant = 0.0;
    bool verbose = true;
    hipStream_t stream = nullptr;
    Comm *comm = nullptr;
    // Second communicator (its own unique id) for the exchanges enqueued on comm_stream beside the local part of a
    // half-step: a communicator is only ever driven from ONE stream.  Null: the in-process group (host-blocking, no
    // concurrency) or a launcher that handed over a single id -- then comm serves both streams as in rounds 1-2.
    Comm *xcomm = nullptr;

    DeviceMatrix A, AT;  // A: m_loc x n (global columns); AT: n_loc x m (global columns)
    // Set-up time locality ordering (reorder.cpp): when set, the device holds P A Q and all per-row / per-column vectors
    // in the permuted numbering; perm_r[i] / perm_c[j] = the caller's index of permuted row i / column j.  The C ABI
    // (get / set_vector, collect_solution) speaks the caller's numbering.
    std::vector<int> perm_r, perm_c;
    double reorder_time = 0.0, reorder_before = 0.0, reorder_after = 0.0;
    // Callers that read A / AT / row_norm / col_norm directly in the caller's numbering (solve_batched) switch the
    // ordering off before setup(): they would otherwise pair permuted scale vectors with unpermuted panels.
    bool allow_reorder = true;
    bool try_reorder(const LP_info_cpu *model);
    void choose_sb_rows(const LP_info_cpu *model);  // super-block heights of this LP's tiled copies (tiled.h), before the matrices are described
    void choose_pb_rows(DeviceMatrix &M, DeviceMatrix &other, int rows, int other_rows);  // ... of a matrix without column locality (all-remainder form, tiled.h)
    // (other_rowptr: the device row pointers of M's transpose, other_rows + 1 of them -- its view need not be described yet)
    bool pb_fallback_wanted(const DeviceMatrix &M, const int *other_rowptr, int other_rows) const;  // unstructured large matrix: tiled form without dense-tile requirement
    // Hand-off of the remainder products between the two kernels of an iteration (kernels.h: FarPush).  far_A_ready: A's
    // remainder buffer holds the products of the current x_hat (written by the x-half's epilogue); far_AT_ready likewise
    // for y.  Every other launch on a tiled matrix refills its buffer for another vector: invalidate_far().
    bool far_A_ready = false, far_AT_ready = false;
    bool graph_end_A = false, graph_end_AT = false;  // what a replayed iteration graph leaves behind
    void invalidate_far() { far_A_ready = far_AT_ready = false; }
    FarPush push_into(const DeviceMatrix &consumer, const DeviceMatrix &producer) const;  // called by setup() for a large matrix that failed the tiling test
    HaloPlan halo_m, halo_n;  // exchange of length-m / length-n gathered vectors (multi-GPU only)
    DBuf<double> AL, AU, l, u, c, row_norm, col_norm;
    DBuf<unsigned char> row_code;  // per row: which of AL, AU the y-half reads (kernels.h); follows AL / AU (refresh_bound_codes)
    DBuf<unsigned char> lu_code;  // per column: which of l, u the x-half reads (kernels.h); follows l / u (refresh_bound_codes)
    void refresh_bound_codes();
    // local work vectors
    DBuf<double> x, last_x, z_bar, last_y, y_obj, y_temp;
    // gathered vectors (length *_pad); the local slice starts at *_off
    DBuf<double> gy, gxh, gxb, gyb, gxt, gsn, gsm;
    double *y = nullptr, *x_hat = nullptr, *x_bar = nullptr, *y_bar = nullptr, *x_temp = nullptr;
    DBuf<double> sm1, sn1;  // local scratch
    DBuf<Ctrl> ctrl;
    DBuf<double> scal;
    HBuf<double> scal_h;
    DBuf<double> part_x, part_y, part_r, part_v;
    int stride_x = 0, stride_y = 0;

    double b_scale = 1, c_scale = 1, norm_b = 0, norm_c = 0, norm_b_org = 1, norm_c_org = 1;
    double sigma = 1.0, lambda_max = 1.0;
    double setup_time = 0, scaling_time = 0, power_time = 0;
    double fetch_enqueue_s = 0, fetch_wait_s = 0;  // HPRLP_TIMING: host time in fetch_scalars (enqueue of the copy / wait for the stream)
    long fetches = 0;
    int power_iters = 0;
    bool use_graph = true;
    bool use_small = false;  // Netlib-scale LP on one GPU: normal iterations run in the single-workgroup kernel (small.hip)
    int max_row_A = 0, max_row_AT = 0;
    DBuf<int> small_order_x, small_order_y;  // rows of A^T / A sorted by length (row ownership in small.hip)
    DBuf<int> small_ij, small_posA;          // per A^T entry: i | j << 16, position in the CSR order of A

    std::map<int, hipGraphExec_t> graphs;
    TraceRow *trace = nullptr;
    int trace_cap = 0, trace_n = 0;

    Solver() = default;
    ~Solver();
    Solver(const Solver &) = delete;

    // reference copy_lpinfo_to_device + allocate_memory (src/preprocess.cu:66-256)
    void setup(const LP_info_cpu *model, const HPRLP_parameters *param);
    // distributed variant: the caller supplies this rank's rows of A and rows of A^T
    void setup_shard(int m_glob, int n_glob, int row_off_, int m_loc_, int col_off_, int n_loc_, const int *Arp,
                     const int *Aci, const double *Av, const int *ATrp, const int *ATci, const double *ATv,
                     const double *AL_, const double *AU_, const double *l_, const double *u_, const double *c_,
                     double obj_constant_, const HPRLP_parameters *param, Comm *comm_);
    void scale();                                                   // src/scaling.cu:88-216
    double power_iteration(int max_iter, double tol, int *iters);   // src/power_iteration.cu:20-119
    void init_iteration_state();                                    // src/HPRLP.cu:154-167
    void set_sigma_lambda(double sigma_, double lambda_, bool reset_k);
    void reset_iterates();                                          // all iterates back to zero (as after create + scale + power iteration)
    void step(bool check);                                          // one HPR iteration
    void run_normal(int count);                                     // count normal iterations (graph replay)
    void run_normal_then_check(int count);                          // count normal iterations, then one check-variant iteration
    void fetch_scalars();
    void compute_residuals(int iter, bool compute_gap, Residuals *r, RestartState *rs);  // main_iterate.cu:229-309
    double weighted_norm_after_restart();                           // main_iterate.cu:486-515
    void update_sigma_and_restart(RestartState *rs, const Residuals &r);  // main_iterate.cu:312-322,367-404
    void solve_loop(HPRLP_results *out);                            // src/HPRLP.cu:154-310
    void collect_solution(HPRLP_results *out);                      // src/utils.cu:143-200
    double reduce_sum_sq(const double *v, int n_local);             // allreduced ||v||^2
    void verify_exchange();                                         // set-up self-test of the exchange (multi-GPU only)
    void gather(double *gbuf, bool is_m) { gather_on(gbuf, is_m, stream); }  // all-gather a length-m or length-n vector
    void gather_on(double *gbuf, bool is_m, hipStream_t s);

    // Multi-GPU overlap of the exchange with the local part of the next half-step: both shards are split
    // by columns into the part that reads this rank's own slice of the gathered vector and the part that reads remote
    // entries.  While the exchange runs on comm_stream, the local part's row sums go to `part`; the remote part's
    // kernel adds them and runs the epilogue.  Built after scaling (the copies carry the scaled values).
    struct SplitShard {
        DeviceMatrix loc, rem;
        DBuf<double> part;
    };
    std::unique_ptr<SplitShard> ovA, ovAT;
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_ready = nullptr, ev_done_x = nullptr, ev_done_y = nullptr;
    // the environment switches in effect when this solver was set up (env.h), and the test hooks found set but ignored
    std::string env_at_setup, env_ignored_at_setup;
    bool hook_no_far_push = false, hook_no_bound_codes = false, hook_store_x = false;  // test hooks read once per solver (read_hooks), used every iteration
    void read_hooks();
    bool overlap_enabled = false, overlap_ready = false, y_exchange_pending = false;
    bool overlap_spmv_first = false;  // launch order of the local SpMV and the exchange (launch_normal_pair)
    void prepare_overlap();
    void ensure_comm_stream();
    void allreduce_scalars();
    void finish_tiling();  // adopt tiled copies whose background build is still pending (no-op otherwise)

    // One normal iteration.  ev (optional, 3 events): recorded before the x-half, between the halves, after the y-half.
    // more_follow (multi-GPU overlap only): the caller launches another normal pair next, so the exchange of y may stay
    // in flight behind the local part of that pair's x-half; otherwise the gathered y is complete on return.
    void launch_normal_pair(bool more_follow = false, hipEvent_t *ev = nullptr, int x_mode = 0);
    // kernels.h XHalfArgs::x_mode of iteration `i` of a run of `count` normal iterations enqueued back to back (0 where the
    // x-half's epilogue has no such mode: no tiled copy, split shards)
    int x_mode_of(int i, int count) const;

   private:
    void alloc_work();
    hipGraphExec_t graph_for(int len);
};

}  // namespace hprlp
