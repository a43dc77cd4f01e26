// kernels.h -- device data views and launch wrappers of the hand-written gfx950 kernels.
// Everything here is private to the library; the C ABI is in include/*.h.
#pragma once

#include <hip/hip_runtime.h>

#include "tiled.h"

namespace hprlp {

constexpr int kWave = 64;         // CDNA wavefront
constexpr int kWavesPerBlock = 4; // 256-thread workgroups, one row block per wave
constexpr int kThreads = kWave * kWavesPerBlock;
constexpr int kStreamW = 512;     // max nonzeros staged through LDS by one wave (4 KiB per vector)
constexpr int kStreamRows = 64;   // max rows per stream block: one lane per row
constexpr int kLongRow = 64;      // rows longer than this get a whole wave (vector mode: lanes stride the row, then a wave sum).  256 until late in
                                  // round 2: one lane adding a 205-entry row in CSR order is a chain of 205 dependent additions -- 1.5 us of the
                                  // config-3 x-half launch (10.9 -> 9.4 us per iteration).  Rows up to 64 entries keep the plain CSR-order sum
                                  // (bit-identical to the oracle and to the single-workgroup kernel); longer ones agree to rounding.
constexpr int kSplitRow = 4096;   // rows longer than this are cut into chunks of this many nonzeros
constexpr int kNumScalars = 24;   // device scalar slots (see enum Slot)

// Scalar slots filled by the reduction epilogues.  0-9 follow the reference's 10-slot buffer
// (reference include/structs.h:196-206); the rest are ours.
enum Slot : int {
    S_CX = 0,      // c . x_bar
    S_YOBJ_Y = 1,  // y_obj . y_bar
    S_XZ = 2,      // x_bar . z_bar
    S_RD2 = 3,     // |Rd|^2
    S_RP2 = 4,     // |Rp|^2
    S_ADX_DY = 5,  // <A x_temp, y_temp>
    S_DY2 = 6,     // |y_temp|^2
    S_DX2 = 7,     // |x_temp|^2
    S_MOVE_X2 = 8, // |x_bar - last_x|^2
    S_MOVE_Y2 = 9, // |y_bar - last_y|^2
    S_LU2 = 10,    // iteration-0 bound violation
    S_PW_ZZ = 11,  // power iteration z.z
    S_PW_QZ = 12,  // power iteration q.z
    S_PW_ERR2 = 13,
    S_NB2 = 14,    // |conceptual b|^2
    S_NC2 = 15,    // |c|^2
    S_TMP0 = 16,
    S_TMP1 = 17,
    S_SMALL_PW_LAMBDA = 18,  // single-launch power iteration (small.hip): lambda ...
    S_SMALL_PW_ITERS = 19,   // ... and the iteration it stopped at (0: the kernel did not run)
};

// Row-block descriptor: one per wave.  {first row, number of rows, first nonzero, nonzero count}.
// rows==1 && nz>kLongRow: vector mode (lanes stride the row); otherwise stream mode (nz<=kStreamW,
// rows<=kStreamRows): products staged in LDS in CSR order, lane t sums row t sequentially.
struct CsrDev {
    int rows = 0, cols = 0;
    int nnz = 0;
    int nt = 1;  // stream the matrix with nontemporal loads (large matrices); 0: keep it in L2 between launches
    const int *rowptr = nullptr;
    const int *col = nullptr;
    double *val = nullptr;
    const int4 *blk = nullptr;
    int nblk = 0;
    TiledDev tiled;  // optional column-tiled copy (tiled.h); when valid every fused launch uses it
    // rows longer than kSplitRow: {row, first chunk slot, one past last slot, 0}; chunk sums land in
    // long_partial[2*slot + v] and k_long_finish completes the row
    const int4 *longrows = nullptr;
    int nlong = 0;
    double *long_partial = nullptr;
    int csr_grid() const { return (nblk + kWavesPerBlock - 1) / kWavesPerBlock; }
    int finish_grid() const { return (nlong + kThreads - 1) / kThreads; }
    // workgroups of a fused launch on this matrix = number of reduction partials it writes
    int tiled_finish_grid() const { return (rows + kThreads - 1) / kThreads; }
    int grid() const { return tiled.valid ? (tiled.n_pieces > 0 ? tiled_finish_grid() : tiled.grid) : csr_grid() + finish_grid(); }
};

// Device-resident iteration scalars (reference Halpern_params[4] + halpern_inner,
// include/structs.h:153-168).  kx is read by the x-half, ky by the y-half; each half writes the
// other's counter, so no kernel reads a word that is written during the same launch.
struct Ctrl {
    double sigma, lam_sigma, inv_lam_sigma, inv_sigma;
    int kx, ky;
    int pad0, pad1;
};

// Hand-off of the remainder products between the two kernels of an iteration (tiled.h, propagation blocking): the
// half-step that PRODUCES a gathered vector (x_hat or y) holds a super-block's fresh values in LDS at its epilogue and
// writes the products a * v[col] of the OTHER matrix' remainder entries whose source group is that super-block straight
// into the other matrix' P -- the consumer then skips its pre-pass (k_far_products).  gptr == nullptr: no hand-off.
struct FarPush {
    const int *gptr = nullptr;       // n_groups + 1 of the consumer matrix (group g = rows of the producer's super-block g)
    const double *val = nullptr;
    const int *pos = nullptr;
    const uint16_t *lcol = nullptr;
    double *P = nullptr;
    // run tables of the consumer's source side (tiled.h: f_rptr / f_rk / f_rp); null: positions come from `pos`
    const int *rptr = nullptr, *rk = nullptr, *rp = nullptr;
};

struct XHalfArgs {
    const double *y_full;  // gather source (all rows of y)
    double *x, *x_hat;     // local slices (x_hat points into the gathered x_hat buffer)
    const double *l, *u, *c, *last_x;
    double *x_bar, *z_bar, *x_temp;  // check variant only
    Ctrl *ctrl;
    double *partials;  // check variant: 3 x stride
    int stride;
    FarPush push;            // x_hat's products into the remainder buffer of A (consumed by the y-half)
    bool far_ready = false;  // A^T's remainder buffer already holds the products of y_full (pushed by the y-half before)
    // one byte per column saying which of l[j], u[j] the update has to READ (launch_bound_codes): the common bounds -- lower
    // bound 0 or -inf, upper bound +inf -- are constants of the code and cost no 8-byte load (LPs have l = 0, u = +inf on
    // most columns: 16 of the x-half's 56 bytes per column).  nullptr: always load both.
    const unsigned char *lu_code = nullptr;
    // Normal x-halves of a matrix with a tiled copy, inside a run of normal iterations (Solver::run_normal): x is
    // f2' * x_hat + f1' * last_x of values the previous x-half stored and this one reads anyway, so it need not travel through
    // memory between them.  kXRebuild: this launch forms its x from x_hat and last_x with the previous iteration's Halpern
    // factors (bit for bit the value the previous launch would have stored); kXNoStore: this launch does not store x (the next
    // one rebuilds it).  The first launch of a run reads x, the last one stores it: everybody else finds x in memory.
    int x_mode = 0;
};
constexpr int kXRebuild = 1, kXNoStore = 2;
// bit 0: l[j] must be loaded (neither -inf nor +0.0); bit 1: u[j] must be loaded (not +inf); bit 2: l[j] is +0.0
constexpr unsigned kLoadL = 1u, kLoadU = 2u, kZeroL = 4u;
void launch_bound_codes(int n, const double *l, const double *u, unsigned char *code, hipStream_t s);

struct YHalfArgs {
    const double *xhat_full;
    double *y;  // local slice inside the gathered y buffer
    const double *AL, *AU, *last_y;
    double *y_bar, *y_obj, *y_temp;
    Ctrl *ctrl;
    double *partials;  // check variant: 2 x stride
    int stride;
    FarPush push;            // y's products into the remainder buffer of A^T (consumed by the next x-half)
    bool far_ready = false;  // A's remainder buffer already holds the products of xhat_full
    // one byte per row saying which of AL[i], AU[i] the update has to READ (launch_row_codes), as lu_code for the x-half: an
    // infinite side is a constant of the code, an equality row (AL == AU) reads one value for both.  nullptr: load both.
    const unsigned char *row_code = nullptr;
};
// bit 0: AL[i] must be loaded (finite, differs from AU[i]); bit 1: AU[i] must be loaded (finite); bit 2: AL[i] == AU[i] (finite)
constexpr unsigned kLoadLo = 1u, kLoadHi = 2u, kRowEq = 4u;
void launch_row_codes(int m, const double *AL, const double *AU, unsigned char *code, hipStream_t s);

struct FinalizeItem {
    const double *partials;
    int count;
    int slot;
};
struct FinalizeArgs {
    FinalizeItem item[8];
    int n;
};

// return true if the launch filled the `push` buffer (the fused tiled kernel ran and a hand-off was requested)
bool launch_x_half(const CsrDev &AT, const XHalfArgs &a, bool check, hipStream_t s);
bool launch_y_half(const CsrDev &A, const YHalfArgs &a, bool check, hipStream_t s);
// the hand-off lists of a tiled matrix as a consumer (empty if it has no remainder lists)
FarPush far_push_of(const CsrDev &consumer);
// normal half-steps over the remote-column part of a split matrix; base = row sums of the local-column part
void launch_x_half_base(const CsrDev &AT_remote, const XHalfArgs &a, const double *base, hipStream_t s);
void launch_y_half_base(const CsrDev &A_remote, const YHalfArgs &a, const double *base, hipStream_t s);

// |(c - AT y_bar - z_bar) .* col_norm|^2 partials (reference residual_compute_Rd, main_iterate.cu:217-226)
void launch_resid_d(const CsrDev &AT, const double *ybar_full, const double *c, const double *z_bar,
                    const double *col_norm, double *partials, hipStream_t s);
// |max(min(AU - A x_bar,0), AL - A x_bar) .* row_norm|^2 partials, optionally <A x_temp, y_temp> in the same pass
void launch_resid_p(const CsrDev &A, const double *xbar_full, const double *xtemp_full, const double *AL,
                    const double *AU, const double *row_norm, const double *y_temp, bool with_gap,
                    double *partials, int stride, hipStream_t s);
// <A x_temp, y_temp> partials only (reference compute_weighted_norm, main_iterate.cu:486-515)
void launch_gap(const CsrDev &A, const double *xtemp_full, const double *y_temp, double *partials, hipStream_t s);
// out = M v ; optionally partials of out.out and out.q  (power iteration)
// plain product whose epilogue hands the remainder products of the OTHER matrix over (returns true if it did)
bool launch_spmv_push(const CsrDev &M, const double *v_full, double *out, const FarPush &push, hipStream_t s);
void launch_spmv_plain(const CsrDev &M, const double *v_full, double *out, const double *q, bool with_dots,
                       double *partials, int stride, hipStream_t s, bool far_ready = false);

void launch_finalize(const FinalizeArgs &f, double *scalars, hipStream_t s);

// x_temp = x_bar - last_x, y_temp = y_bar - last_y, squared norms -> partials (2 x stride);
// then last_x = x = x_bar, last_y = y = y_bar and the Halpern counter is reset
// (reference update_sigma movement + do_restart, main_iterate.cu:312-322,369-375)
void launch_movement(int n, int m, const double *x_bar, const double *last_x, double *x_temp, const double *y_bar,
                     const double *last_y, double *y_temp, double *partials, int stride, int nblocks, hipStream_t s);
void launch_restart_copy(int n, int m, const double *x_bar, double *x, double *last_x, const double *y_bar,
                         double *y, double *last_y, Ctrl *ctrl, hipStream_t s);
// iteration-0 bound violation (reference residual_compute_lu_kernel)
void launch_lu(int n, const double *x_bar, const double *l, const double *u, const double *col_norm, double *x_temp,
               double *partials, int nblocks, hipStream_t s);
void launch_set_ctrl(Ctrl *ctrl, double sigma, double lambda_max, int reset_k, hipStream_t s);

// scaling (reference src/scaling.cu)
// tval_log / fval_log: the log values of M's tiled copy (launch_tiled_refresh_log) -- the pass then runs through the tiled
// kernel if cr_runs_tiled(M); null: stream kernel
void launch_cr_log_update(const CsrDev &M, const double *other_full, double *result, hipStream_t s, const double *tval_log = nullptr,
                          const double *fval_log = nullptr);
bool cr_runs_tiled(const CsrDev &M);
void launch_exp_clamp(double *v, int n, hipStream_t s);
void launch_row_norm(const CsrDev &M, double *result, int norm, hipStream_t s);
double launch_line_density(const int *rowptr, const int *col, int rows, hipStream_t s);  // distinct 64-byte lines gathered per entry, sampled (synchronises)
// max_i rowptr[i+1] - rowptr[i] (synchronises the stream); entries_in_long (optional): the entries in rows of more than `limit` entries
int launch_longest_row(const int *rowptr, int rows, hipStream_t s, int limit = 0, long *entries_in_long = nullptr);
// the most entries any block of `height` consecutive rows holds (synchronises the stream)
int launch_heaviest_block(const int *rowptr, int rows, int height, hipStream_t s);
// val = op(op(val, first), second) where first/second are the row vector or the gathered column vector
void launch_scale_matrix(const CsrDev &M, const double *rowvec, const double *colvec_full, bool row_first,
                         bool divide, hipStream_t s, double *next_max_norm = nullptr);
void launch_vec_scale(double *x, const double *s, int n, bool divide, hipStream_t s_);
void launch_vec_scal(double *x, double a, int n, hipStream_t s);
void launch_fill(double *x, double a, int n, hipStream_t s);
void launch_bnorm2(const double *AL, const double *AU, int m, double *partials, int nblocks, hipStream_t s);
void launch_norm2(const double *x, int n, double *partials, int nblocks, hipStream_t s);

// power iteration helpers
void launch_check_columns(long nnz, int cols, const int *col, int *bad, hipStream_t s);
void launch_pw_start(int m, unsigned long long seed, long long offset, double *z, hipStream_t s);  // power_start_vector on the device
void launch_pw_normalize(const double *z, double *q, int m, const double *scalars, hipStream_t s);
void launch_pw_err(const double *z, const double *q, int m, const double *scalars, double *partials, int nblocks,
                   hipStream_t s);

// x = b_scale * x_bar / col_norm etc. (reference collect_solution, utils.cu:143-200)
void launch_unscale(int n, int m, const double *x_bar, const double *y_bar, const double *z_bar,
                    const double *col_norm, const double *row_norm, double b_scale, double c_scale, double *xo,
                    double *yo, double *zo, hipStream_t s);

// ---- small-LP path (small.hip): `count` normal iterations in one single-workgroup launch -----------
constexpr int kSmallMaxK = 12;     // matrix entries per thread (1024 threads) -> nnz <= 12288
constexpr int kSmallMaxR = 2;      // rows per thread -> m, n <= 2048
constexpr int kSmallMaxRow = 256;  // longest row / column (its owner thread sums it sequentially)
struct SmallArgs {
    int m, n, nnz;
    const int *A_rowptr, *AT_rowptr;
    const double *AT_val;          // the matrix is read once, in A^T order
    const int *ent_ij;             // per A^T entry: row i of A | column j << 16
    const int *ent_posA;           // per A^T entry: position of the same entry in the CSR order of A
    const int *order_x, *order_y;  // rows of A^T / of A sorted by length, longest first (n / m entries)
    double *x, *x_hat, *y;
    const double *l, *u, *c, *last_x, *AL, *AU, *last_y;
    Ctrl *ctrl;
};
bool small_path_fits(int m, int n, long nnz, int max_row_A, int max_row_AT);
void launch_small_iterations(const SmallArgs &a, int count, hipStream_t s);
// the whole power iteration of a small LP in one launch (small.hip): z0 = start vector (m), out = {lambda, iterations done}
void launch_small_power(const SmallArgs &a, const double *z0, int max_iter, double tol, double *out, hipStream_t s);

// device CSR -> device CSR of the transpose, stable in row order (transpose.hip); all pointers are device memory,
// trp has cols+1 entries, tci / tv nnz
void device_transpose(int rows, int cols, long nnz, const int *rowptr, const int *col, const double *val, int *trp,
                      int *tci, double *tv, hipStream_t s);

// column split of a device CSR matrix into the entries with lo <= column < hi and the rest (transpose.hip)
void device_split_columns(int rows, long nnz, const int *rowptr, const int *col, const double *val, int lo, int hi, DBuf<int> &rp_loc,
                          DBuf<int> &col_loc, DBuf<double> &val_loc, DBuf<int> &rp_rem, DBuf<int> &col_rem, DBuf<double> &val_rem,
                          hipStream_t s);

// multi-GPU neighbour exchange: dst[k] = src[idx[k]] before the sends, dst[idx[k]] = src[k] after the receives
void launch_pack(const double *src, const int *idx, double *dst, int n, hipStream_t s);
void launch_scatter(double *dst, const int *idx, const double *src, int n, hipStream_t s);

constexpr int kReduceBlocks = 512;  // grid of the plain vector reductions

}  // namespace hprlp
