/*
 * hprlp_amd.h -- step-level C ABI of the MI355X HPR-LP library (extension of HPRLP.h).
 *
 * The reference exposes its solver phases only as C++-linkage functions on a CUDA workspace struct
 * (reference include/preprocess.h, scaling.h, power_iteration.h, main_iterate.h).  These entry
 * points expose the same phases over an opaque handle with plain pointers and sizes, so that the
 * parity tests, bench.py and a multi-GPU launcher can drive them from any language.  Each function
 * cites the reference function it stands for.  All functions return 0 / a valid value on success
 * and a negative value (or NULL) on failure with the message available from hprlp_last_error();
 * none throws across the boundary.
 */
#ifndef HPRLP_AMD_H
#define HPRLP_AMD_H

#include "HPRLP.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hprlp_solver hprlp_solver; /* opaque: device-resident scaled LP + iteration state */

/* One row per residual evaluation = one line of the reference's iteration log (src/HPRLP.cu:207-218). */
typedef struct hprlp_trace_row {
    int iter, restart_flag;
    double err_Rp, err_Rd, primal_obj, dual_obj, gap, kkt, sigma, current_gap, lambda_max;
} hprlp_trace_row;

const char *hprlp_last_error(void);
const char *hprlp_backend(void); /* "hip-gfx950" */
/* Warm-up: start the HIP runtime, create the device context and a first stream and load the library's code objects NOW instead
 * of inside the first solve of the process (0.10 s there against a 0.04 s solve of a Netlib-scale LP, profiles/r04_cold_start.txt).
 * For callers that serve many solves: call once at start-up.  Returns 0; -1 without a usable GPU or for a device
 * outside [0, device count); -2 when a code object did not load (hprlp_last_error() says which).
 * hprlp_warmup_seconds: {runtime start-up, device context + first stream, code objects, total} of the last call. */
int hprlp_warmup(int device);
int hprlp_warmup_seconds(double out[4]);

/* The library keeps freed device blocks of 1 MiB and more for its next solver (a hipMalloc of a multi-GB set-up temporary right
 * behind a large hipFree stalls for up to a second on this platform); this returns them all to the driver.  HPRLP_NO_ALLOC_CACHE=1
 * disables the cache. */
void hprlp_release_device_cache(void);
/* Bookkeeping self-test of that cache without a GPU (blocks are filed under the device that owns them, the cap is per
 * device): 0 = passed, else the number of the failed check. */
int hprlp_alloc_cache_selftest(void);

/* copy_lpinfo_to_device + allocate_memory (reference src/preprocess.cu:66-256): uploads A, builds A^T
 * and the wave row-block descriptors, allocates the work vectors.  Does not scale. */
hprlp_solver *hprlp_solver_create(const LP_info_cpu *model, const HPRLP_parameters *param);
void hprlp_solver_destroy(hprlp_solver *s);
void hprlp_solver_set_verbose(hprlp_solver *s, int verbose);

/* scaling() (reference src/scaling.cu:88-216) */
int hprlp_solver_scale(hprlp_solver *s);
/* power_method_cusparse() (reference src/power_iteration.cu:20-119); returns lambda (not x1.01) */
double hprlp_solver_power_iteration(hprlp_solver *s, int max_iter, double tol, int *iters_out);
/* sigma_0, Halpern reset (reference src/HPRLP.cu:154-167).  sigma<=0: norm_b/norm_c rule. */
int hprlp_solver_init(hprlp_solver *s, double sigma, double lambda_max);
/* All iterates and work vectors back to zero -- the state after create + scale + power iteration (the reference starts every
 * solve from zero vectors, src/preprocess.cu:41-101 allocate_memory); follow with hprlp_solver_init.  Collective-free. */
int hprlp_solver_reset_iterates(hprlp_solver *s);
/* `normal` normal iterations then, if then_check, one check-variant iteration
 * (reference update_zx_*_gpu + update_y_*_gpu, src/main_iterate.cu:422-481) */
int hprlp_solver_iterate(hprlp_solver *s, int normal, int then_check);
/* compute_residuals() (reference src/main_iterate.cu:229-309):
 * out = {err_Rp, err_Rd, primal_obj, dual_obj, rel_gap, kkt, weighted_norm (if compute_gap), lambda_max} */
int hprlp_solver_residuals(hprlp_solver *s, int iter, int compute_gap, double out[8]);
/* update_sigma + do_restart as if check_restart had raised a flag (reference main_iterate.cu:312-404);
 * in = {current_gap, best_gap, best_sigma, err_Rd, err_Rp, rel_gap}; returns new sigma in *sigma_out */
int hprlp_solver_restart(hprlp_solver *s, const double in[6], double *sigma_out);
/* compute_weighted_norm() (reference main_iterate.cu:486-515) */
double hprlp_solver_weighted_norm(hprlp_solver *s);
/* the whole loop from the current state (reference src/HPRLP.cu:154-310) + collect_solution */
int hprlp_solver_run(hprlp_solver *s, HPRLP_results *out, hprlp_trace_row *trace, int max_trace, int *n_trace);

/* Named device vectors: x y x_hat x_bar y_bar z_bar x_temp y_temp y_obj last_x last_y AL AU l u c
 * row_norm col_norm A_val AT_val.  get returns the length (or -1); cap is the capacity of out. */
long hprlp_solver_get_vector(hprlp_solver *s, const char *name, double *out, long cap);
int hprlp_solver_set_vector(hprlp_solver *s, const char *name, const double *in, long len);
/* out = {b_scale, c_scale, norm_b, norm_c, norm_b_org, norm_c_org, sigma, lambda_max,
 *        setup_time, scaling_time, power_time, power_iters, kx, ky} */
int hprlp_solver_get_scalars(hprlp_solver *s, double out[16]);
/* out = {m, n, nnz, row blocks of A, row blocks of A^T, grid of y-half, grid of x-half,
 *        tiled flags (bit0: A, bit1: A^T use the column-tiled kernel)} */
int hprlp_solver_info(hprlp_solver *s, long out[8]);

/* Wall-clock phases [s] of the calling thread's last HPRLP_main_solve (what solve() runs after presolve):
 * out = {device set-up (upload, transpose, tiled copies, ordering), scaling, power iteration, loop, solution's way back,
 *        teardown of the device state, whole call, 0}.  The reference's instrument covers power iteration + loop only
 * (HPRLP_results.time, src/HPRLP.cu:150,246). */
int hprlp_last_solve_phases(double out[8]);

/* Human-readable: which kernel form runs on A and on A^T (stream / tiled fused / tiled pieces), super-blocks, steps, share of the
 * entries in staged tiles, long rows kept aside, small-LP kernel, locality ordering.  Returns the length (truncated to cap). */
int hprlp_solver_describe(hprlp_solver *s, char *buf, int cap);
/* Every environment switch the library reads, one per line: "<name>\t<integrator|hook>\t<what>" (returns the text's length; buf
 * may be NULL).  "integrator" switches are always honoured; "hook" switches (tests, measurements) only when HPRLP_TEST_HOOKS=1 is
 * set too.  hprlp_solver_describe ends with the switches a solver was set up under, so a non-default path is never silent. */
int hprlp_env_switches(char *buf, int cap);

/* Timed normal iterations for bench.py.  mode 0: graph replay as the product runs it; wall time by
 * HIP events around the whole batch.  mode 1: eager launches with an event pair around every kernel
 * on the solver's stream; xhalf_ms / yhalf_ms are the SUMS of the x-half / y-half kernel durations.  mode 2: the
 * bare SpMVs A^T y and A x_hat into scratch (no update, iterate untouched), timed like mode 1. */
int hprlp_solver_time_iterations(hprlp_solver *s, int warmup, int steps, int mode, double *total_ms,
                                 double *xhalf_ms, double *yhalf_ms);

/* ---- benchmark utility -------------------------------------------------------------------------
 * Rows [row0,row0+rows) of the banded-random matrix of BASELINE.json config 5 (per_row entries per
 * row, 95 % within +-band of the diagonal, 5 % anywhere, N(0,1) values).  Each row depends only on
 * (seed,row).  rowptr has rows+1 entries, col/val rows*per_row.  Not part of the solve path. */
int hprlp_gen_banded_csr(int m, int n, int per_row, int band, unsigned long long seed, int row0, int rows,
                         int *rowptr, int *col, double *val, int nthreads);

/* Rows [col_off, col_off + n_loc) of the TRANSPOSE of that matrix (= the columns a rank of a row-partitioned run owns:
 * hprlp_shard::AT_*), produced by sweeping all m rows of the generator and keeping the owned columns -- no rank holds the whole
 * matrix and nothing is communicated.  Equal entry for entry to the stable host transpose (reference src/utils.cu:203-232) of the
 * matrix generated whole.  trp: n_loc + 1 entries; *tci / *tv are malloc'd (release with hprlp_host_free); *nnz = their length. */
int hprlp_gen_banded_csr_transposed(int m, int n, int per_row, int band, unsigned long long seed, int col_off, int n_loc,
                                    int *trp, int **tci, double **tv, long *nnz, int nthreads);
void hprlp_host_free(void *p);

/* Benchmark utility: P A Q on the host (row i of the result = row row_new2old[i] of A, columns renumbered by col_old2new and
 * sorted) -- builds the randomly permuted variant of config 5 that the set-up time locality ordering has to undo. */
int hprlp_permute_csr_host(int m, int n, const int *rowptr, const int *col, const double *val, const int *row_new2old,
                           const int *col_old2new, int *rowptr_out, int *col_out, double *val_out, int nthreads);

/* ---- row-partitioned multi-GPU solve (new design; the reference is single-GPU) -----------------
 * Rank p owns rows [p*ceil(m/P),...) of A with y/AL/AU and rows [p*ceil(n/P),...) of A^T with
 * x/c/l/u; one in-place RCCL all-gather of the fresh vector slice follows each half-step. */
typedef struct hprlp_shard {
    int m, n;              /* global sizes */
    int row_off, m_loc;    /* rows of A owned by the rank */
    int col_off, n_loc;    /* rows of A^T (= columns of A) owned by the rank */
    int *A_rowptr, *A_col; /* m_loc x n, global column indices */
    double *A_val;
    int *AT_rowptr, *AT_col; /* n_loc x m, global column (= row of A) indices */
    double *AT_val;
    double *AL, *AU;       /* m_loc */
    double *l, *u, *c;     /* n_loc */
    double obj_constant;
} hprlp_shard;

/* block partition of `total` items over `parts` ranks: returns the chunk size ceil(total/parts) */
int hprlp_partition(int total, int parts, int rank, int *offset, int *count);
/* host only: cut this rank's shard out of a full model (arrays malloc'd; release with hprlp_free_shard) */
int hprlp_extract_shard(const LP_info_cpu *model, int rank, int size, hprlp_shard *out);
void hprlp_free_shard(hprlp_shard *s);
/* rank 0: create the RCCL unique id(s), 128 bytes each; the launcher broadcasts them (bench.py: torch.distributed).
 * bytes >= 256 yields TWO ids: hprlp_solver_create_dist* called with id_bytes >= 256 then builds a second communicator
 * for the exchange stream (exchanges that overlap the local part of a half-step), so that no communicator is driven
 * from two streams; with one id the single communicator serves both. */
int hprlp_dist_unique_id(void *out, int bytes);
/* HPRLP_DIST_TRANSPORT=shm in the environment of hprlp_dist_unique_id's caller: the id names a POSIX shared-memory segment and
 * hprlp_solver_create_dist* given that id build a host-staged group of processes on ONE node (device -> pinned shared area ->
 * the reader's device; no RCCL, no device IPC handle): the transport of last resort, and the multi-process form a one-GPU box
 * can run.  hprlp_shm_transport_selftest: the protocol alone on host buffers (no GPU): `rounds` rounds of all-gather, scalar
 * all-reduce and a ragged neighbour exchange, every payload checked; hang_rank >= 0 leaves half-way without a word (the others
 * must fail after HPRLP_DIST_TIMEOUT_S seconds, default 120).  0, or -1 + hprlp_last_error(). */
int hprlp_shm_transport_selftest(const void *unique_id, int id_bytes, int rank, int size, int rounds, int hang_rank);
/* every rank: create the solver for its shard of `model` (param->device_number selects the GPU).
 * get_vector/run then return this rank's slices; scalars/residuals are global. */
hprlp_solver *hprlp_solver_create_dist(const LP_info_cpu *model, const HPRLP_parameters *param, int rank, int size,
                                       const void *unique_id, int id_bytes);
/* The same from a shard the caller assembled itself -- rows [row_off, row_off + m_loc) of A and rows [col_off, col_off + n_loc)
 * of A^T in the block partition of hprlp_partition(), global column indices, arrays owned by the caller (copied to the
 * device during the call).  No rank has to hold the whole matrix (config 5 is generated per shard);
 * hpr-lp-c_amd/shard.py builds the A^T rows of every rank from the ranks' A rows with one all-to-all. */
hprlp_solver *hprlp_solver_create_dist_from_shard(const hprlp_shard *shard, const HPRLP_parameters *param, int rank, int size,
                                                  const void *unique_id, int id_bytes);
/* How the fresh slices travel: if the shards' column indices name at most half of the remote entries
 * (banded / block-structured LPs) each rank sends exactly the entries its peers read (pack kernel, one grouped
 * RCCL send/recv, scatter kernel); otherwise one in-place all-gather.  HPRLP_DIST_EXCHANGE=sparse|allgather
 * overrides (same value on every rank).
 * out = {m-vectors sparse?, entries sent, received, n-vectors sparse?, sent, received, all ranks' requests m, n} */
int hprlp_solver_dist_info(hprlp_solver *s, long out[8]);
/* What the transport itself reports: out = {ranks, this rank, device of the main communicator (RCCL: ncclCommCount,
 * ncclCommUserRank, ncclCommCuDevice), the same three of the exchange stream's communicator (0, -1, -1 without one),
 * the solver's HIP device, 1 if exchanges overlap the half-steps}.  bench.py prints these as rccl_ranks / devices. */
int hprlp_solver_dist_comm_info(hprlp_solver *s, long out[8]);
/* Test hook: one grouped send/recv of `count` doubles from this rank to itself through the solver's communicator,
 * verified on the host (0 = intact).  Exercises the point-to-point transport calls where no second rank exists. */
int hprlp_solver_dist_loopback(hprlp_solver *s, int count);
/* The same multi-rank solver with `size` ranks as host THREADS of one process on one GPU, exchanging through
 * device copies and host barriers instead of RCCL: lets the sharded path run on a one-GPU box (tests). Every
 * rank's thread must make the same sequence of solver calls. */
typedef struct hprlp_local_group hprlp_local_group;
hprlp_local_group *hprlp_local_group_create(int size);
void hprlp_local_group_destroy(hprlp_local_group *g);
hprlp_solver *hprlp_solver_create_local(const LP_info_cpu *model, const HPRLP_parameters *param, int rank, int size,
                                        hprlp_local_group *group);
hprlp_solver *hprlp_solver_create_local_from_shard(const hprlp_shard *shard, const HPRLP_parameters *param, int rank, int size,
                                                   hprlp_local_group *group);

/* ---- presolve / postsolve as separate host-side steps (what solve() does around the iteration when
 * use_presolve is set; replaces the reference's forked PSLP worker, src/pslp_integration.cpp:628-787).
 * hprlp_presolve_run returns NULL when the model is left unchanged or looks infeasible/unbounded.
 * LIFETIME: the handle keeps a pointer to `model` (postsolve reads the original rows and costs); the model must stay
 * alive and unchanged until hprlp_presolve_free(). */
typedef struct hprlp_presolve hprlp_presolve;
hprlp_presolve *hprlp_presolve_run(const LP_info_cpu *model);
const LP_info_cpu *hprlp_presolve_reduced(const hprlp_presolve *p); /* owned by p */
/* out = {reduced m, reduced n, fixed cols, empty cols, singleton rows, empty rows, redundant rows, passes,
 *        dual-fixed cols, slack cols, parallel rows, parallel cols, forcing rows, doubleton rows, tightened bounds,
 *        rounds of the chain} */
int hprlp_presolve_stats(const hprlp_presolve *p, int out[16]);
/* (xr, yr, zr) of the reduced model -> (x, y, z) in the original dimensions */
int hprlp_presolve_postsolve(const hprlp_presolve *p, const double *xr, const double *yr, const double *zr, double *x,
                             double *y, double *z);
void hprlp_presolve_free(hprlp_presolve *p);
/* out = {primal infeasibility, dual infeasibility, gap (all relative), primal objective, dual objective} on the
 * model as given (reference compute_original_kkt_metrics, src/pslp_integration.cpp:499-580) */
int hprlp_original_kkt(const LP_info_cpu *model, const double *x, const double *y, const double *z, double out[5]);

/* Set-up time locality ordering (host; hpr-lp-c_amd/csrc/reorder.cpp): row / column permutations (new -> old) that make
 * the pattern band-like so that the column-tiled kernels apply; the solver runs it by itself when a large matrix
 * fails the tiling test in its given order (HPRLP_NO_REORDER=1 disables) and returns x, y, z in the caller's numbering
 * like the reference's collect_solution (src/utils.cu:143-200).  out = {accepted, tiled share before, after, clusters,
 * components, seconds}. */
int hprlp_locality_ordering(int m, int n, const int *rowptr, const int *col, int *row_new2old, int *col_new2old, double out[6]);
/* host only: the stream kernel's row blocks of a CSR pattern, built and checked as a solver's set-up does (dense rows cut by
 * column eighths when with_cuts != 0); out = {blocks, split rows, chunk slots, rows cut, longest chunk, entries covered};
 * -1 + hprlp_last_error() when the block list would be refused. */
int hprlp_row_block_plan(int m, int n, const int *rowptr, const int *col, int with_cuts, long out[6]);
/* host only: the column-tiled copy of a CSR pattern as the host builder lays it out (super-blocks of R rows: a multiple of 64 up to
 * 8192; tiles of T = 2048 or 1024 columns; min_dense: least share of the entries in staged tiles), verified entry by entry -- every
 * entry exactly once, codes name their entries, one chunk per accumulator inside a step, at most four layers per tile.
 * out = {tile entries incl. padding, remainder entries, steps, padding, most consecutive steps of one tile, staged share x 1e6};
 * -1 + hprlp_last_error() on the first violation or when the build declines the pattern. */
int hprlp_tiled_host_check(int m, int n, const int *rowptr, const int *col, int R, int T, double min_dense, long out[6]);

#ifdef __cplusplus
}
#endif
#endif /* HPRLP_AMD_H */
