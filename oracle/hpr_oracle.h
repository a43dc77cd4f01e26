/*
 * hpr_oracle.h -- CPU restatement of the HPR-LP main iteration (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle and the timed CPU baseline ("port") of the hot path named by
 * BASELINE.json:north_star.  It is NOT part of the product: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The shipped library (lib/libhprlp.so) never links,
 * loads or calls anything in this directory.
 *
 * Pinning status: the reference ships no tests and no golden vectors for this path.  The only
 * known answer in the reference tree is data/model.mps => x=(2.8,3.6), obj=-26.4
 * (reference examples/cpp/example_direct_lp.cpp:14); the oracle is checked against it and against
 * independent LP optima (HiGHS via scipy) in tests/test_oracle.py.  At the vendor-library boundary
 * (cuSPARSE SpMV/SpMM, cuBLAS dot/nrm2, cuRAND normal) parity is UNPINNED: those libraries are not
 * in /root/reference and no reference test fixes their rounding; they are restated here as the plain
 * mathematical operations (sequential CSR row dot, sequential dot, sqrt of sum of squares, our own
 * counter RNG).
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 */
#ifndef HPR_ORACLE_H
#define HPR_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int max_iter;
    double stop_tol;
    double time_limit;
    int check_iter;
    int use_CR_scaling;
    int use_Ruiz_scaling;
    int use_Pock_Chambolle_scaling;
    int use_bc_scaling;
} orc_params;

typedef struct {
    double b_scale, c_scale, norm_b, norm_c, norm_b_org, norm_c_org;
} orc_scaling_scalars;

/* One row per residual evaluation (reference log line, src/HPRLP.cu:207-218). */
typedef struct {
    int iter;
    int restart_flag;
    double err_Rp, err_Rd, primal_obj, dual_obj, gap, kkt, sigma, current_gap, lambda_max;
    /* what the restart test of this row compares current_gap with (src/main_iterate.cu:341-351), as they stand BEFORE the test:
     * the weighted norm after the last restart, the previous check's gap, inner iterations since the restart -- the decision
     * margins of tests/fuzz_parity.py: acceptable() */
    double last_gap, save_gap, inner;
} orc_trace_row;

typedef struct {
    double residuals, primal_obj, gap;
    double time4, time6, time8, time;
    int iter4, iter6, iter8, iter;
    char status[64];
    double lambda_max;      /* final lambda_max (may have been bumped) */
    double power_time;
    int power_iters;
    int n_trace;
    int n_restarts;
} orc_result;

/* src/utils.cu:203-232 */
void orc_csr_transpose(int rows, int cols, int nnz, const int *rp, const int *ci, const double *v,
                       int *trp, int *tci, double *tv);

/* y = M x, sequential CSR row dot (cusparseSpMV call sites, src/main_iterate.cu:423-471) */
void orc_spmv(int rows, const int *rp, const int *ci, const double *v, const double *x, double *y);

/* deterministic start vector for the power iteration: z_i = N(0,1)_i + 1e-8
 * (src/power_iteration.cu:44-57; cuRAND replaced by a documented counter RNG) */
void orc_power_start_vector(int m, unsigned long long seed, long long offset, double *z);

/* src/scaling.cu:88-216; in-place on A, AT, AL, AU, l, u, c */
void orc_scaling(int m, int n, const int *Arp, const int *Aci, double *Av, const int *ATrp,
                 const int *ATci, double *ATv, double *AL, double *AU, double *l, double *u, double *c,
                 const orc_params *p, double *row_norm, double *col_norm, orc_scaling_scalars *out);

/* src/power_iteration.cu:20-119 (returns lambda, not yet multiplied by 1.01) */
double orc_power_iteration(int m, int n, const int *Arp, const int *Aci, const double *Av,
                           const int *ATrp, const int *ATci, const double *ATv, const double *z0,
                           int max_iter, double tol, int *iters_out);

/* src/cuda_kernels/HPR_cuda_kernels.cu:203-247 ; check!=0 also writes x_bar,z_bar,x_temp */
void orc_x_half(int n, const int *ATrp, const int *ATci, const double *ATv, const double *y, double *x,
                double *x_hat, double *x_bar, double *z_bar, double *x_temp, const double *l,
                const double *u, const double *c, const double *last_x, double sigma, int k, int check);

/* src/cuda_kernels/HPR_cuda_kernels.cu:249-295 ; check!=0 also writes y_bar,y_obj,y_temp */
void orc_y_half(int m, const int *Arp, const int *Aci, const double *Av, const double *x_hat, double *y,
                double *y_bar, double *y_obj, double *y_temp, const double *AL, const double *AU,
                const double *last_y, double sigma, double lambda_max, int k, int check);

/* Full single-LP solve: src/HPRLP.cu:116-311 with src/main_iterate.cu:229-515.
 * Input is the UNSCALED model (CSR A, bounds, cost); x,y,z are caller-allocated (n,m,n).
 * lambda_override>0 skips the power iteration and uses that value as lambda_max.
 * trace may be NULL. */
int orc_solve(int m, int n, int nnz, const int *Arp, const int *Aci, const double *Av, const double *AL,
              const double *AU, const double *l, const double *u, const double *c, double obj_constant,
              const orc_params *p, double lambda_override, double *x, double *y, double *z,
              orc_result *res, orc_trace_row *trace, int max_trace);

/* Batched shared-A solve: src/batched_solver.cu:939-1092.  Panels column-major as in the ABI. */
int orc_solve_batched(int m, int n, int nnz, const int *Arp, const int *Aci, const double *Av, int B,
                      const double *C, const double *AL, const double *AU, const double *L,
                      const double *U, const double *obj_constants, double model_obj_constant,
                      const orc_params *p, double lambda_override, double *X, double *Y, double *Z,
                      double *primal_obj, double *residuals, double *gap, int *iter, char *status,
                      double *lambda_out);

/* Timed leg for bench.py cpu_baseline: run `iters` normal HPR iterations (no checks) on an
 * already scaled problem; returns seconds. */
double orc_time_iterations(int m, int n, const int *Arp, const int *Aci, const double *Av,
                           const int *ATrp, const int *ATci, const double *ATv, const double *AL,
                           const double *AU, const double *l, const double *u, const double *c,
                           double sigma, double lambda_max, int iters, double *x, double *y);

int orc_num_threads(void);
void orc_set_num_threads(int n); /* OpenMP team size of the timed loops (bench.py: the process' CPU share) */

#ifdef __cplusplus
}
#endif
#endif
