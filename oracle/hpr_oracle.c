/*
 * hpr_oracle.c -- CPU restatement of the HPR-LP main iteration.  TEST INFRASTRUCTURE ONLY:
 * see hpr_oracle.h for the rules (never linked or loaded by the product) and the pinning status
 * ("parity unpinned" at the vendor-library boundary; one known answer from the reference).
 *
 * Compile with -ffp-contract=off so that `s += a*b` is a rounded product followed by a rounded add,
 * which is what the HIP kernels do for stream-mode rows (product staged through LDS, then summed in
 * CSR order).  OpenMP only parallelises loops whose result does not depend on the thread count,
 * except the dot/norm reductions (used for scalars that are compared with a tolerance anyway).
 */
#include "hpr_oracle.h"

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_INF (1.0 / 0.0)

static double now_sec(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ---------------------------------------------------------------- small vector helpers -------- */

static double dotv(const double *a, const double *b, long n) {
    double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static) if (n > 100000)
    for (long i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}
static double nrm2v(const double *a, long n) { return sqrt(dotv(a, a, n)); }

/* src/cuda_kernels/HPR_cuda_kernels.cu:34-43 (conceptual_b_kernel) followed by l2_norm */
static double conceptual_b_norm(const double *AL, const double *AU, long m) {
    double s = 0.0;
    for (long i = 0; i < m; ++i) {
        double a = isinf(AL[i]) ? 0.0 : AL[i];
        double b = isinf(AU[i]) ? 0.0 : AU[i];
        double v = fmax(fabs(a), fabs(b));
        s += v * v;
    }
    return sqrt(s);
}

/* src/utils.cu:100-102 */
static int step_of(int iter) {
    double p = pow(10.0, floor(log10((double)iter)));
    int v = (int)(p / 10.0);
    return v > 10 ? v : 10;
}

/* ---------------------------------------------------------------- CSR helpers ----------------- */

/* src/utils.cu:203-232: counting sort by column, stable in row order */
void orc_csr_transpose(int rows, int cols, int nnz, const int *rp, const int *ci, const double *v,
                       int *trp, int *tci, double *tv) {
    int *next = (int *)calloc((size_t)cols + 1, sizeof(int));
    for (int i = 0; i <= cols; ++i) trp[i] = 0;
    for (int k = 0; k < nnz; ++k) trp[ci[k] + 1]++;
    for (int j = 0; j < cols; ++j) trp[j + 1] += trp[j];
    for (int j = 0; j < cols; ++j) next[j] = trp[j];
    for (int i = 0; i < rows; ++i)
        for (int k = rp[i]; k < rp[i + 1]; ++k) {
            int j = ci[k];
            int pos = next[j]++;
            tv[pos] = v[k];
            tci[pos] = i;
        }
    free(next);
}

void orc_spmv(int rows, const int *rp, const int *ci, const double *v, const double *x, double *y) {
#pragma omp parallel for schedule(static) if (rows > 20000)
    for (int i = 0; i < rows; ++i) {
        double s = 0.0;
        for (int k = rp[i]; k < rp[i + 1]; ++k) s += v[k] * x[ci[k]];
        y[i] = s;
    }
}

/* ---------------------------------------------------------------- RNG -------------------------- */

static uint64_t splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

/* src/power_iteration.cu:44-57.  cuRAND XORWOW(seed=1) is not reproducible without cuRAND; the
 * specification used by both the oracle and the product is: element i (global row index) takes
 * h1 = splitmix64(seed*0x100000001B3 + 2i), h2 = splitmix64(seed*0x100000001B3 + 2i + 1),
 * u1 = ((h1>>11)+1)*2^-53 in (0,1], u2 = (h2>>11)*2^-53 in [0,1),
 * z_i = sqrt(-2 ln u1) * cos(2 pi u2) + 1e-8. */
void orc_power_start_vector(int m, unsigned long long seed, long long offset, double *z) {
    const double two53 = 1.0 / 9007199254740992.0;
    const double twopi = 6.283185307179586476925286766559;
    uint64_t base = (uint64_t)seed * 0x100000001B3ULL;
    for (int i = 0; i < m; ++i) {
        uint64_t g = (uint64_t)(offset + i);
        uint64_t h1 = splitmix64(base + 2 * g);
        uint64_t h2 = splitmix64(base + 2 * g + 1);
        double u1 = (double)((h1 >> 11) + 1) * two53;
        double u2 = (double)(h2 >> 11) * two53;
        z[i] = sqrt(-2.0 * log(u1)) * cos(twopi * u2) + 1e-8;
    }
}

/* ---------------------------------------------------------------- scaling ---------------------- */

/* src/scaling.cu:5-31 */
static void cr_log_update(int rows, const int *rp, const int *ci, const double *v, const double *other,
                          double *result) {
    for (int r = 0; r < rows; ++r) {
        int s = rp[r], e = rp[r + 1];
        if (e - s <= 0) {
            result[r] = 0.0;
            continue;
        }
        double sum = 0.0;
        for (int k = s; k < e; ++k) {
            double a = fmax(fabs(v[k]), 1e-300);
            sum += -log(a) - other[ci[k]];
        }
        result[r] = sum / (double)(e - s);
    }
}

/* src/cuda_kernels/HPR_cuda_kernels.cu:91-120 */
static void csr_row_norm(int rows, const int *rp, const double *v, double *result, int norm) {
    for (int i = 0; i < rows; ++i) {
        double r = 0.0;
        if (norm == 99) {
            for (int k = rp[i]; k < rp[i + 1]; ++k)
                if (r < fabs(v[k])) r = fabs(v[k]);
        } else {
            for (int k = rp[i]; k < rp[i + 1]; ++k) r += fabs(v[k]);
        }
        r = sqrt(r);
        if (r < 1e-15) r = 1.0;
        result[i] = r;
    }
}

/* src/cuda_kernels/HPR_cuda_kernels.cu:122-157 */
static void mul_rows(int rows, const int *rp, double *v, const double *s, int divide) {
    for (int i = 0; i < rows; ++i)
        for (int k = rp[i]; k < rp[i + 1]; ++k) v[k] = divide ? v[k] / s[i] : v[k] * s[i];
}
static void mul_cols(int rows, const int *rp, const int *ci, double *v, const double *s, int divide) {
    for (int i = 0; i < rows; ++i)
        for (int k = rp[i]; k < rp[i + 1]; ++k) v[k] = divide ? v[k] / s[ci[k]] : v[k] * s[ci[k]];
}
static void vmul(double *x, const double *s, long n, int divide) {
    for (long i = 0; i < n; ++i) x[i] = divide ? x[i] / s[i] : x[i] * s[i];
}

/* src/scaling.cu:88-216 */
void orc_scaling(int m, int n, const int *Arp, const int *Aci, double *Av, const int *ATrp,
                 const int *ATci, double *ATv, double *AL, double *AU, double *l, double *u, double *c,
                 const orc_params *p, double *row_norm, double *col_norm, orc_scaling_scalars *out) {
    double *t1 = (double *)calloc((size_t)m, sizeof(double));
    double *t2 = (double *)calloc((size_t)n, sizeof(double));
    for (int i = 0; i < m; ++i) row_norm[i] = 1.0;
    for (int j = 0; j < n; ++j) col_norm[j] = 1.0;

    out->norm_b_org = 1.0 + conceptual_b_norm(AL, AU, m); /* :114-116 */
    out->norm_c_org = 1.0 + nrm2v(c, n);                  /* :117 */

    if (p->use_CR_scaling) { /* :40-83 */
        for (int it = 0; it < 20; ++it) {
            cr_log_update(m, Arp, Aci, Av, t2, t1);
            cr_log_update(n, ATrp, ATci, ATv, t1, t2);
        }
        for (int i = 0; i < m; ++i) t1[i] = fmin(fmax(exp(t1[i]), 1e-30), 1e30);
        for (int j = 0; j < n; ++j) t2[j] = fmin(fmax(exp(t2[j]), 1e-30), 1e30);
        vmul(row_norm, t1, m, 1);
        vmul(col_norm, t2, n, 1);
        mul_rows(m, Arp, Av, t1, 0);
        mul_cols(n, ATrp, ATci, ATv, t1, 0);
        mul_rows(n, ATrp, ATv, t2, 0);
        mul_cols(m, Arp, Aci, Av, t2, 0);
        vmul(AL, t1, m, 0);
        vmul(AU, t1, m, 0);
        vmul(c, t2, n, 0);
        vmul(l, t2, n, 1);
        vmul(u, t2, n, 1);
    }

    int passes = (p->use_Ruiz_scaling ? 10 : 0) + (p->use_Pock_Chambolle_scaling ? 1 : 0);
    for (int it = 0; it < passes; ++it) { /* Ruiz :123-153, then Pock-Chambolle :157-183 */
        int norm = (p->use_Ruiz_scaling && it < 10) ? 99 : 1;
        csr_row_norm(m, Arp, Av, t1, norm);
        vmul(row_norm, t1, m, 0);
        vmul(AL, t1, m, 1);
        vmul(AU, t1, m, 1);
        csr_row_norm(n, ATrp, ATv, t2, norm);
        vmul(col_norm, t2, n, 0);
        mul_rows(m, Arp, Av, t1, 1);
        mul_cols(n, ATrp, ATci, ATv, t1, 1);
        mul_rows(n, ATrp, ATv, t2, 1);
        mul_cols(m, Arp, Aci, Av, t2, 1);
        vmul(c, t2, n, 1);
        vmul(l, t2, n, 0);
        vmul(u, t2, n, 0);
    }

    if (p->use_bc_scaling) { /* :185-202 (cublasDscal by the reciprocal) */
        out->b_scale = 1.0 + conceptual_b_norm(AL, AU, m);
        out->c_scale = 1.0 + nrm2v(c, n);
        const double bs = 1.0 / out->b_scale, cs = 1.0 / out->c_scale;
        for (int i = 0; i < m; ++i) {
            AU[i] *= bs;
            AL[i] *= bs;
        }
        for (int j = 0; j < n; ++j) {
            l[j] *= bs;
            u[j] *= bs;
            c[j] *= cs;
        }
    } else {
        out->b_scale = 1.0;
        out->c_scale = 1.0;
    }
    out->norm_b = conceptual_b_norm(AL, AU, m); /* :206-211 */
    out->norm_c = nrm2v(c, n);
    free(t1);
    free(t2);
}

/* ---------------------------------------------------------------- power iteration -------------- */

/* src/power_iteration.cu:20-119 */
double orc_power_iteration(int m, int n, const int *Arp, const int *Aci, const double *Av,
                           const int *ATrp, const int *ATci, const double *ATv, const double *z0,
                           int max_iter, double tol, int *iters_out) {
    double *z = (double *)malloc((size_t)m * sizeof(double));
    double *q = (double *)malloc((size_t)m * sizeof(double));
    double *ATq = (double *)malloc((size_t)n * sizeof(double));
    memcpy(z, z0, (size_t)m * sizeof(double));
    double lambda = 1.0;
    int it_done = max_iter;
    for (int i = 1; i <= max_iter; ++i) {
        double z2 = dotv(z, z, m);
        double invn = 1.0 / sqrt(z2 + 2.220446049250313e-16);
        for (int r = 0; r < m; ++r) q[r] = invn * z[r];
        orc_spmv(n, ATrp, ATci, ATv, q, ATq);
        orc_spmv(m, Arp, Aci, Av, ATq, z);
        if (i % 10 == 0) {
            lambda = dotv(q, z, m);
            for (int r = 0; r < m; ++r) q[r] = -lambda * q[r] + 1.0 * z[r]; /* axpby(-lambda,q,1,z,q) */
            double err = nrm2v(q, m);
            if (err < tol) {
                it_done = i;
                break;
            }
        }
    }
    if (iters_out) *iters_out = it_done;
    free(z);
    free(q);
    free(ATq);
    return lambda;
}

/* ---------------------------------------------------------------- one HPR step ----------------- */

/* src/cuda_kernels/HPR_cuda_kernels.cu:203-247; Halpern factors from :192-200 with inner = k */
void orc_x_half(int n, const int *ATrp, const int *ATci, const double *ATv, const double *y, double *x,
                double *x_hat, double *x_bar, double *z_bar, double *x_temp, const double *l,
                const double *u, const double *c, const double *last_x, double sigma, int k, int check) {
    const double f1 = 1.0 / ((double)k + 2.0);
    const double f2 = 1.0 - f1;
#pragma omp parallel for schedule(static) if (n > 20000)
    for (int j = 0; j < n; ++j) {
        double g = 0.0;
        for (int p = ATrp[j]; p < ATrp[j + 1]; ++p) g += ATv[p] * y[ATci[p]];
        double xi = x[j];
        double gc = g - c[j];
        double zt = xi + sigma * gc;
        double xb = fmin(u[j], fmax(l[j], zt));
        double xh = 2.0 * xb - xi;
        double xn = f2 * xh + f1 * last_x[j];
        if (check) {
            z_bar[j] = (xb - zt) / sigma;
            x_bar[j] = xb;
            x_temp[j] = xb - xh;
        }
        x_hat[j] = xh;
        x[j] = xn;
    }
}

/* src/cuda_kernels/HPR_cuda_kernels.cu:249-295 */
void orc_y_half(int m, const int *Arp, const int *Aci, const double *Av, const double *x_hat, double *y,
                double *y_bar, double *y_obj, double *y_temp, const double *AL, const double *AU,
                const double *last_y, double sigma, double lambda_max, int k, int check) {
    const double hf1 = 1.0 / ((double)k + 2.0);
    const double hf2 = 1.0 - hf1;
    const double fact1 = lambda_max * sigma; /* src/main_iterate.cu:17-52 */
    const double fact2 = 1.0 / fact1;
#pragma omp parallel for schedule(static) if (m > 20000)
    for (int i = 0; i < m; ++i) {
        double h = 0.0;
        for (int p = Arp[i]; p < Arp[i + 1]; ++p) h += Av[p] * x_hat[Aci[p]];
        double yi = y[i];
        double v = h - fact1 * yi;
        double d = fmax(AL[i] - v, fmin(AU[i] - v, 0.0));
        double yb = fact2 * d;
        double yh = 2.0 * yb - yi;
        double yn = hf2 * yh + hf1 * last_y[i];
        if (check) {
            y_temp[i] = yb - yh;
            y_bar[i] = yb;
            y_obj[i] = v + d;
        }
        y[i] = yn;
    }
}

/* ---------------------------------------------------------------- single-LP solve -------------- */

typedef struct {
    int m, n;
    const int *Arp, *Aci, *ATrp, *ATci;
    double *Av, *ATv;
    double *AL, *AU, *l, *u, *c;
    double *row_norm, *col_norm;
    orc_scaling_scalars sc;
    double obj_constant;
    double *x, *last_x, *x_hat, *x_bar, *z_bar, *x_temp;
    double *y, *last_y, *y_bar, *y_obj, *y_temp;
    double *Ax, *ATy; /* scratch, sizes m / n */
    double sigma, lambda_max;
    int k; /* device-side Halpern inner counter */
} orc_ws;

typedef struct {
    double err_Rp, err_Rd, pobj, dobj, gap, kkt;
} orc_resid;

typedef struct {
    int flag, first, inner, times;
    double last_gap, current_gap, save_gap, best_gap, best_sigma;
} orc_restart;

/* src/main_iterate.cu:486-515 and the same formula inline at :292-308 */
static double weighted_norm_from(orc_ws *w, double dot_Adx_dy, double dy2, double dx2) {
    double dot_prod = 2.0 * dot_Adx_dy;
    double wn = w->sigma * (w->lambda_max * dy2) + dx2 / w->sigma + dot_prod;
    if (wn < 0) {
        w->lambda_max = -(dot_prod + dx2 / w->sigma) / (w->sigma * dy2) * 1.05;
        wn = sqrt(-(dot_prod + dx2 / w->sigma) * 0.05);
    } else {
        wn = sqrt(wn);
    }
    return wn;
}

static double compute_weighted_norm(orc_ws *w) {
    orc_spmv(w->m, w->Arp, w->Aci, w->Av, w->x_temp, w->Ax);
    double d0 = dotv(w->Ax, w->y_temp, w->m);
    double d1 = dotv(w->y_temp, w->y_temp, w->m);
    double d2 = dotv(w->x_temp, w->x_temp, w->n);
    return weighted_norm_from(w, d0, d1, d2);
}

/* src/main_iterate.cu:229-309 */
static void compute_residuals(orc_ws *w, orc_resid *r, int iter, orc_restart *rs, int compute_gap) {
    const int m = w->m, n = w->n;
    const double obj_scale = w->sc.b_scale * w->sc.c_scale;
    double s0 = dotv(w->c, w->x_bar, n);
    double s1 = dotv(w->y_obj, w->y_bar, m);
    double s2 = dotv(w->x_bar, w->z_bar, n);
    double s5 = 0, s6 = 0, s7 = 0;
    if (compute_gap) {
        orc_spmv(m, w->Arp, w->Aci, w->Av, w->x_temp, w->Ax);
        s5 = dotv(w->Ax, w->y_temp, m);
        s6 = dotv(w->y_temp, w->y_temp, m);
        s7 = dotv(w->x_temp, w->x_temp, n);
    }
    /* Rd: :217-226 + kernel HPR_cuda_kernels.cu:183-189 */
    orc_spmv(n, w->ATrp, w->ATci, w->ATv, w->y_bar, w->ATy);
    double rd2 = 0.0;
    for (int j = 0; j < n; ++j) {
        double rd = (w->c[j] - w->ATy[j] - w->z_bar[j]) * w->col_norm[j];
        rd2 += rd * rd;
    }
    /* Rp: :207-215 + kernel :160-172 */
    orc_spmv(m, w->Arp, w->Aci, w->Av, w->x_bar, w->Ax);
    double rp2 = 0.0;
    for (int i = 0; i < m; ++i) {
        double v = w->Ax[i];
        double rp = fmax(fmin(w->AU[i] - v, 0.0), w->AL[i] - v) * w->row_norm[i];
        rp2 += rp * rp;
    }
    r->pobj = obj_scale * s0 + w->obj_constant;
    r->dobj = obj_scale * (s1 + s2) + w->obj_constant;
    r->gap = fabs(r->pobj - r->dobj) / (1.0 + fabs(r->pobj) + fabs(r->dobj));
    r->err_Rd = w->sc.c_scale * sqrt(rd2) / w->sc.norm_c_org;
    r->err_Rp = w->sc.b_scale * sqrt(rp2) / w->sc.norm_b_org;
    if (iter == 0) { /* :264-267,285-289 + kernel :174-180 (overwrites x_temp) */
        double lu2 = 0.0;
        for (int j = 0; j < n; ++j) {
            double xb = w->x_bar[j];
            double t = (xb < w->l[j]) ? (w->l[j] - xb) : ((xb > w->u[j]) ? (xb - w->u[j]) : 0.0);
            w->x_temp[j] = t / w->col_norm[j];
            lu2 += w->x_temp[j] * w->x_temp[j];
        }
        r->err_Rp = fmax(r->err_Rp, w->sc.b_scale * sqrt(lu2));
    }
    r->kkt = fmax(fmax(r->err_Rd, r->err_Rp), r->gap);
    if (compute_gap && rs) rs->current_gap = weighted_norm_from(w, s5, s6, s7);
}

/* src/main_iterate.cu:324-364 */
static void check_restart(orc_restart *rs, int iter, int check_iter, double sigma) {
    rs->flag = 0;
    if (rs->first) {
        if (iter == check_iter) {
            rs->first = 0;
            rs->flag = 1;
            rs->best_gap = rs->current_gap;
            rs->best_sigma = sigma;
        }
    } else if (iter % check_iter == 0) {
        if (rs->current_gap < 0) rs->current_gap = 1e-6;
        if (rs->current_gap <= 0.2 * rs->last_gap) rs->flag = 1;
        if (rs->current_gap <= 0.6 * rs->last_gap && rs->current_gap > 1.00 * rs->save_gap) rs->flag = 2;
        if (rs->inner >= 0.2 * iter) rs->flag = 3;
        if (rs->best_gap > rs->current_gap) {
            rs->best_gap = rs->current_gap;
            rs->best_sigma = sigma;
        }
        rs->save_gap = rs->current_gap;
    }
}

/* src/main_iterate.cu:367-404; returns the new sigma given the movement norms */
static double sigma_formula(double primal_move, double dual_move, double lambda_max, double current_gap,
                            double best_gap, double best_sigma, double err_Rd, double err_Rp, double gap) {
    if (primal_move > 1e-16 && dual_move > 1e-16 && primal_move < 1e12 && dual_move < 1e12) {
        double ratio = (primal_move / dual_move) / sqrt(lambda_max);
        double fact = exp(-0.05 * (current_gap / best_gap));
        double temp1 = fmax(fmin(err_Rd, err_Rp), fmin(gap, current_gap));
        double sigma_cand = exp(fact * log(ratio) + (1 - fact) * log(best_sigma));
        double kappa;
        if (temp1 > 9e-10) {
            kappa = 1.0;
        } else if (temp1 > 5e-10) {
            kappa = fmax(fmin(sqrt(err_Rd / err_Rp), 100.0), 1e-2);
        } else {
            kappa = fmax(fmin(err_Rd / err_Rp, 100.0), 1e-2);
        }
        return kappa * sigma_cand;
    }
    return 1.0;
}

static void update_sigma(orc_restart *rs, orc_ws *w, const orc_resid *r) {
    if (rs->flag <= 0) return;
    for (int j = 0; j < w->n; ++j) w->x_temp[j] = 1.0 * w->x_bar[j] + (-1.0) * w->last_x[j];
    for (int i = 0; i < w->m; ++i) w->y_temp[i] = 1.0 * w->y_bar[i] + (-1.0) * w->last_y[i];
    double pm = nrm2v(w->x_temp, w->n), dm = nrm2v(w->y_temp, w->m);
    w->sigma = sigma_formula(pm, dm, w->lambda_max, rs->current_gap, rs->best_gap, rs->best_sigma,
                             r->err_Rd, r->err_Rp, r->gap);
}

/* src/main_iterate.cu:312-322 + Halpern reset :54-66 */
static void do_restart(orc_ws *w, orc_restart *rs) {
    if (rs->flag <= 0) return;
    memcpy(w->last_x, w->x_bar, (size_t)w->n * sizeof(double));
    memcpy(w->last_y, w->y_bar, (size_t)w->m * sizeof(double));
    memcpy(w->x, w->x_bar, (size_t)w->n * sizeof(double));
    memcpy(w->y, w->y_bar, (size_t)w->m * sizeof(double));
    rs->inner = 0;
    rs->times += 1;
    rs->save_gap = ORC_INF;
    w->k = 0;
}

static double *dup_vec(const double *src, long n) {
    double *d = (double *)malloc((size_t)(n > 0 ? n : 1) * sizeof(double));
    memcpy(d, src, (size_t)n * sizeof(double));
    return d;
}
static double *zeros(long n) { return (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double)); }

/* src/HPRLP.cu:116-311 */
int orc_solve(int m, int n, int nnz, const int *Arp, const int *Aci, const double *Av_in,
              const double *AL_in, const double *AU_in, const double *l_in, const double *u_in,
              const double *c_in, double obj_constant, const orc_params *p, double lambda_override,
              double *x_out, double *y_out, double *z_out, orc_result *res, orc_trace_row *trace,
              int max_trace) {
    orc_ws w;
    memset(&w, 0, sizeof(w));
    w.m = m;
    w.n = n;
    w.obj_constant = obj_constant;
    /* src/preprocess.cu:66-101: device copy of A and host-built AT */
    int *ATrp = (int *)malloc(((size_t)n + 1) * sizeof(int));
    int *ATci = (int *)malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(int));
    w.ATv = (double *)malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(double));
    orc_csr_transpose(m, n, nnz, Arp, Aci, Av_in, ATrp, ATci, w.ATv);
    w.Arp = Arp;
    w.Aci = Aci;
    w.ATrp = ATrp;
    w.ATci = ATci;
    w.Av = dup_vec(Av_in, nnz);
    w.AL = dup_vec(AL_in, m);
    w.AU = dup_vec(AU_in, m);
    w.l = dup_vec(l_in, n);
    w.u = dup_vec(u_in, n);
    w.c = dup_vec(c_in, n);
    w.row_norm = zeros(m);
    w.col_norm = zeros(n);
    w.x = zeros(n); w.last_x = zeros(n); w.x_hat = zeros(n); w.x_bar = zeros(n);
    w.z_bar = zeros(n); w.x_temp = zeros(n); w.ATy = zeros(n);
    w.y = zeros(m); w.last_y = zeros(m); w.y_bar = zeros(m); w.y_obj = zeros(m);
    w.y_temp = zeros(m); w.Ax = zeros(m);

    orc_scaling(m, n, w.Arp, w.Aci, w.Av, w.ATrp, w.ATci, w.ATv, w.AL, w.AU, w.l, w.u, w.c, p,
                w.row_norm, w.col_norm, &w.sc);

    const double t_start = now_sec(); /* t_start_alg, src/HPRLP.cu:150 */
    memset(res, 0, sizeof(*res));
    if (lambda_override > 0) {
        w.lambda_max = lambda_override;
    } else { /* src/HPRLP.cu:81-97 */
        double *z0 = (double *)malloc((size_t)m * sizeof(double));
        orc_power_start_vector(m, 1ULL, 0, z0);
        w.lambda_max = orc_power_iteration(m, n, w.Arp, w.Aci, w.Av, w.ATrp, w.ATci, w.ATv, z0, 5000,
                                           1e-4, &res->power_iters) * 1.01;
        free(z0);
    }
    res->power_time = now_sec() - t_start;

    w.sigma = (w.sc.norm_b > 1e-8 && w.sc.norm_c > 1e-8) ? w.sc.norm_b / w.sc.norm_c : 1.0; /* :156-161 */
    orc_restart rs;
    memset(&rs, 0, sizeof(rs));
    rs.first = 1;
    rs.last_gap = rs.current_gap = rs.save_gap = rs.best_gap = ORC_INF;
    rs.best_sigma = w.sigma;
    orc_resid r;
    memset(&r, 0, sizeof(r));
    r.kkt = ORC_INF;
    int first4 = 1, first6 = 1, first8 = 1, ntrace = 0;
    const char *status = "CONTINUE";
    int iter = 0;
    for (;; ++iter) {
        const int at_limit = (iter >= p->max_iter);
        int periodic = (iter % p->check_iter == 0);
        int compute_gap = (periodic && iter > 0);
        double elapsed = now_sec() - t_start;
        int print_flag = (iter % step_of(iter) == 0) || at_limit || (elapsed > p->time_limit);
        if (periodic || print_flag) {
            compute_residuals(&w, &r, iter, &rs, compute_gap);
            if (trace && ntrace < max_trace) {
                orc_trace_row *t = &trace[ntrace];
                t->iter = iter; t->restart_flag = 0;
                t->err_Rp = r.err_Rp; t->err_Rd = r.err_Rd; t->primal_obj = r.pobj; t->dual_obj = r.dobj;
                t->gap = r.gap; t->kkt = r.kkt; t->sigma = w.sigma; t->current_gap = rs.current_gap;
                t->lambda_max = w.lambda_max;
                t->last_gap = rs.last_gap; t->save_gap = rs.save_gap; t->inner = (double)rs.inner;
            }
        }
        /* check_stopping, src/main_iterate.cu:406-420 (max_iter: see DESIGN.md, reference is UB) */
        if (r.kkt < p->stop_tol) status = "OPTIMAL";
        else if (at_limit) status = "ITER_LIMIT";
        else if (now_sec() - t_start > p->time_limit) status = "TIME_LIMIT";
        if (periodic && !at_limit) check_restart(&rs, iter, p->check_iter, w.sigma);
        else rs.flag = 0;
        if ((periodic || print_flag) && trace && ntrace < max_trace) {
            trace[ntrace].restart_flag = rs.flag;
            ntrace++;
        }
        if (first4 && r.kkt < 1e-4) { res->iter4 = iter; res->time4 = now_sec() - t_start; first4 = 0; }
        if (first6 && r.kkt < 1e-6) { res->iter6 = iter; res->time6 = now_sec() - t_start; first6 = 0; }
        if (first8 && r.kkt < 1e-8) { res->iter8 = iter; res->time8 = now_sec() - t_start; first8 = 0; }
        if (strcmp(status, "CONTINUE") != 0) break;

        update_sigma(&rs, &w, &r);
        do_restart(&w, &rs);
        int check = ((iter + 1) % p->check_iter == 0) || rs.flag > 0 ||
                    ((iter + 1) % step_of(iter + 1) == 0) || (iter + 1 >= p->max_iter);
        orc_x_half(n, w.ATrp, w.ATci, w.ATv, w.y, w.x, w.x_hat, w.x_bar, w.z_bar, w.x_temp, w.l, w.u,
                   w.c, w.last_x, w.sigma, w.k, check);
        orc_y_half(m, w.Arp, w.Aci, w.Av, w.x_hat, w.y, w.y_bar, w.y_obj, w.y_temp, w.AL, w.AU,
                   w.last_y, w.sigma, w.lambda_max, w.k, check);
        w.k += 1;
        if (rs.flag > 0) rs.last_gap = compute_weighted_norm(&w);
        rs.inner += 1;
    }
    /* result fill, src/HPRLP.cu:239-257 */
    strncpy(res->status, status, sizeof(res->status) - 1);
    res->iter = iter;
    res->gap = r.gap;
    res->residuals = r.kkt;
    res->primal_obj = r.pobj;
    res->time = now_sec() - t_start;
    if (res->time4 == 0.0) res->time4 = res->time;
    if (res->time6 == 0.0) res->time6 = res->time;
    if (res->time8 == 0.0) res->time8 = res->time;
    if (res->iter4 == 0) res->iter4 = res->iter;
    if (res->iter6 == 0) res->iter6 = res->iter;
    if (res->iter8 == 0) res->iter8 = res->iter;
    res->lambda_max = w.lambda_max;
    res->n_trace = ntrace;
    res->n_restarts = rs.times;
    /* collect_solution, src/utils.cu:143-200 */
    for (int j = 0; j < n; ++j) {
        x_out[j] = (w.x_bar[j] / w.col_norm[j]) * w.sc.b_scale;
        z_out[j] = (w.z_bar[j] * w.col_norm[j]) * w.sc.c_scale;
    }
    for (int i = 0; i < m; ++i) y_out[i] = (w.y_bar[i] / w.row_norm[i]) * w.sc.c_scale;

    free(ATrp); free(ATci); free(w.ATv); free(w.Av); free(w.AL); free(w.AU); free(w.l); free(w.u);
    free(w.c); free(w.row_norm); free(w.col_norm); free(w.x); free(w.last_x); free(w.x_hat);
    free(w.x_bar); free(w.z_bar); free(w.x_temp); free(w.ATy); free(w.y); free(w.last_y);
    free(w.y_bar); free(w.y_obj); free(w.y_temp); free(w.Ax);
    return 0;
}

/* ---------------------------------------------------------------- timed loop for bench.py ------ */

double orc_time_iterations(int m, int n, const int *Arp, const int *Aci, const double *Av,
                           const int *ATrp, const int *ATci, const double *ATv, const double *AL,
                           const double *AU, const double *l, const double *u, const double *c,
                           double sigma, double lambda_max, int iters, double *x, double *y) {
    double *last_x = dup_vec(x, n), *last_y = dup_vec(y, m), *x_hat = zeros(n);
    double t0 = now_sec();
    for (int k = 0; k < iters; ++k) {
        orc_x_half(n, ATrp, ATci, ATv, y, x, x_hat, NULL, NULL, NULL, l, u, c, last_x, sigma, k, 0);
        orc_y_half(m, Arp, Aci, Av, x_hat, y, NULL, NULL, NULL, AL, AU, last_y, sigma, lambda_max, k, 0);
    }
    double t = now_sec() - t0;
    free(last_x); free(last_y); free(x_hat);
    return t;
}

/* ---------------------------------------------------------------- batched solve ---------------- */

/* src/batched_solver.cu:332-354 (long double accumulation, as the reference does on the host) */
static double bound_norm_host(const double *AL, const double *AU, int m, int k) {
    long double sum = 0.0;
    const long off = (long)k * m;
    for (int i = 0; i < m; ++i) {
        double lo = AL[off + i], hi = AU[off + i];
        double a = (isinf(lo) && lo < 0) ? 0.0 : fabs(lo);
        double b = (isinf(hi) && hi > 0) ? 0.0 : fabs(hi);
        double v = fmax(a, b);
        sum += (long double)v * v;
    }
    return sqrt((double)sum);
}
static double column_norm_host(const double *X, int n, int k) {
    long double sum = 0.0;
    const long off = (long)k * n;
    for (int i = 0; i < n; ++i) {
        double v = X[off + i];
        sum += (long double)v * v;
    }
    return sqrt((double)sum);
}

typedef struct {
    int m, n, B;
    const int *Arp, *Aci, *ATrp, *ATci;
    const double *Av, *ATv;
    double *C, *AL, *AU, *L, *U; /* scaled panels, column-major */
    double *X, *Xh, *Xb, *DX, *Y, *Yb, *DY, *Yobj, *Zb, *AX, *ATY, *lastX, *lastY;
    double *sigma;
    unsigned char *active;
    double lambda_max;
} orc_bws;

static void spmm(int rows, int cols_in, const int *rp, const int *ci, const double *v, const double *Xin,
                 double *Yout, int B) {
    for (int k = 0; k < B; ++k) orc_spmv(rows, rp, ci, v, Xin + (long)k * cols_in, Yout + (long)k * rows);
}

/* src/batched_solver.cu:626-666 */
static void b_weighted_norm(orc_bws *w, double *out) {
    spmm(w->m, w->n, w->Arp, w->Aci, w->Av, w->DX, w->AX, w->B);
    for (int k = 0; k < w->B; ++k) {
        const double *ax = w->AX + (long)k * w->m, *dy = w->DY + (long)k * w->m,
                     *dx = w->DX + (long)k * w->n;
        double dot_prod = 2.0 * dotv(ax, dy, w->m);
        double dyn = nrm2v(dy, w->m), dxn = nrm2v(dx, w->n);
        double dy_sq = dyn * dyn, dx_sq = dxn * dxn;
        double sg = w->sigma[k];
        double value = sg * (w->lambda_max * dy_sq) + dx_sq / sg + dot_prod;
        if (value < 0.0 && dy_sq > 0.0) {
            double cand = -(dot_prod + dx_sq / sg) / (sg * dy_sq) * 1.05;
            if (cand > w->lambda_max) w->lambda_max = cand;
            value = sg * (w->lambda_max * dy_sq) + dx_sq / sg + dot_prod;
        }
        out[k] = sqrt(fmax(value, 0.0));
    }
}

int orc_solve_batched(int m, int n, int nnz, const int *Arp, const int *Aci, const double *Av_in, int B,
                      const double *C, const double *AL, const double *AU, const double *L,
                      const double *U, const double *obj_constants, double model_obj_constant,
                      const orc_params *p, double lambda_override, double *Xo, double *Yo, double *Zo,
                      double *primal_obj, double *residuals, double *gap, int *iter_out, char *status,
                      double *lambda_out) {
    const long nB = (long)n * B, mB = (long)m * B;
    /* shared-A scaling with zero vectors, bc off: src/batched_solver.cu:959-981 */
    int *ATrp = (int *)malloc(((size_t)n + 1) * sizeof(int));
    int *ATci = (int *)malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(int));
    double *ATv = (double *)malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(double));
    double *Av = dup_vec(Av_in, nnz);
    orc_csr_transpose(m, n, nnz, Arp, Aci, Av_in, ATrp, ATci, ATv);
    double *zm1 = zeros(m), *zm2 = zeros(m), *zn1 = zeros(n), *zn2 = zeros(n), *zn3 = zeros(n);
    double *row_norm = zeros(m), *col_norm = zeros(n);
    orc_params mp = *p;
    mp.use_bc_scaling = 0;
    orc_scaling_scalars ssc;
    orc_scaling(m, n, Arp, Aci, Av, ATrp, ATci, ATv, zm1, zm2, zn1, zn2, zn3, &mp, row_norm, col_norm, &ssc);

    /* per-column vector scaling on the host: :792-885 */
    orc_bws w;
    memset(&w, 0, sizeof(w));
    w.m = m; w.n = n; w.B = B;
    w.Arp = Arp; w.Aci = Aci; w.Av = Av; w.ATrp = ATrp; w.ATci = ATci; w.ATv = ATv;
    w.C = dup_vec(C, nB); w.AL = dup_vec(AL, mB); w.AU = dup_vec(AU, mB);
    w.L = dup_vec(L, nB); w.U = dup_vec(U, nB);
    double *b_scale = zeros(B), *c_scale = zeros(B), *norm_b = zeros(B), *norm_c = zeros(B),
           *norm_b_org = zeros(B), *norm_c_org = zeros(B), *objc = zeros(B);
    for (int k = 0; k < B; ++k) {
        b_scale[k] = 1.0; c_scale[k] = 1.0;
        norm_b_org[k] = 1.0 + bound_norm_host(w.AL, w.AU, m, k);
        norm_c_org[k] = 1.0 + column_norm_host(w.C, n, k);
        for (int i = 0; i < m; ++i) {
            w.AL[(long)k * m + i] /= row_norm[i];
            w.AU[(long)k * m + i] /= row_norm[i];
        }
        for (int i = 0; i < n; ++i) {
            w.C[(long)k * n + i] /= col_norm[i];
            w.L[(long)k * n + i] *= col_norm[i];
            w.U[(long)k * n + i] *= col_norm[i];
        }
    }
    if (p->use_bc_scaling) {
        for (int k = 0; k < B; ++k) {
            b_scale[k] = 1.0 + bound_norm_host(w.AL, w.AU, m, k);
            c_scale[k] = 1.0 + column_norm_host(w.C, n, k);
            for (int i = 0; i < m; ++i) {
                w.AL[(long)k * m + i] /= b_scale[k];
                w.AU[(long)k * m + i] /= b_scale[k];
            }
            for (int i = 0; i < n; ++i) {
                w.C[(long)k * n + i] /= c_scale[k];
                w.L[(long)k * n + i] /= b_scale[k];
                w.U[(long)k * n + i] /= b_scale[k];
            }
        }
    }
    for (int k = 0; k < B; ++k) {
        norm_b[k] = bound_norm_host(w.AL, w.AU, m, k);
        norm_c[k] = column_norm_host(w.C, n, k);
        for (int i = 0; i < m; ++i) {
            double *lo = &w.AL[(long)k * m + i], *hi = &w.AU[(long)k * m + i];
            if (isinf(*lo) && *lo < 0) *lo = -1.0e100;
            if (isinf(*hi) && *hi > 0) *hi = 1.0e100;
        }
        for (int i = 0; i < n; ++i) {
            double *lo = &w.L[(long)k * n + i], *hi = &w.U[(long)k * n + i];
            if (isinf(*lo) && *lo < 0) *lo = -1.0e100;
            if (isinf(*hi) && *hi > 0) *hi = 1.0e100;
        }
        objc[k] = obj_constants ? obj_constants[k] : model_obj_constant;
    }
    /* power iteration on the scaled shared matrix: :994-1001 */
    if (lambda_override > 0) {
        w.lambda_max = lambda_override;
    } else {
        double *z0 = (double *)malloc((size_t)m * sizeof(double));
        orc_power_start_vector(m, 1ULL, 0, z0);
        w.lambda_max = orc_power_iteration(m, n, Arp, Aci, Av, ATrp, ATci, ATv, z0, 5000, 1e-4, NULL) * 1.01;
        free(z0);
    }
    /* workspace: :479-532 */
    w.X = zeros(nB); w.Xh = zeros(nB); w.Xb = zeros(nB); w.DX = zeros(nB); w.Zb = zeros(nB);
    w.ATY = zeros(nB); w.lastX = zeros(nB);
    w.Y = zeros(mB); w.Yb = zeros(mB); w.DY = zeros(mB); w.Yobj = zeros(mB); w.AX = zeros(mB);
    w.lastY = zeros(mB);
    w.sigma = zeros(B);
    w.active = (unsigned char *)malloc((size_t)B);
    for (int k = 0; k < B; ++k) {
        w.sigma[k] = (norm_b[k] > 1.0e-8 && norm_c[k] > 1.0e-8) ? norm_b[k] / norm_c[k] : 1.0;
        w.active[k] = 1;
    }
    /* restart state: :534-556 */
    int *rflag = (int *)calloc((size_t)B, sizeof(int)), *inner = (int *)calloc((size_t)B, sizeof(int));
    unsigned char *first = (unsigned char *)malloc((size_t)B);
    double *last_gap = zeros(B), *cur_gap = zeros(B), *save_gap = zeros(B), *best_gap = zeros(B),
           *best_sigma = zeros(B), *rsigma = zeros(B), *tmpB = zeros(B);
    double *r_pobj = zeros(B), *r_dobj = zeros(B), *r_rp = zeros(B), *r_rd = zeros(B), *r_gap = zeros(B),
           *r_kkt = zeros(B);
    int *final_iter = (int *)malloc((size_t)B * sizeof(int));
    const char **st = (const char **)malloc((size_t)B * sizeof(char *));
    for (int k = 0; k < B; ++k) {
        first[k] = 1;
        last_gap[k] = cur_gap[k] = save_gap[k] = best_gap[k] = ORC_INF;
        rsigma[k] = best_sigma[k] = w.sigma[k];
        r_kkt[k] = ORC_INF;
        final_iter[k] = p->max_iter;
        st[k] = "CONTINUE";
    }
    const int check_iter = p->check_iter > 1 ? p->check_iter : 1;
    const double t0 = now_sec();

    for (int iter = 0; iter <= p->max_iter; ++iter) { /* :1017-1084 */
        int periodic = (iter % check_iter) == 0;
        double elapsed = now_sec() - t0;
        if (periodic) {
            if (iter > 0) b_weighted_norm(&w, cur_gap);
            /* compute_residuals :578-624 */
            spmm(n, m, ATrp, ATci, ATv, w.Yb, w.ATY, B);
            spmm(m, n, Arp, Aci, Av, w.Xb, w.AX, B);
            for (int k = 0; k < B; ++k) {
                const long on = (long)k * n, om = (long)k * m;
                double rd2 = 0, rp2 = 0, lu2 = 0;
                for (int j = 0; j < n; ++j) {
                    double rd = (w.C[on + j] - w.ATY[on + j] - w.Zb[on + j]) * col_norm[j];
                    rd2 += rd * rd;
                }
                for (int i = 0; i < m; ++i) {
                    double v = w.AX[om + i];
                    double rp = row_norm[i] * fmax(fmin(w.AU[om + i] - v, 0.0), w.AL[om + i] - v);
                    rp2 += rp * rp;
                }
                double dot_cx = dotv(w.C + on, w.Xb + on, n);
                double dot_yy = dotv(w.Yobj + om, w.Yb + om, m);
                double dot_xz = dotv(w.Xb + on, w.Zb + on, n);
                double obj_scale = b_scale[k] * c_scale[k];
                r_pobj[k] = obj_scale * dot_cx + objc[k];
                r_dobj[k] = obj_scale * (dot_yy + dot_xz) + objc[k];
                r_rd[k] = c_scale[k] * sqrt(rd2) / norm_c_org[k];
                r_rp[k] = b_scale[k] * sqrt(rp2) / norm_b_org[k];
                if (iter == 0) {
                    for (int j = 0; j < n; ++j) {
                        double x = w.Xb[on + j];
                        double viol = x < w.L[on + j] ? w.L[on + j] - x : (x > w.U[on + j] ? x - w.U[on + j] : 0.0);
                        w.DX[on + j] = viol / col_norm[j];
                        lu2 += w.DX[on + j] * w.DX[on + j];
                    }
                    r_rp[k] = fmax(r_rp[k], b_scale[k] * sqrt(lu2));
                }
                r_gap[k] = fabs(r_pobj[k] - r_dobj[k]) / (1.0 + fabs(r_pobj[k]) + fabs(r_dobj[k]));
                r_kkt[k] = fmax(r_rp[k], fmax(r_rd[k], r_gap[k]));
            }
            for (int k = 0; k < B; ++k)
                if (w.active[k] && r_kkt[k] <= p->stop_tol) {
                    st[k] = "OPTIMAL";
                    final_iter[k] = iter;
                    w.active[k] = 0;
                }
        }
        int all_done = 1;
        for (int k = 0; k < B; ++k) all_done = all_done && strcmp(st[k], "CONTINUE") != 0;
        if (all_done) break;
        if (iter >= p->max_iter || elapsed >= p->time_limit) {
            const char *fs = elapsed >= p->time_limit ? "TIME_LIMIT" : "ITER_LIMIT";
            for (int k = 0; k < B; ++k)
                if (strcmp(st[k], "CONTINUE") == 0) {
                    st[k] = fs;
                    final_iter[k] = iter;
                    w.active[k] = 0;
                }
            break;
        }
        for (int k = 0; k < B; ++k) rflag[k] = 0;
        if (periodic) { /* check_restart :667-700 */
            for (int k = 0; k < B; ++k) rsigma[k] = w.sigma[k];
            for (int k = 0; k < B; ++k) {
                if (!w.active[k]) continue;
                if (first[k]) {
                    if (iter == check_iter) {
                        first[k] = 0; rflag[k] = 1;
                        best_gap[k] = cur_gap[k]; best_sigma[k] = rsigma[k];
                    }
                } else {
                    if (cur_gap[k] < 0.0) cur_gap[k] = 1.0e-6;
                    if (cur_gap[k] <= 0.2 * last_gap[k]) rflag[k] = 1;
                    if (cur_gap[k] <= 0.6 * last_gap[k] && cur_gap[k] > save_gap[k]) rflag[k] = 2;
                    if (inner[k] >= 0.2 * iter) rflag[k] = 3;
                    if (best_gap[k] > cur_gap[k]) { best_gap[k] = cur_gap[k]; best_sigma[k] = rsigma[k]; }
                    save_gap[k] = cur_gap[k];
                }
            }
        }
        /* update_sigma :702-745 */
        int any = 0;
        for (int k = 0; k < B; ++k) any = any || (rflag[k] >= 1 && rflag[k] <= 3);
        if (any) {
            for (long t = 0; t < nB; ++t) w.DX[t] = w.Xb[t] - w.lastX[t];
            for (long t = 0; t < mB; ++t) w.DY[t] = w.Yb[t] - w.lastY[t];
            for (int k = 0; k < B; ++k) {
                if (!w.active[k]) continue;
                if (rflag[k] >= 1 && rflag[k] <= 3) {
                    double pm = nrm2v(w.DX + (long)k * n, n), dm = nrm2v(w.DY + (long)k * m, m);
                    rsigma[k] = sigma_formula(pm, dm, w.lambda_max, cur_gap[k], best_gap[k], best_sigma[k],
                                              r_rd[k], r_rp[k], r_gap[k]);
                }
            }
            for (int k = 0; k < B; ++k) w.sigma[k] = rsigma[k];
        }
        /* do_restart :747-769 */
        int restarted = 0;
        for (int k = 0; k < B; ++k) restarted = restarted || rflag[k] > 0;
        if (restarted) {
            for (int k = 0; k < B; ++k) {
                if (rflag[k] <= 0) continue;
                memcpy(w.X + (long)k * n, w.Xb + (long)k * n, (size_t)n * sizeof(double));
                memcpy(w.lastX + (long)k * n, w.Xb + (long)k * n, (size_t)n * sizeof(double));
                memcpy(w.Y + (long)k * m, w.Yb + (long)k * m, (size_t)m * sizeof(double));
                memcpy(w.lastY + (long)k * m, w.Yb + (long)k * m, (size_t)m * sizeof(double));
                if (w.active[k]) { inner[k] = 0; save_gap[k] = ORC_INF; }
            }
        }
        int to_check = ((iter + 1) % check_iter) == 0 || restarted || ((iter + 1) % step_of(iter + 1) == 0);
        /* update_x_z / update_y :771-790 with kernels :122-236 */
        spmm(n, m, ATrp, ATci, ATv, w.Y, w.ATY, B);
        for (int k = 0; k < B; ++k) {
            if (!w.active[k]) continue;
            const long on = (long)k * n;
            const double sg = w.sigma[k], f1 = 1.0 / (inner[k] + 2.0), f2 = 1.0 - f1;
            for (int j = 0; j < n; ++j) {
                long t = on + j;
                double xi = w.X[t];
                double zt = xi + sg * (w.ATY[t] - w.C[t]);
                double xb = fmin(fmax(zt, w.L[t]), w.U[t]);
                double xh = 2.0 * xb - xi;
                if (to_check) {
                    w.DX[t] = xb - xh;
                    w.Zb[t] = (xb - zt) / sg;
                    w.Xb[t] = xb;
                }
                w.Xh[t] = xh;
                w.X[t] = f2 * xh + f1 * w.lastX[t];
            }
        }
        spmm(m, n, Arp, Aci, Av, w.Xh, w.AX, B);
        for (int k = 0; k < B; ++k) {
            if (!w.active[k]) continue;
            const long om = (long)k * m;
            const double fact1 = w.lambda_max * w.sigma[k], f1 = 1.0 / (inner[k] + 2.0), f2 = 1.0 - f1;
            for (int i = 0; i < m; ++i) {
                long t = om + i;
                double yi = w.Y[t];
                double v = w.AX[t] - fact1 * yi;
                double d = fmax(w.AL[t] - v, fmin(w.AU[t] - v, 0.0));
                double yb = d / fact1;
                double yh = 2.0 * yb - yi;
                if (to_check) {
                    w.DY[t] = yb - yh;
                    w.Yb[t] = yb;
                    w.Yobj[t] = v + d;
                }
                w.Y[t] = f2 * yh + f1 * w.lastY[t];
            }
        }
        for (int k = 0; k < B; ++k)
            if (w.active[k]) inner[k] += 1;
        if (restarted) {
            b_weighted_norm(&w, tmpB);
            for (int k = 0; k < B; ++k)
                if (rflag[k] > 0) last_gap[k] = tmpB[k];
        }
    }
    /* collect_results :887-935 */
    for (int k = 0; k < B; ++k) {
        for (int i = 0; i < n; ++i) {
            long idx = (long)k * n + i;
            Xo[idx] = (w.Xb[idx] / col_norm[i]) * b_scale[k];
            Zo[idx] = (w.Zb[idx] * col_norm[i]) * c_scale[k];
        }
        for (int i = 0; i < m; ++i) {
            long idx = (long)k * m + i;
            Yo[idx] = (w.Yb[idx] / row_norm[i]) * c_scale[k];
        }
        primal_obj[k] = r_pobj[k];
        residuals[k] = r_kkt[k];
        gap[k] = r_gap[k];
        iter_out[k] = final_iter[k];
        memset(status + 64 * (long)k, 0, 64);
        strncpy(status + 64 * (long)k, st[k], 63);
    }
    if (lambda_out) *lambda_out = w.lambda_max;

    free(ATrp); free(ATci); free(ATv); free(Av); free(zm1); free(zm2); free(zn1); free(zn2); free(zn3);
    free(row_norm); free(col_norm); free(w.C); free(w.AL); free(w.AU); free(w.L); free(w.U);
    free(b_scale); free(c_scale); free(norm_b); free(norm_c); free(norm_b_org); free(norm_c_org); free(objc);
    free(w.X); free(w.Xh); free(w.Xb); free(w.DX); free(w.Zb); free(w.ATY); free(w.lastX);
    free(w.Y); free(w.Yb); free(w.DY); free(w.Yobj); free(w.AX); free(w.lastY); free(w.sigma); free(w.active);
    free(rflag); free(inner); free(first); free(last_gap); free(cur_gap); free(save_gap); free(best_gap);
    free(best_sigma); free(rsigma); free(tmpB); free(r_pobj); free(r_dobj); free(r_rp); free(r_rd);
    free(r_gap); free(r_kkt); free(final_iter); free((void *)st);
    return 0;
}
