/*
 * structs.h -- public ABI structs of the HPR-LP boundary, MI355X build.
 *
 * Layout contract (x86-64 SysV), field for field with the reference's public structs
 * (reference include/structs.h:16-22 sparseMatrix, :25-40 HPRLP_parameters, :44-65 HPRLP_results,
 * :68-90 HPRLP_batched_results, :231-240 LP_info_cpu, :286-306 HPRLP_LP_Data).  Julia binds these by
 * raw layout (reference bindings/julia/package/src/wrapper.jl:92-164), so sizes and offsets are
 * pinned by the static_asserts at the end of this file and by tests/test_abi.py.
 *
 * Unlike the reference header this one pulls in no vendor (cuBLAS/cuSPARSE) types: the internal
 * workspace structs are private to hpr-lp-c_amd/csrc and never cross the boundary.
 */
#ifndef HPRLP_STRUCTS_H
#define HPRLP_STRUCTS_H

#include <math.h>   /* INFINITY / HUGE_VAL for callers' bounds: the reference headers pull <cmath> in transitively */
#include <stddef.h>
#include <stdint.h>

#define HPRLP_FLOAT double

#ifdef __cplusplus
#define HPRLP_DFLT(v) = v
#else
#include <stdbool.h>
#define HPRLP_DFLT(v)
#endif

/* CSR matrix, int32 indices / FP64 values (reference include/structs.h:16-22). */
struct sparseMatrix {
    int row, col;
    int numElements;
    int *colIndex;
    int *rowPtr;
    HPRLP_FLOAT *value;
};

/* Solver parameters; defaults as reference include/structs.h:26-39.  In C (no default member
 * initialisers) use HPRLP_PARAMETERS_DEFAULT or pass NULL to solve(). */
struct HPRLP_parameters {
    int max_iter HPRLP_DFLT(INT32_MAX);
    HPRLP_FLOAT stop_tol HPRLP_DFLT(1e-4);
    HPRLP_FLOAT time_limit HPRLP_DFLT(3600.0);
    int device_number HPRLP_DFLT(0);
    int check_iter HPRLP_DFLT(150);
    bool CUSPARSE_spmv HPRLP_DFLT(false);    /* accepted and ignored: one native HIP back-end */
    bool autotune_verbose HPRLP_DFLT(false); /* accepted and ignored: nothing to autotune     */
    bool use_CR_scaling HPRLP_DFLT(true);
    bool use_Ruiz_scaling HPRLP_DFLT(true);
    bool use_Pock_Chambolle_scaling HPRLP_DFLT(true);
    bool use_bc_scaling HPRLP_DFLT(true);
    bool use_presolve HPRLP_DFLT(true);
};
#define HPRLP_PARAMETERS_DEFAULT \
    { INT32_MAX, 1e-4, 3600.0, 0, 150, false, false, true, true, true, true, true }

/* Result of one solve (reference include/structs.h:44-65).  x, y, z come from libc malloc and are
 * released by the caller with free(). */
struct HPRLP_results {
    HPRLP_FLOAT residuals;
    HPRLP_FLOAT primal_obj;
    HPRLP_FLOAT gap;
    HPRLP_FLOAT time4 HPRLP_DFLT(0.0);
    HPRLP_FLOAT time6 HPRLP_DFLT(0.0);
    HPRLP_FLOAT time8 HPRLP_DFLT(0.0);
    HPRLP_FLOAT time HPRLP_DFLT(0.0);
    int iter4 HPRLP_DFLT(0);
    int iter6 HPRLP_DFLT(0);
    int iter8 HPRLP_DFLT(0);
    int iter HPRLP_DFLT(0);
    char status[64]; /* "OPTIMAL", "TIME_LIMIT", "ITER_LIMIT", "ERROR" */
    HPRLP_FLOAT *x HPRLP_DFLT(NULL);
    HPRLP_FLOAT *y HPRLP_DFLT(NULL);
    HPRLP_FLOAT *z HPRLP_DFLT(NULL);
};

/* Result of solve_batched (reference include/structs.h:68-90).  Column-major host arrays: x/z are
 * n x batch_size, y is m x batch_size; status is batch_size slots of 64 bytes.  Released by
 * free_batched_results(). */
struct HPRLP_batched_results {
    int m HPRLP_DFLT(0);
    int n HPRLP_DFLT(0);
    int batch_size HPRLP_DFLT(0);
    HPRLP_FLOAT *x HPRLP_DFLT(NULL);
    HPRLP_FLOAT *y HPRLP_DFLT(NULL);
    HPRLP_FLOAT *z HPRLP_DFLT(NULL);
    HPRLP_FLOAT *primal_obj HPRLP_DFLT(NULL);
    HPRLP_FLOAT *residuals HPRLP_DFLT(NULL);
    HPRLP_FLOAT *gap HPRLP_DFLT(NULL);
    int *iter HPRLP_DFLT(NULL);
    char *status HPRLP_DFLT(NULL);
    HPRLP_FLOAT time HPRLP_DFLT(0.0);
    HPRLP_FLOAT setup_time HPRLP_DFLT(0.0);
    HPRLP_FLOAT solve_time HPRLP_DFLT(0.0);
    HPRLP_FLOAT power_time HPRLP_DFLT(0.0);
};

/* Host model (reference include/structs.h:231-240): min c'x + obj_constant, AL<=Ax<=AU, l<=x<=u. */
struct LP_info_cpu {
    int m, n;
    struct sparseMatrix *A;
    HPRLP_FLOAT *AL;
    HPRLP_FLOAT *AU;
    HPRLP_FLOAT *c;
    HPRLP_FLOAT *l;
    HPRLP_FLOAT *u;
    HPRLP_FLOAT obj_constant;
};

/* Caller-owned array view of an LP (reference include/structs.h:286-306). */
struct HPRLP_LP_Data {
    int m;
    int n;
    int nnz;
    int *rowPtr;
    int *colIndex;
    HPRLP_FLOAT *values;
    bool is_csc;
    HPRLP_FLOAT *AL;
    HPRLP_FLOAT *AU;
    HPRLP_FLOAT *l;
    HPRLP_FLOAT *u;
    HPRLP_FLOAT *c;
};

#ifndef __cplusplus
typedef struct sparseMatrix sparseMatrix;
typedef struct HPRLP_parameters HPRLP_parameters;
typedef struct HPRLP_results HPRLP_results;
typedef struct HPRLP_batched_results HPRLP_batched_results;
typedef struct LP_info_cpu LP_info_cpu;
typedef struct HPRLP_LP_Data HPRLP_LP_Data;
#else
static_assert(sizeof(sparseMatrix) == 40 && offsetof(sparseMatrix, colIndex) == 16 &&
                  offsetof(sparseMatrix, value) == 32, "sparseMatrix ABI");
static_assert(sizeof(HPRLP_parameters) == 40 && offsetof(HPRLP_parameters, stop_tol) == 8 &&
                  offsetof(HPRLP_parameters, device_number) == 24 &&
                  offsetof(HPRLP_parameters, check_iter) == 28 &&
                  offsetof(HPRLP_parameters, CUSPARSE_spmv) == 32 &&
                  offsetof(HPRLP_parameters, use_presolve) == 38, "HPRLP_parameters ABI");
static_assert(sizeof(HPRLP_results) == 160 && offsetof(HPRLP_results, iter4) == 56 &&
                  offsetof(HPRLP_results, status) == 72 && offsetof(HPRLP_results, x) == 136,
              "HPRLP_results ABI");
static_assert(sizeof(HPRLP_batched_results) == 112 && offsetof(HPRLP_batched_results, x) == 16 &&
                  offsetof(HPRLP_batched_results, status) == 72 &&
                  offsetof(HPRLP_batched_results, time) == 80, "HPRLP_batched_results ABI");
static_assert(sizeof(LP_info_cpu) == 64 && offsetof(LP_info_cpu, A) == 8 &&
                  offsetof(LP_info_cpu, obj_constant) == 56, "LP_info_cpu ABI");
#endif

#endif /* HPRLP_STRUCTS_H */
