/*
 * HPRLP.h -- the drop-in boundary of the HPR-LP solver, MI355X (gfx950) build.
 *
 * Same seven unmangled entry points, signatures, ownership and status strings as the reference
 * library (reference include/HPRLP.h:41,105-111,140,180,202 and include/batched_solver.h:23-33),
 * implemented by hpr-lp-c_amd/csrc as hand-written HIP for CDNA4.  Built as lib/libhprlp.so so the
 * reference's Python/Julia/MATLAB bindings and its solve_mps_file driver link against it unchanged
 * (INTEGRATION.md).  This header compiles with a plain C or C++ host compiler: no HIP, CUDA or
 * vendor-library headers are needed by callers.
 *
 * Errors: no error codes.  create_model_* return NULL and write a message to stderr; solve* return
 * status "ERROR" with NULL vectors.  No exception crosses this boundary (the reference lets
 * std::runtime_error escape, reference include/cuda_kernels/cuda_check.h:56-63).
 */
#ifndef HPRLP_H
#define HPRLP_H

#include "structs.h"
#include "batched_solver.h"

#ifdef __cplusplus
#define HPRLP_DEFAULT_ARG(v) = v
extern "C" {
#else
#define HPRLP_DEFAULT_ARG(v)
#endif

/* Scale, estimate lambda_max, run the HPR loop on one device, unscale (no presolve).
 * Replaces reference src/HPRLP.cu:116-311.  param must not be NULL. */
HPRLP_results HPRLP_main_solve(const LP_info_cpu *lp_info_cpu, const HPRLP_parameters *param);

/* Deep-copy caller arrays into a new model.  is_csc: rowPtr/colIndex describe columns.
 * NULL on m,n,nnz<=0, NULL arrays, or inconsistent row pointers.
 * Replaces reference src/HPRLP.cu:321-446. */
LP_info_cpu *create_model_from_arrays(int m, int n, int nnz, const int *rowPtr, const int *colIndex,
                                      const HPRLP_FLOAT *values, const HPRLP_FLOAT *AL,
                                      const HPRLP_FLOAT *AU, const HPRLP_FLOAT *l, const HPRLP_FLOAT *u,
                                      const HPRLP_FLOAT *c, bool is_csc HPRLP_DEFAULT_ARG(false));

/* Parse a (free-format) .mps or .mps.gz file into a new model; NULL on error.
 * Replaces reference src/HPRLP.cu:451-488. */
LP_info_cpu *create_model_from_mps(const char *mps_file_path);

/* Solve a model; param==NULL means defaults.  The model is not modified and can be re-used.
 * Caller free()s result.x/y/z.  Replaces reference src/HPRLP.cu:493-524. */
HPRLP_results solve(const LP_info_cpu *model, const HPRLP_parameters *param);

/* Release a model created by create_model_*; NULL is a no-op.  Replaces reference src/HPRLP.cu:529-537. */
void free_model(LP_info_cpu *model);

#ifdef __cplusplus
}
#endif
#endif /* HPRLP_H */
