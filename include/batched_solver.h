/*
 * batched_solver.h -- B linear programs sharing one sparse matrix A, solved together on one GPU.
 * Same two entry points as reference include/batched_solver.h:23-33.
 *
 * Member k solves  min C[:,k]'x + obj_constants[k]  s.t.  AL[:,k] <= A x <= AU[:,k],  l[:,k] <= x <= u[:,k].
 * All dense inputs are column-major: C, l, u are n x batch_size; AL, AU are m x batch_size.
 * obj_constants may be NULL (the model's constant is used for every member).
 */
#ifndef HPRLP_BATCHED_SOLVER_H
#define HPRLP_BATCHED_SOLVER_H

#include "structs.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Replaces reference src/batched_solver.cu:939-1092. */
HPRLP_batched_results solve_batched(const LP_info_cpu *model, int batch_size, const HPRLP_FLOAT *C,
                                    const HPRLP_FLOAT *AL, const HPRLP_FLOAT *AU, const HPRLP_FLOAT *l,
                                    const HPRLP_FLOAT *u, const HPRLP_FLOAT *obj_constants,
                                    const HPRLP_parameters *param);

/* Frees the eight result arrays and zeroes the struct.  Replaces reference src/batched_solver.cu:1094-1105. */
void free_batched_results(HPRLP_batched_results *results);

#ifdef __cplusplus
}
#endif
#endif /* HPRLP_BATCHED_SOLVER_H */
