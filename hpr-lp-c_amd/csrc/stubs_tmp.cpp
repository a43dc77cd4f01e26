// temporary: replaced by mps_reader.cpp / batched.cpp / dist.cpp
#include <iostream>
#include "HPRLP.h"
#include "dist.h"
extern "C" LP_info_cpu *create_model_from_mps(const char *) { std::cerr << "[error] MPS reader not built yet\n"; return nullptr; }
extern "C" HPRLP_batched_results solve_batched(const LP_info_cpu *, int, const double *, const double *, const double *, const double *, const double *, const double *, const HPRLP_parameters *) { return HPRLP_batched_results(); }
extern "C" void free_batched_results(HPRLP_batched_results *) {}
