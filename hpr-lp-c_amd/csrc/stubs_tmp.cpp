// temporary: replaced by mps_reader.cpp / batched.cpp / dist.cpp
#include <iostream>
#include "HPRLP.h"
#include "dist.h"
extern "C" LP_info_cpu *create_model_from_mps(const char *) { std::cerr << "[error] MPS reader not built yet\n"; return nullptr; }
