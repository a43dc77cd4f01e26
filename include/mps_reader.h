/* mps_reader.h -- compatibility header.  The reference's Python and MATLAB bindings and its
 * solve_mps_file driver #include "mps_reader.h" (reference bindings/python/src/hprlp_pybind.cpp:20,
 * bindings/matlab/src/hprlp_mex.cpp:12, src/solve_mps_file.cpp:5) but use nothing from it beyond the
 * boundary types; the MPS reader itself is reached through create_model_from_mps() in HPRLP.h. */
#ifndef HPRLP_MPS_READER_H
#define HPRLP_MPS_READER_H
#include "structs.h"
#endif
