/* preprocess.h -- compatibility header: the reference's MATLAB MEX file includes it (reference
 * bindings/matlab/src/hprlp_mex.cpp:13) and only touches LP_info_cpu::{m,n,obj_constant}. */
#ifndef HPRLP_PREPROCESS_H
#define HPRLP_PREPROCESS_H
#include "structs.h"
#endif
