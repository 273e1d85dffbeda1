// The fast-arithmetic build of the path kernel (PT_ARITH=fast): pt_kernels.hip compiled a second time with the shading
// half on the hardware's approximate divide / sqrt / rsq / exp2 / log2 / sin / cos and with FMA contraction
// (this translation unit is built with -ffp-contract=fast; build.py). Ray generation and the intersection tests keep
// contraction off (pt_vec.h, *_x helpers), so primary-hit ids are those of the exact build. Results are validated by
// relMSE and id agreement against the oracle, not by equality; the exact build stays the default.
#define PT_FAST_ARITH 1
#include "pt_kernels.hip"
