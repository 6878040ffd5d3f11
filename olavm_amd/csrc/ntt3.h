// Interface of the third-generation NTT pass kernels (ntt3.hip) towards the planner (ntt2.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "ntt3_core.cuh"

namespace ola {

// Launch one pass over `cols` columns x `cosets` cosets of a 2^p.log_n transform (log_n >= 18): mode N3_STRIDED with
// R in 5..9 transform bits [p.lo, p.lo + R), N3_LAST_BITREV (R = 13, p.lo = 0) or N3_LAST_NATURAL (R = 9, p.lo = 0).
void ntt3_launch(const N3Params& p, int R, int mode, bool inverse, size_t cols, size_t cosets, hipStream_t stream);

}  // namespace ola
