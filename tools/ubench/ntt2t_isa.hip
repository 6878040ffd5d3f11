// Instantiates the T-form pass kernels of the 2^22 transforms for tools/isa_count.py (device code only, seconds to compile):
//   python tools/isa_count.py tools/ubench/ntt2t_isa.hip --per 16 --top 8
#include "ntt2t.cuh"
namespace ola {
template __global__ void ntt2t_pass_kernel<7, N2_STRIDED, false, 8, 0>(Ntt2Params);        // first pass of a plain transform
template __global__ void ntt2t_pass_kernel<7, N2_STRIDED, false, 8, 1>(Ntt2Params);        // strided pass with load multipliers (LDS table)
template __global__ void ntt2t_pass_kernel<8, N2_STRIDED, false, 8, 1>(Ntt2Params);
template __global__ void ntt2t_pass_kernel<8, N2_NATURAL_LAST, false, 8, 2>(Ntt2Params);   // closing passes (register multipliers)
template __global__ void ntt2t_pass_kernel<8, N2_BITREV_LAST, false, 8, 2>(Ntt2Params);
}
