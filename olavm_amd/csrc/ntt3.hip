// Third-generation NTT pass kernels (gfx950): the per-thread phases of ntt3_core.cuh strung together with barriers.
// Own translation unit (the kernels are long straight-line code); the planner that calls ntt3_launch lives with the twiddle
// tables in ntt2.hip.  Reference semantics: plonky2/field/src/cfft/mod.rs:22-231.
#include <hip/hip_runtime.h>

#include "device_ctx.h"
#include "ntt3.h"
#include "ntt3_core.cuh"

namespace ola {

// 256 threads, 32 elements each; two workgroups per CU (64 KB of LDS each, <= 256 VGPRs).  Grid: x = column, y = tile, so
// that workgroups launched together read the same 64 KB slice of the pass-multiplier table (p.col_major = 0 swaps the roles).
template <int R, int MODE, bool INV>
__global__ __launch_bounds__(N3_THREADS, 2) void ntt3_pass_kernel(N3Params p, int col_major) {
    typedef N3Cfg<R, MODE> C;
    extern __shared__ __attribute__((aligned(16))) unsigned char n3_smem[];
    u64* lds = reinterpret_cast<u64*>(n3_smem);
    const int tid = threadIdx.x;
    const u32 coset = blockIdx.z;
    const u32 tile = col_major ? blockIdx.y : blockIdx.x;
    const size_t col = col_major ? blockIdx.x : blockIdx.y;
    N3Addr<R, MODE> a;
    a.init(p.log_n, p.lo, tile);
    const u64* __restrict__ in = p.in + col * p.in_col_stride + coset * p.in_coset_stride;
    u64* __restrict__ out = p.out + col * p.out_col_stride + coset * p.out_coset_stride;
    T4<i32> x[N3_REGS];
    n3_load<R, MODE, i32>(p, a, in, tid, coset, x);
    n3_round<R, MODE, INV, 0, i32>(p, a, tid, coset, x);
    if constexpr (C::NR > 1) {
        n3_xchg_write<R, MODE, 0, 0, i32>(tid, x, lds);
        __syncthreads();
        n3_xchg_read<R, MODE, 0, 0, i32>(tid, x, lds);
        __syncthreads();
        n3_xchg_write<R, MODE, 0, 1, i32>(tid, x, lds);
        __syncthreads();
        n3_xchg_read<R, MODE, 0, 1, i32>(tid, x, lds);
        n3_round<R, MODE, INV, 1, i32>(p, a, tid, coset, x);
    }
    if constexpr (C::NR > 2) {
        __syncthreads();
        n3_xchg_write<R, MODE, 1, 0, i32>(tid, x, lds);
        __syncthreads();
        n3_xchg_read<R, MODE, 1, 0, i32>(tid, x, lds);
        __syncthreads();
        n3_xchg_write<R, MODE, 1, 1, i32>(tid, x, lds);
        __syncthreads();
        n3_xchg_read<R, MODE, 1, 1, i32>(tid, x, lds);
        n3_round<R, MODE, INV, 2, i32>(p, a, tid, coset, x);
    }
    if (MODE == N3_LAST_BITREV) {
        __syncthreads();
        n3_final_write<R, MODE, i32>(tid, x, lds);
        __syncthreads();
        n3_final_store<R, MODE>(a, out, tid, lds);
    } else {
        n3_store_direct<R, MODE, i32>(a, out, tid, x);
    }
}

template <int R, int MODE, bool INV>
static void n3_launch_t(const N3Params& p, size_t cols, size_t cosets, hipStream_t stream) {
    auto kern = ntt3_pass_kernel<R, MODE, INV>;
    const size_t lds_bytes = (size_t)8 << N3_TILE_BITS;
    static bool attr_set = false;
    if (!attr_set) {
        HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        attr_set = true;
    }
    static const int col_major = [] { const char* e = getenv("OLA_NTT3_COL_MAJOR"); return e ? atoi(e) : 1; }();
    const size_t tiles = (size_t)1 << (p.log_n - N3_TILE_BITS);
    const bool cm = col_major && tiles <= 65535;
    dim3 grid(cm ? (unsigned)cols : (unsigned)tiles, cm ? (unsigned)tiles : (unsigned)cols, (unsigned)cosets);
    hipLaunchKernelGGL(kern, grid, dim3(N3_THREADS), lds_bytes, stream, p, cm ? 1 : 0);
}

template <bool INV>
static void n3_launch_i(const N3Params& p, int R, int mode, size_t cols, size_t cosets, hipStream_t s) {
    if (mode == N3_LAST_BITREV) { n3_launch_t<13, N3_LAST_BITREV, INV>(p, cols, cosets, s); return; }
    if (mode == N3_LAST_NATURAL) { n3_launch_t<9, N3_LAST_NATURAL, INV>(p, cols, cosets, s); return; }
    switch (R) {
        case 5: n3_launch_t<5, N3_STRIDED, INV>(p, cols, cosets, s); break;
        case 6: n3_launch_t<6, N3_STRIDED, INV>(p, cols, cosets, s); break;
        case 7: n3_launch_t<7, N3_STRIDED, INV>(p, cols, cosets, s); break;
        case 8: n3_launch_t<8, N3_STRIDED, INV>(p, cols, cosets, s); break;
        case 9: n3_launch_t<9, N3_STRIDED, INV>(p, cols, cosets, s); break;
        default: throw OlaError(-7, "ntt3: strided pass width out of range");
    }
}

void ntt3_launch(const N3Params& p0, int R, int mode, bool inverse, size_t cols, size_t cosets, hipStream_t stream) {
    N3Params p = p0;
    p.ncols = cols;
    if (mode == N3_STRIDED && p.lo < N3_TILE_BITS - R) throw OlaError(-7, "ntt3: strided pass below its own tile");
    if (inverse) n3_launch_i<true>(p, R, mode, cols, cosets, stream);
    else n3_launch_i<false>(p, R, mode, cols, cosets, stream);
}

}  // namespace ola
