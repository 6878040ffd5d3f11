// Third-generation NTT pass kernels (gfx950): the per-thread phases of ntt3_core.cuh strung together with barriers.
// Own translation unit (the kernels are long straight-line code); the planner that calls ntt3_launch lives with the twiddle
// tables in ntt2.hip.  Reference semantics: plonky2/field/src/cfft/mod.rs:22-231.
#include <hip/hip_runtime.h>

#include "device_ctx.h"
#include "ntt3.h"
#include "ntt3_core.cuh"

namespace ola {

// 256 threads, 32 elements each.  WPS = waves per SIMD the kernel is built for: 2 (two workgroups per CU, 64 KB of LDS each,
// elements exchanged as two 8-byte halves) or 3 (three workgroups, <= 168 VGPRs, 32 KB of LDS, exchanged limb by limb).
// Grid: x = column, y = tile, so that workgroups launched together read the same 64 KB slice of the pass-multiplier table
// (col_major = 0 swaps the roles).
template <int R, int MODE, bool INV, int RND, int WPS>
__device__ __forceinline__ void n3_exchange(int tid, T4<i32> (&x)[N3_REGS], unsigned char* smem, bool first) {
    if (WPS == 2) {
        u64* lds = reinterpret_cast<u64*>(smem);
        if (!first) __syncthreads();   // the buffer may still be read by the previous exchange
        n3_xchg_write<R, MODE, RND, 0, i32>(tid, x, lds);
        __syncthreads();
        n3_xchg_read<R, MODE, RND, 0, i32>(tid, x, lds);
        __syncthreads();
        n3_xchg_write<R, MODE, RND, 1, i32>(tid, x, lds);
        __syncthreads();
        n3_xchg_read<R, MODE, RND, 1, i32>(tid, x, lds);
    } else {
        u32* lds = reinterpret_cast<u32*>(smem);
        if (!first) __syncthreads();
        n3_xchg4_write<R, MODE, RND, 0, i32>(tid, x, lds);
        __syncthreads();
        n3_xchg4_read<R, MODE, RND, 0, i32>(tid, x, lds);
        __syncthreads();
        n3_xchg4_write<R, MODE, RND, 1, i32>(tid, x, lds);
        __syncthreads();
        n3_xchg4_read<R, MODE, RND, 1, i32>(tid, x, lds);
        __syncthreads();
        n3_xchg4_write<R, MODE, RND, 2, i32>(tid, x, lds);
        __syncthreads();
        n3_xchg4_read<R, MODE, RND, 2, i32>(tid, x, lds);
        __syncthreads();
        n3_xchg4_write<R, MODE, RND, 3, i32>(tid, x, lds);
        __syncthreads();
        n3_xchg4_read<R, MODE, RND, 3, i32>(tid, x, lds);
    }
}

template <int R, int MODE, bool INV, int WPS>
__global__ __launch_bounds__(N3_THREADS, WPS) void ntt3_pass_kernel(N3Params p, int col_major) {
    typedef N3Cfg<R, MODE> C;
    extern __shared__ __attribute__((aligned(16))) unsigned char n3_smem[];
    const int tid = threadIdx.x;
    const u32 coset = blockIdx.z;
    const u32 tile = col_major ? blockIdx.y : blockIdx.x;
    const size_t col = col_major ? blockIdx.x : blockIdx.y;
    N3Addr<R, MODE> a;
    a.init(p.log_n, p.lo, tile);
    const u64* __restrict__ in = p.in + col * p.in_col_stride + coset * p.in_coset_stride;
    u64* __restrict__ out = p.out + col * p.out_col_stride + coset * p.out_coset_stride;
    T4<i32> x[N3_REGS];
    constexpr int G = (WPS == 2) ? 8 : 4;
    n3_load<R, MODE, i32, G>(p, a, in, tid, coset, x);
    n3_round<R, MODE, INV, 0, i32, G>(p, a, tid, coset, x);
    if constexpr (C::NR > 1) {
        n3_exchange<R, MODE, INV, 0, WPS>(tid, x, n3_smem, true);
        n3_round<R, MODE, INV, 1, i32, G>(p, a, tid, coset, x);
    }
    if constexpr (C::NR > 2) {
        n3_exchange<R, MODE, INV, 1, WPS>(tid, x, n3_smem, false);
        n3_round<R, MODE, INV, 2, i32, G>(p, a, tid, coset, x);
    }
    if (MODE == N3_LAST_BITREV) {
        if (WPS == 2) {
            u64* lds = reinterpret_cast<u64*>(n3_smem);
            __syncthreads();
            n3_final_write<R, MODE, i32>(tid, x, lds);
            __syncthreads();
            n3_final_store<R, MODE>(a, out, tid, lds);
        } else {
            u32* lds = reinterpret_cast<u32*>(n3_smem);
            u64 c[N3_REGS];
#pragma unroll
            for (int j = 0; j < N3_REGS; j++) c[j] = tf_to_u64(x[j]);
            u32 wl[N3_REGS], wh[N3_REGS];
            __syncthreads();
            n3_final4_write<R, MODE, 0>(tid, c, lds);
            __syncthreads();
            n3_final4_read<0>(tid, wl, lds);
            __syncthreads();
            n3_final4_write<R, MODE, 1>(tid, c, lds);
            __syncthreads();
            n3_final4_read<1>(tid, wh, lds);
#pragma unroll
            for (int jj = 0; jj < N3_REGS; jj++) *n3_at(out + a.tile_base + (jj << 8), (u32)tid) = (u64)wl[jj] | ((u64)wh[jj] << 32);
        }
    } else {
        n3_store_direct<R, MODE, i32>(a, out, tid, x);
    }
}

static int n3_env(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

template <int R, int MODE, bool INV, int WPS>
static void n3_launch_w(const N3Params& p, size_t cols, size_t cosets, hipStream_t stream) {
    auto kern = ntt3_pass_kernel<R, MODE, INV, WPS>;
    const size_t lds_bytes = (size_t)(WPS == 2 ? 8 : 4) << N3_TILE_BITS;
    static bool attr_set = false;
    if (!attr_set) {
        HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        attr_set = true;
    }
    static const int col_major = n3_env("OLA_NTT3_COL_MAJOR", 1);
    const size_t tiles = (size_t)1 << (p.log_n - N3_TILE_BITS);
    const bool cm = col_major && tiles <= 65535;
    dim3 grid(cm ? (unsigned)cols : (unsigned)tiles, cm ? (unsigned)tiles : (unsigned)cols, (unsigned)cosets);
    hipLaunchKernelGGL(kern, grid, dim3(N3_THREADS), lds_bytes, stream, p, cm ? 1 : 0);
}

template <int R, int MODE, bool INV>
static void n3_launch_t(const N3Params& p, size_t cols, size_t cosets, hipStream_t stream) {
    static const int wps = n3_env("OLA_NTT3_WPS", 2);
    if (wps == 3) n3_launch_w<R, MODE, INV, 3>(p, cols, cosets, stream);
    else n3_launch_w<R, MODE, INV, 2>(p, cols, cosets, stream);
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
