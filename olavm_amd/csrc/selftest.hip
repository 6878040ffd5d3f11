// Device self-test of the field arithmetic (ola_gpu_selftest): the carry-flag forms of the 128 -> 64 bit reduction (gl.cuh, inline
// assembly with hand-placed wait states) against the plain C++ forms of the same header, on every pair of a table of edge values
// and on `pairs` pseudo-random operand pairs biased towards words whose halves are all ones or all zeros -- where the carries
// live.  A product check an integrator can run once at start-up on a new driver / compiler; the GPU test-suite runs it too.
#include <hip/hip_runtime.h>

#include <vector>

#include "device_ctx.h"
#include "gl.cuh"

namespace ola {

__device__ __forceinline__ u64 st_mix(u64 x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ unsigned st_check(u64 x, u64 y) {
    const u64 want = gl_reduce128(x, y);           // C++ form
    unsigned bad = (gl_canon(gl_reduce128_weak_cc(x, y)) != want) + (gl_reduce128_cc(x, y) != want);
    u64 lo, hi;
    mul_wide(x, y, lo, hi);
    bad += gl_mul(x, y) != gl_reduce128(lo, hi);
    return bad;
}
__global__ __launch_bounds__(256) void field_selftest_kernel(const u64* __restrict__ edges, int nedges, u64 per_thread, unsigned long long* __restrict__ bad) {
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (u64)gridDim.x * blockDim.x;
    unsigned long long local = 0;
    for (u64 k = gid; k < (u64)nedges * nedges; k += nthreads) local += st_check(edges[k / nedges], edges[k % nedges]);
    u64 s = st_mix(gid * 0xD1342543DE82EF95ull + 0xC0FFEEull);
    for (u64 i = 0; i < per_thread; i++) {
        const u64 x = st_mix(s), y = st_mix(s + 1);
        s = y;
        const u64 xx = (i & 7) == 3 ? (x | 0xFFFFFFFF00000000ull) : (i & 7) == 5 ? (x & 0xFFFFFFFFull) : x;
        const u64 yy = (i & 15) == 9 ? (y | 0xFFFFFFFFull) : (i & 15) == 11 ? (y << 32) : y;
        local += st_check(xx, yy);
    }
    if (local) atomicAdd(bad, local);
}

u64 field_selftest(DeviceCtx* ctx, u64 pairs) {
    std::vector<u64> e = {0, 1, 2, 0xFFFFFFFFull, 0x100000000ull, 0x100000001ull, 0xFFFFFFFEull, GL_P - 1, GL_P, GL_P + 1, GL_P - 2, ~0ull, ~0ull - 1,
                          0xFFFFFFFF00000000ull, 0xFFFFFFFEFFFFFFFFull, 0x8000000000000000ull, 0x7FFFFFFFFFFFFFFFull, 0xFFFFFFFF00000002ull,
                          0x00000001FFFFFFFFull, 0xFFFFFFFE00000001ull, 0xFFFFFFFE00000000ull};
    for (int s = 1; s < 64; s++) { e.push_back(1ull << s); e.push_back((1ull << s) - 1); e.push_back(~0ull << s); }
    DevBuf mem(ctx);                              // released on every path, errors included
    u64* d_e = mem.alloc(e.size());
    unsigned long long* d_bad = (unsigned long long*)mem.alloc_bytes(8);
    unsigned long long bad = 0;
    HIP_CHECK(hipMemcpyAsync(d_e, e.data(), e.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_CHECK(hipMemsetAsync(d_bad, 0, 8, ctx->stream));
    const unsigned blocks = 2048;
    hipLaunchKernelGGL(field_selftest_kernel, dim3(blocks), dim3(256), 0, ctx->stream, d_e, (int)e.size(), (pairs + blocks * 256 - 1) / (blocks * 256), d_bad);
    HIP_CHECK(hipMemcpyAsync(&bad, d_bad, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return bad;
}

}  // namespace ola
