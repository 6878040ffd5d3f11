// GPU self-test + micro-benchmark of the carry-flag reduction in olavm_amd/csrc/gl.cuh (gl_reduce128 / gl_reduce128_weak written
// with explicit carry chains) against the plain C++ forms compiled from the same header with OLA_GL_NO_ASM.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DOLA_GL_NO_ASM -c tests/gpu_glasm_ref.cpp ... (see tests/test_gpu_parity.py)
// Usage: gpu_glasm_selftest [millions of random pairs per launch]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../olavm_amd/csrc/gl.cuh"

using namespace ola;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)

// reference forms, spelled out here so that both live in one translation unit
__device__ __forceinline__ u64 ref_reduce_weak(u64 lo, u64 hi) {
    u64 hh = hi >> 32, hl = hi & GL_EPS;
    u64 t0 = lo - hh;
    if (lo < hh) t0 -= GL_EPS;
    u64 t1 = (hl << 32) - hl;
    u64 t2 = t0 + t1;
    if (t2 < t0) t2 += GL_EPS;
    return t2;
}
__device__ __forceinline__ u64 ref_mul(u64 a, u64 b) {
    u64 lo, hi;
    mul_wide(a, b, lo, hi);
    return gl_canon(ref_reduce_weak(lo, hi));
}
__device__ __forceinline__ u64 splitmix(u64 x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ void check_kernel(const u64* edges, int nedges, u64 seed, u64 per_thread, unsigned long long* bad, u64* first_bad) {
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long local_bad = 0;
    // every pair of edge values as (lo, hi) of the 128-bit input and as factors
    for (u64 k = gid; k < (u64)nedges * nedges; k += (u64)gridDim.x * blockDim.x) {
        const u64 x = edges[k / nedges], y = edges[k % nedges];
        const u64 w = gl_reduce128_weak_cc(x, y), c = gl_reduce128_cc(x, y), r = gl_canon(ref_reduce_weak(x, y));
        if (gl_canon(w) != r || c != r) { if (!local_bad) { first_bad[0] = x; first_bad[1] = y; first_bad[2] = c; first_bad[3] = r; } local_bad++; }
        if (gl_mul(x, y) != ref_mul(x, y)) { if (!local_bad) { first_bad[0] = x; first_bad[1] = y; first_bad[2] = gl_mul(x, y); first_bad[3] = ref_mul(x, y); } local_bad++; }
    }
    u64 s = splitmix(seed ^ (gid * 0xD1342543DE82EF95ull));
    for (u64 i = 0; i < per_thread; i++) {
        const u64 x = splitmix(s), y = splitmix(s + 1);
        s = y;
        // mix in words that are all ones / all zeros in one half: the carries live there
        const u64 xx = (i & 7) == 3 ? (x | 0xFFFFFFFF00000000ull) : (i & 7) == 5 ? (x & 0xFFFFFFFFull) : x;
        const u64 yy = (i & 15) == 9 ? (y | 0xFFFFFFFFull) : (i & 15) == 11 ? (y << 32) : y;
        const u64 r = gl_canon(ref_reduce_weak(xx, yy));
        if (gl_canon(gl_reduce128_weak_cc(xx, yy)) != r || gl_reduce128_cc(xx, yy) != r) { if (!local_bad) { first_bad[0] = xx; first_bad[1] = yy; first_bad[2] = gl_reduce128_cc(xx, yy); first_bad[3] = r; } local_bad++; }
        if (gl_mul(xx, yy) != ref_mul(xx, yy)) local_bad++;
    }
    if (local_bad) atomicAdd(bad, local_bad);
}

// throughput: ILP independent chains of dependent multiplications per thread
template <int ILP, bool ASM>
__global__ void bench_kernel(u64* out, int iters) {
    u64 x[ILP];
    const u64 w = 0x123456789ABCDEFull + threadIdx.x;
#pragma unroll
    for (int k = 0; k < ILP; k++) x[k] = splitmix(threadIdx.x + 64 * k + blockIdx.x);
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < ILP; k++) x[k] = ASM ? gl_mul(x[k], w) : ref_mul(x[k], w);
    }
    u64 acc = 0;
#pragma unroll
    for (int k = 0; k < ILP; k++) acc ^= x[k];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int ILP, bool ASM>
static double run_bench(u64* d_out, int blocks, int iters) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((bench_kernel<ILP, ASM>), dim3(blocks), dim3(256), 0, 0, d_out, 16);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL((bench_kernel<ILP, ASM>), dim3(blocks), dim3(256), 0, 0, d_out, iters);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return (double)blocks * 256 * iters * ILP / (ms * 1e-3);   // multiplications per second
}

int main(int argc, char** argv) {
    const u64 millions = argc > 1 ? strtoull(argv[1], nullptr, 10) : 2000;
    const u64 P = GL_P;
    std::vector<u64> edges = {0, 1, 2, 0xFFFFFFFFull, 0x100000000ull, 0x100000001ull, 0xFFFFFFFEull, P - 1, P, P + 1, P - 2, ~0ull, ~0ull - 1,
                              0xFFFFFFFF00000000ull, 0xFFFFFFFEFFFFFFFFull, 0x8000000000000000ull, 0x7FFFFFFFFFFFFFFFull, 0x00000000FFFFFFFEull,
                              0xFFFFFFFF00000002ull, 0x0000000100000000ull - 2, 0x00000001FFFFFFFFull, 0xFFFFFFFE00000001ull, 0xFFFFFFFE00000000ull};
    for (int s = 1; s < 64; s++) { edges.push_back(1ull << s); edges.push_back((1ull << s) - 1); edges.push_back(~0ull << s); }
    u64 *d_edges, *d_first, *d_out;
    unsigned long long* d_bad;
    CK(hipMalloc(&d_edges, edges.size() * 8));
    CK(hipMemcpy(d_edges, edges.data(), edges.size() * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_bad, 8)); CK(hipMemset(d_bad, 0, 8));
    CK(hipMalloc(&d_first, 32)); CK(hipMemset(d_first, 0, 32));
    const int blocks = 2048, threads = 256;
    const u64 per_thread = millions * 1000000ull / ((u64)blocks * threads);
    hipLaunchKernelGGL(check_kernel, dim3(blocks), dim3(threads), 0, 0, d_edges, (int)edges.size(), 0xC0FFEEull, per_thread, d_bad, d_first);
    CK(hipDeviceSynchronize());
    unsigned long long bad = 0; u64 first[4];
    CK(hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(first, d_first, 32, hipMemcpyDeviceToHost));
    printf("edge pairs %zu x %zu, random (lo, hi) / (a, b) pairs %llu: mismatches %llu\n", edges.size(), edges.size(),
           (unsigned long long)(per_thread * blocks * threads), bad);
    if (bad) printf("first mismatch: x=%016llx y=%016llx got=%016llx want=%016llx\n", first[0], first[1], first[2], first[3]);
    CK(hipMalloc(&d_out, (size_t)8192 * 256 * 8));
    const int bb = 256 * 8 * 4;   // 8 workgroups of 4 waves per CU
    printf("modular multiplications per second (G/s), 256 CUs, ILP chains per thread: carry-flag form / plain C++ form\n");
    printf("  ILP 1: %7.1f / %7.1f\n", run_bench<1, true>(d_out, bb, 4096) / 1e9, run_bench<1, false>(d_out, bb, 4096) / 1e9);
    printf("  ILP 2: %7.1f / %7.1f\n", run_bench<2, true>(d_out, bb, 4096) / 1e9, run_bench<2, false>(d_out, bb, 4096) / 1e9);
    printf("  ILP 4: %7.1f / %7.1f\n", run_bench<4, true>(d_out, bb, 2048) / 1e9, run_bench<4, false>(d_out, bb, 2048) / 1e9);
    printf("  ILP 8: %7.1f / %7.1f\n", run_bench<8, true>(d_out, bb, 1024) / 1e9, run_bench<8, false>(d_out, bb, 1024) / 1e9);
    printf(bad ? "FAILED\n" : "all ok\n");
    return bad ? 1 : 0;
}
