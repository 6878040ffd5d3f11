// Micro-benchmark: the memory access pattern of one strided NTT pass (ntt2_pass_kernel<7, N2_STRIDED>) without any
// arithmetic: every thread loads 16 elements that are 2^(lo+3) apart, exchanges through LDS, stores 16 elements 2^lo apart.
// Tells how much of a pass is the access pattern and how much the integer work.
// build: hipcc --offload-arch=gfx950 -O3 -o strided_copy tools/ubench/strided_copy.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;

template <bool LDS>
__global__ __launch_bounds__(256) void pass_copy(const u64* __restrict__ in, u64* __restrict__ out, int L, int lo, size_t col_stride) {
    constexpr int R = 7, K2 = 3, D = 1;
    __shared__ u64 lds[4096 + 256];
    const int tid = threadIdx.x;
    const unsigned blk = blockIdx.x;
    const u64* ip = in + blockIdx.y * col_stride;
    u64* op = out + blockIdx.y * col_stride;
    const int u = tid & 15, rest = tid >> 4, m_low = rest & 7, t = rest >> 3;
    const unsigned ntile = (blk << D) + t, lowblks = 1u << (lo - 4);
    const unsigned lb = ntile & (lowblks - 1), hi = ntile >> (lo - 4);
    const size_t a0 = ((size_t)hi << (lo + R)) + ((size_t)lb << 4) + ((size_t)m_low << lo) + u;
    const size_t js = (size_t)1 << (lo + K2);
    u64 x[16];
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = ip[a0 + j * js];
    if (LDS) {
#pragma unroll
        for (int j = 0; j < 16; j++) lds[((((t << R) + (j << K2) + m_low) << 4) + u) + (((t << R) + (j << K2) + m_low) >> 3)] = x[j];
        __syncthreads();
        const int m_hi = tid >> 4;
#pragma unroll
        for (int tt = 0; tt < 2; tt++)
#pragma unroll
            for (int j2 = 0; j2 < 8; j2++) { const int m = (m_hi << K2) + j2; x[tt * 8 + j2] = lds[((((tt << R) + m) << 4) + u) + (((tt << R) + m) >> 3)]; }
#pragma unroll
        for (int tt = 0; tt < 2; tt++) {
            const unsigned nt = (blk << D) + tt, lb2 = nt & (lowblks - 1), hi2 = nt >> (lo - 4);
            const size_t b0 = ((size_t)hi2 << (lo + R)) + ((size_t)lb2 << 4) + u;
#pragma unroll
            for (int j2 = 0; j2 < 8; j2++) op[b0 + ((size_t)((m_hi << K2) + j2) << lo)] = x[tt * 8 + j2];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 16; j++) op[a0 + j * js] = x[j];
    }
}

__global__ __launch_bounds__(256) void plain_copy(const u64* __restrict__ in, u64* __restrict__ out, size_t n) {
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
    const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(in + i);
    *reinterpret_cast<ulonglong2*>(out + i) = v;
}

int main() {
    const int L = 22, cols = 94;
    const size_t n = (size_t)1 << L, total = n * cols;
    u64 *a, *b;
    hipMalloc(&a, total * 8); hipMalloc(&b, total * 8);
    hipMemset(a, 1, total * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char* name, auto launch) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 10; r++) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        printf("%-34s %7.3f ms  %7.1f GB/s (read+write)\n", name, ms, 2.0 * total * 8 / ms / 1e6);
    };
    time("plain copy 16 B/lane", [&] { hipLaunchKernelGGL(plain_copy, dim3((unsigned)(total / 512)), dim3(256), 0, 0, a, b, total); });
    for (int lo : {15, 8}) {
        char nm[64];
        snprintf(nm, sizeof nm, "strided pass pattern lo=%d no LDS", lo);
        time(nm, [&] { hipLaunchKernelGGL(pass_copy<false>, dim3((unsigned)(n >> 12), cols), dim3(256), 0, 0, a, b, L, lo, n); });
        snprintf(nm, sizeof nm, "strided pass pattern lo=%d via LDS", lo);
        time(nm, [&] { hipLaunchKernelGGL(pass_copy<true>, dim3((unsigned)(n >> 12), cols), dim3(256), 0, 0, a, b, L, lo, n); });
    }
    return 0;
}
