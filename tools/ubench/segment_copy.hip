// Micro-benchmark: what HBM makes of the access patterns a two-pass natural-order NTT would need (no arithmetic).
//   seg<S>:     a 256-thread workgroup moves a tile of 8192 elements = (8192 / S) rows of S consecutive u64, rows 2^lo apart,
//               reading and writing in place (the strided pass pattern of ntt3.hip at S = 16; S = 8 / 4 = 64- / 32-byte segments);
//   scatter<M>: reads 8192 consecutive elements and writes element t of tile T to bitrev13(t) * 2^(L-13) + bitrev(T): every
//               store lane hits a different 128-byte line, 16 workgroups complete each line.  M = 0: tiles in launch order;
//               M = 1: the 16 workgroups that share lines are launched 8 apart (same XCD, same time), so that the XCD's L2 can
//               merge their partial lines before they leave for HBM.
// build: hipcc --offload-arch=gfx950 -O3 -o segment_copy tools/ubench/segment_copy.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;

template <int S>
__global__ __launch_bounds__(256, 2) void seg(const u64* __restrict__ in, u64* __restrict__ out, int L, int lo, size_t col_stride) {
    constexpr int LS = (S == 32) ? 5 : (S == 16) ? 4 : (S == 8) ? 3 : 2;
    constexpr int R = 13 - LS;   // rows = 2^R
    const int tid = threadIdx.x;
    const unsigned tile = blockIdx.x;
    const int mb = lo - LS;
    const size_t mid = tile & ((1u << mb) - 1), hi = tile >> mb;
    const size_t base = (hi << (lo + R)) + (mid << LS) + blockIdx.y * col_stride;
    // thread: s = tid % S, row_low = tid / S ; register j: row = j * (256 / S) + row_low
    const int s = tid & (S - 1), rl = tid >> LS;
    u64 x[32];
#pragma unroll
    for (int j = 0; j < 32; j++) x[j] = in[base + ((size_t)(j * (256 / S) + rl) << lo) + s];
#pragma unroll
    for (int j = 0; j < 32; j++) out[base + ((size_t)(j * (256 / S) + rl) << lo) + s] = x[j] + 1;
}

__device__ __forceinline__ unsigned brev(unsigned x, int bits) { return __brev(x) >> (32 - bits); }

template <int M>
__global__ __launch_bounds__(256, 2) void scatter(const u64* __restrict__ in, u64* __restrict__ out, int L, size_t col_stride) {
    const int tid = threadIdx.x;
    const int tb = L - 13;   // tile bits
    unsigned b = blockIdx.x, T;
    if (M == 0) T = b;
    else {   // b = [low3 of T][top4 of T][the rest]
        const unsigned low3 = b & 7, top4 = (b >> 3) & 15, rest = b >> 7;
        T = low3 | (rest << 3) | (top4 << (tb - 4));
    }
    const u64* ip = in + blockIdx.y * col_stride + ((size_t)T << 13);
    u64* op = out + blockIdx.y * col_stride + brev(T, tb);
    u64 x[32];
#pragma unroll
    for (int j = 0; j < 32; j++) x[j] = ip[(j << 8) + tid];
#pragma unroll
    for (int j = 0; j < 32; j++) op[(size_t)brev((j << 8) + tid, 13) << tb] = x[j] + 1;
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
        for (int r = 0; r < 5; r++) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("%-44s %7.3f ms  %7.1f GB/s (read+write)\n", name, ms, 2.0 * total * 8 / ms / 1e6);
        fflush(stdout);
    };
    dim3 g((unsigned)(n >> 13), cols);
    for (int lo : {11}) {
        char nm[80];
        snprintf(nm, sizeof nm, "tile rows x 32 elements (256 B), lo=%d", lo);
        time(nm, [&] { hipLaunchKernelGGL(seg<32>, g, dim3(256), 0, 0, a, b, L, lo, n); });
        snprintf(nm, sizeof nm, "tile rows x 16 elements (128 B), lo=%d", lo);
        time(nm, [&] { hipLaunchKernelGGL(seg<16>, g, dim3(256), 0, 0, a, b, L, lo, n); });
        snprintf(nm, sizeof nm, "tile rows x 8 elements (64 B), lo=%d", lo);
        time(nm, [&] { hipLaunchKernelGGL(seg<8>, g, dim3(256), 0, 0, a, b, L, lo, n); });
        snprintf(nm, sizeof nm, "tile rows x 4 elements (32 B), lo=%d", lo);
        time(nm, [&] { hipLaunchKernelGGL(seg<4>, g, dim3(256), 0, 0, a, b, L, lo, n); });
    }
    time("contiguous read, scattered 8 B writes", [&] { hipLaunchKernelGGL(scatter<0>, g, dim3(256), 0, 0, a, b, L, n); });
    time("same, line-sharing workgroups on one XCD", [&] { hipLaunchKernelGGL(scatter<1>, g, dim3(256), 0, 0, a, b, L, n); });
    return 0;
}
