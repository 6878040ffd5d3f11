// Per-kernel timing of the third-generation NTT passes on the headline shape (94 columns x 2^22), tables filled with random
// words (the arithmetic does not care).  Knobs through the environment: OLA_NTT3_WPS=2|3, OLA_NTT3_COL_MAJOR=0|1.
// build (GPU box): hipcc --offload-arch=gfx950 -O2 -std=c++17 -c -o /tmp/b.o tests/gpu_ntt3_bench.cpp && hipcc -o tests/gpu_ntt3_bench /tmp/b.o olavm_amd/lib/obj/ntt3.o
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../olavm_amd/csrc/device_ctx.h"
#include "../olavm_amd/csrc/ntt3.h"
using namespace ola;
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(2); } } while (0)

__global__ void fill(u64* p, size_t n, u64 seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        u64 z = seed + i * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        p[i] = (z ^ (z >> 31)) >> 1;
    }
}

int main(int argc, char** argv) {
    const int L = argc > 1 ? atoi(argv[1]) : 22;
    const size_t cols = argc > 2 ? atoi(argv[2]) : 94, n = (size_t)1 << L;
    u64 *a, *b, *tw, *ptw, *dig;
    CK(hipMalloc(&a, cols * n * 8)); CK(hipMalloc(&b, cols * n * 8)); CK(hipMalloc(&tw, 8192 * 8)); CK(hipMalloc(&ptw, n * 8)); CK(hipMalloc(&dig, 8192 * 8));
    fill<<<4096, 256>>>(a, cols * n, 1); fill<<<4096, 256>>>(b, cols * n, 2); fill<<<64, 256>>>(tw, 8192, 3); fill<<<4096, 256>>>(ptw, n, 4); fill<<<64, 256>>>(dig, 8192, 5);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Case { const char* name; int R, mode, lo; bool pre, inplace; };
    const Case cases[] = {{"strided R=5", 5, N3_STRIDED, L - 5, false, false}, {"strided R=8", 8, N3_STRIDED, L - 8, false, false}, {"strided R=8 middle", 8, N3_STRIDED, 9, false, true},
                          {"strided R=5 middle", 5, N3_STRIDED, 9, false, true},
                          {"strided R=9", 9, N3_STRIDED, L - 9, false, false}, {"strided R=9 + coset", 9, N3_STRIDED, L - 9, true, false},
                          {"contiguous R=13", 13, N3_LAST_BITREV, 0, false, true}, {"natural R=9", 9, N3_LAST_NATURAL, 0, false, false}};
    for (const Case& c : cases) {
        N3Params p = {};
        p.log_n = L; p.lo = c.lo; p.in = a; p.out = c.inplace ? a : b; p.in_col_stride = p.out_col_stride = n; p.tw = tw; p.ptw = ptw; p.sc_dig = c.pre ? dig : nullptr;
        auto run = [&] { ntt3_launch(p, c.R, c.mode, false, cols, 1, 0); };
        run(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < 3; r++) run();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
        printf("%-22s %7.3f ms  %7.1f GB/s (16 B per element)\n", c.name, ms, 16.0 * n * cols / ms / 1e6);
    }
    return 0;
}
