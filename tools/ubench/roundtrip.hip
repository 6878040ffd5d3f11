// Host <-> device round trip of the prover's small-table steps: [k tiny kernels + a 512-byte read-back] then wait, with
// hipStreamSynchronize against a spin on hipStreamQuery; and the back-to-back launch rate of tiny kernels.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/roundtrip tools/ubench/roundtrip.hip && /tmp/roundtrip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void tiny(unsigned long long* p) { if (threadIdx.x == 0) p[blockIdx.x] += 1; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    unsigned long long* d; hipMalloc(&d, 4096);
    hipMemset(d, 0, 4096);
    std::vector<unsigned long long> h(64);
    unsigned long long* hp; hipHostMalloc(&hp, 512, hipHostMallocDefault);
    for (int i = 0; i < 100; i++) { hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d); } hipStreamSynchronize(s);
    for (int k : {1, 4, 16}) {
        for (int mode = 0; mode < 4; mode++) {
            const int reps = 300;
            double t0 = now();
            for (int r = 0; r < reps; r++) {
                for (int i = 0; i < k; i++) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d);
                hipMemcpyAsync((mode & 2) ? hp : h.data(), d, 512, hipMemcpyDeviceToHost, s);
                if (mode & 1) { while (hipStreamQuery(s) == hipErrorNotReady) {} }
                else hipStreamSynchronize(s);
            }
            double dt = (now() - t0) / reps;
            printf("%2d kernels + 512 B read-back (%s host buffer), wait by %-22s %8.2f us per round trip\n", k, (mode & 2) ? "pinned  " : "pageable",
                   (mode & 1) ? "hipStreamQuery spin:" : "hipStreamSynchronize:", dt);
        }
    }
    {
        const int n = 2000;
        double t0 = now();
        for (int i = 0; i < n; i++) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d);
        double t1 = now();
        hipStreamSynchronize(s);
        double t2 = now();
        printf("%d tiny kernels back to back: enqueue %.2f us each, executed in %.2f us each\n", n, (t1 - t0) / n, (t2 - t0) / n);
    }
    {   // upload from pageable / pinned + kernel + wait
        for (int pinned = 0; pinned < 2; pinned++) {
            const int reps = 300;
            double t0 = now();
            for (int r = 0; r < reps; r++) {
                hipMemcpyAsync(d, pinned ? hp : h.data(), 512, hipMemcpyHostToDevice, s);
                hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d);
            }
            hipStreamSynchronize(s);
            printf("512 B upload from %s memory + kernel, no wait between: %.2f us each\n", pinned ? "pinned" : "pageable", (now() - t0) / reps);
        }
    }
    return 0;
}
