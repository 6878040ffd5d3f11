// Micro-benchmark for the cold-start question: what does hipMalloc of dirty VRAM cost, does it run concurrently from two
// threads, and does a kernel stream keep running while another thread is inside hipMalloc?
// build: hipcc --offload-arch=gfx950 -O2 -o malloc_overlap tools/ubench/malloc_overlap.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void spin(unsigned long long* p, size_t n, int iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long v = p[i % n];
    for (int k = 0; k < iters; k++) v = v * 6364136223846793005ull + 1442695040888963407ull;
    p[i % n] = v;
}
int main() {
    const size_t GB = 1ull << 30;
    void* p; double t;
    // dirty the memory first: allocate most of it, write, free
    t = now(); hipMalloc(&p, 200 * GB); printf("first hipMalloc(200 GB) on this process: %.0f ms\n", (now() - t) * 1e3);
    hipMemset(p, 1, 200 * GB); hipDeviceSynchronize();
    t = now(); hipFree(p); printf("hipFree(200 GB): %.0f ms\n", (now() - t) * 1e3);
    t = now(); hipMalloc(&p, 24 * GB); printf("hipMalloc(24 GB) of dirty memory: %.0f ms\n", (now() - t) * 1e3);
    void* q[2];
    t = now();
    { std::thread a([&] { hipSetDevice(0); hipMalloc(&q[0], 12 * GB); }), b([&] { hipSetDevice(0); hipMalloc(&q[1], 12 * GB); }); a.join(); b.join(); }
    printf("two threads x hipMalloc(12 GB): %.0f ms\n", (now() - t) * 1e3);
    // a busy stream next to a hipMalloc in another thread
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    unsigned long long* w; hipMalloc(&w, 1 << 24);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto burst = [&](const char* name) {
        hipEventRecord(e0, s);
        for (int k = 0; k < 50; k++) hipLaunchKernelGGL(spin, dim3(4096), dim3(256), 0, s, w, (size_t)1 << 21, 2000);
        hipEventRecord(e1, s);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s: 50 kernels in %.1f ms\n", name, ms);
    };
    burst("kernels alone");
    void* r;
    double tm = 0;
    std::thread m([&] { hipSetDevice(0); double t0 = now(); hipMalloc(&r, 24 * GB); tm = now() - t0; });
    burst("kernels next to hipMalloc(24 GB)");
    m.join();
    printf("that hipMalloc took %.0f ms\n", tm * 1e3);
    // stream-ordered allocation
    hipMemPool_t pool; hipDeviceGetDefaultMemPool(&pool, 0);
    unsigned long long thr = ~0ull; hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr);
    void* a2; t = now(); hipMallocAsync(&a2, 24 * GB, s); hipStreamSynchronize(s); printf("hipMallocAsync(24 GB): %.0f ms\n", (now() - t) * 1e3);
    hipFreeAsync(a2, s); hipStreamSynchronize(s);
    t = now(); hipMallocAsync(&a2, 24 * GB, s); hipStreamSynchronize(s); printf("hipMallocAsync(24 GB) again (pooled): %.0f ms\n", (now() - t) * 1e3);
    return 0;
}
