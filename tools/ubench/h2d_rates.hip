// Host -> device rates of the box, the ceilings behind olavm_amd/csrc/upload.h:
//   pinned      hipHostMalloc'ed source, one hipMemcpyAsync per 64 MB, queued back to back      (the link's rate)
//   pageable    malloc'ed source, hipMemcpyAsync + synchronise per 64 MB                         (round 4's upload path)
//   register    hipHostRegister of the malloc'ed source (time reported), then as pinned, then hipHostUnregister
//   staged      K threads memcpy pieces into a ring of pinned slots, each sent at once on one of S streams in turn
//               (upload.h's path, without a prover)
// hipcc --offload-arch=gfx950 -O3 -pthread -o h2d_rates h2d_rates.hip && ./h2d_rates [GB=4]
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const size_t total = (size_t)((argc > 1 ? atof(argv[1]) : 4.0) * (1u << 30)), chunk = (size_t)64 << 20;
    char* dev = nullptr;
    CK(hipMalloc(&dev, total));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    printf("hardware threads: %u, bytes per run: %.2f GB\n", std::thread::hardware_concurrency(), total / 1e9);
    {   // pinned
        char* h = nullptr;
        double t0 = now();
        CK(hipHostMalloc(&h, total, hipHostMallocDefault));
        const double t_alloc = now() - t0;
        memset(h, 1, total);
        for (int rep = 0; rep < 3; rep++) {
            t0 = now();
            for (size_t o = 0; o < total; o += chunk) CK(hipMemcpyAsync(dev + o, h + o, std::min(chunk, total - o), hipMemcpyHostToDevice, st));
            CK(hipStreamSynchronize(st));
            printf("pinned      %7.1f ms  %6.1f GB/s   (hipHostMalloc of it: %.0f ms)\n", (now() - t0) * 1e3, total / 1e9 / (now() - t0), t_alloc * 1e3);
        }
        CK(hipHostFree(h));
    }
    char* src = (char*)malloc(total);
    memset(src, 2, total);
    for (int rep = 0; rep < 3; rep++) {   // pageable
        const double t0 = now();
        for (size_t o = 0; o < total; o += chunk) {
            CK(hipMemcpyAsync(dev + o, src + o, std::min(chunk, total - o), hipMemcpyHostToDevice, st));
            CK(hipStreamSynchronize(st));
        }
        printf("pageable    %7.1f ms  %6.1f GB/s\n", (now() - t0) * 1e3, total / 1e9 / (now() - t0));
    }
    for (int rep = 0; rep < 2; rep++) {   // register
        double t0 = now();
        const hipError_t e = hipHostRegister(src, total, hipHostRegisterDefault);
        const double t_reg = now() - t0;
        if (e != hipSuccess) { printf("register    hipHostRegister failed: %s\n", hipGetErrorString(e)); (void)hipGetLastError(); break; }
        t0 = now();
        for (size_t o = 0; o < total; o += chunk) CK(hipMemcpyAsync(dev + o, src + o, std::min(chunk, total - o), hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        const double t_copy = now() - t0;
        t0 = now();
        CK(hipHostUnregister(src));
        printf("register    %7.1f ms registering + %7.1f ms copying (%.1f GB/s) + %.1f ms unregistering = %.1f GB/s over all\n", t_reg * 1e3, t_copy * 1e3,
               total / 1e9 / t_copy, (now() - t0) * 1e3, total / 1e9 / (t_reg + t_copy + (now() - t0)));
    }
    {   // staged
        const size_t ring_bytes = (size_t)128 << 20;
        char* ring = nullptr;
        CK(hipHostMalloc(&ring, ring_bytes, hipHostMallocDefault));
        hipStream_t sts[3];
        for (auto& x : sts) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
        struct V { unsigned K, S; size_t mb; };
        const size_t piece0 = (size_t)4 << 20, slots0 = 32, npieces0 = total / piece0;
        for (V v : {V{1, 1, 4}, V{2, 1, 4}, V{4, 1, 4}, V{8, 1, 4}, V{4, 1, 8}, V{4, 1, 16}, V{4, 2, 4}, V{4, 2, 8}, V{4, 2, 16}, V{4, 3, 8}, V{2, 2, 8}, V{8, 2, 8}}) {
            const unsigned K = v.K;
            const size_t piece = v.mb << 20, slots = ring_bytes / piece, npieces = total / piece;
            std::vector<hipEvent_t> ev(slots);
            for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            std::atomic<size_t> next{0};
            std::vector<std::atomic<char>> issued(npieces);
            for (auto& x : issued) x.store(0);
            std::mutex mu, issue_mu;
            std::condition_variable cv;
            size_t completed = 0;
            const double t0 = now();
            std::vector<std::thread> th;
            for (unsigned k = 0; k < K; k++)
                th.emplace_back([&] {
                    for (;;) {
                        const size_t i = next.fetch_add(1);
                        if (i >= npieces) return;
                        { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return completed + slots > i; }); }
                        char* slot = ring + (i % slots) * piece;
                        memcpy(slot, src + i * piece, piece);
                        {
                            std::lock_guard<std::mutex> lk(issue_mu);
                            CK(hipMemcpyAsync(dev + i * piece, slot, piece, hipMemcpyHostToDevice, sts[i % v.S]));
                            CK(hipEventRecord(ev[i % slots], sts[i % v.S]));
                        }
                        issued[i].store(1);
                        { std::lock_guard<std::mutex> lk(mu); cv.notify_all(); }
                    }
                });
            for (size_t i = 0; i < npieces; i++) {
                { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return issued[i].load() != 0; }); }
                CK(hipEventSynchronize(ev[i % slots]));
                { std::lock_guard<std::mutex> lk(mu); completed = i + 1; cv.notify_all(); }
            }
            for (auto& t : th) t.join();
            printf("staged K=%-2u streams=%u piece=%2zu MB  %7.1f ms  %6.1f GB/s\n", K, v.S, v.mb, (now() - t0) * 1e3, total / 1e9 / (now() - t0));
            for (auto& e : ev) CK(hipEventDestroy(e));
        }
        const size_t piece = piece0, slots = slots0, npieces = npieces0;
        // how fast the host side alone is (no DMA): K threads copying into the ring
        for (unsigned K : {1u, 4u, 8u}) {
            std::atomic<size_t> next{0};
            const double t0 = now();
            std::vector<std::thread> th;
            for (unsigned k = 0; k < K; k++)
                th.emplace_back([&] { for (;;) { const size_t i = next.fetch_add(1); if (i >= npieces) return; memcpy(ring + (i % slots) * piece, src + i * piece, piece); } });
            for (auto& t : th) t.join();
            printf("memcpy K=%-2u %7.1f ms  %6.1f GB/s (host copies alone)\n", K, (now() - t0) * 1e3, total / 1e9 / (now() - t0));
        }
        CK(hipHostFree(ring));
    }
    free(src);
    return 0;
}
