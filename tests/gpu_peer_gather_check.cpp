// The library's own all-gather (olavm_amd/csrc/peer_group.h) by itself, on the GPU box: `world` rank threads with one stream each --
// on the devices the box has, aliased round-robin when it has fewer than `world` -- gather blocks of several sizes; every rank's
// result is compared with the expected concatenation, a kernel-free reuse pattern (each rank overwrites its send block right after
// the exchange) checks the "done" half of the protocol, and the per-exchange latency (small blocks) and rate (large blocks) are
// printed -- the latency is what the 2 / 4 / 8-GPU projection of bench.py assumes 30 us for.
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -pthread -o gpu_peer_gather_check tests/gpu_peer_gather_check.cpp && ./gpu_peer_gather_check 8
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "../olavm_amd/csrc/peer_group.h"

using namespace ola;

int main(int argc, char** argv) {
    const uint32_t world = argc > 1 ? (uint32_t)atoi(argv[1]) : 8;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { printf("no device\n"); return 2; }
    PeerGroup g;
    g.world = world;
    g.ranks.resize(world);
    for (uint32_t r = 0; r < world; r++) {
        PeerRank& pr = g.ranks[r];
        pr.group = &g; pr.rank = r; pr.device = (int)(r % (uint32_t)ndev);
        HIP_CHECK(hipSetDevice(pr.device));
        HIP_CHECK(hipStreamCreateWithFlags(&pr.stream, hipStreamNonBlocking));
        HIP_CHECK(hipEventCreateWithFlags(&pr.ready, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&pr.done, hipEventDisableTiming));
    }
    for (uint32_t a = 0; a < world; a++)
        for (uint32_t b = 0; b < world; b++)
            if (g.ranks[a].device != g.ranks[b].device) {
                HIP_CHECK(hipSetDevice(g.ranks[a].device));
                const hipError_t e = hipDeviceEnablePeerAccess(g.ranks[b].device, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) HIP_CHECK(e);
                (void)hipGetLastError();
            }
    int failures = 0;
    const size_t sizes[] = {512, 4096, (size_t)1 << 20, (size_t)64 << 20};
    for (size_t bytes : sizes) {
        const int rounds = bytes <= 4096 ? 2000 : (bytes <= ((size_t)1 << 20) ? 200 : 20);
        std::vector<double> secs(world, 0.0);
        std::vector<int> bad(world, 0);
        std::vector<std::thread> ts;
        for (uint32_t r = 0; r < world; r++)
            ts.emplace_back([&, r] {
                PeerRank& me = g.ranks[r];
                (void)hipSetDevice(me.device);
                unsigned char *send = nullptr, *recv = nullptr;
                if (hipMalloc(&send, bytes) != hipSuccess || hipMalloc(&recv, bytes * world) != hipSuccess) { bad[r] = 1; g.fail(); return; }
                std::vector<unsigned char> host(bytes * world);
                // correctness: three exchanges with changing contents; the send block is overwritten right after each exchange
                for (int it = 0; it < 3 && !bad[r]; it++) {
                    (void)hipMemsetAsync(send, (int)(17 * r + it + 1), bytes, me.stream);
                    if (peer_all_gather(&me, send, recv, bytes) != 0) { bad[r] = 1; break; }
                    (void)hipMemsetAsync(send, 0xEE, bytes, me.stream);          // must not reach a peer that is still pulling
                    (void)hipMemcpyAsync(host.data(), recv, bytes * world, hipMemcpyDeviceToHost, me.stream);
                    (void)hipStreamSynchronize(me.stream);
                    for (uint32_t j = 0; j < world; j++)
                        for (size_t k = 0; k < bytes; k += (bytes > 4096 ? 4093 : 1))
                            if (host[j * bytes + k] != (unsigned char)(17 * j + it + 1)) { bad[r] = 1; break; }
                }
                // timing
                g.barrier();
                const auto t0 = std::chrono::steady_clock::now();
                for (int it = 0; it < rounds && !bad[r]; it++)
                    if (peer_all_gather(&me, send, recv, bytes) != 0) bad[r] = 1;
                (void)hipStreamSynchronize(me.stream);
                secs[r] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                (void)hipFree(send);
                (void)hipFree(recv);
            });
        for (auto& t : ts) t.join();
        double worst = 0;
        for (uint32_t r = 0; r < world; r++) { failures += bad[r]; if (secs[r] > worst) worst = secs[r]; }
        printf("world %u (%d device%s), block %zu bytes: %s, %.1f us per exchange, %.2f GB/s received per rank\n", world, ndev, ndev == 1 ? ", ranks aliased" : "s",
               bytes, failures ? "MISMATCH" : "gathered data correct", worst / rounds * 1e6, (double)bytes * (world - 1) * rounds / worst / 1e9);
        g.reset();
    }
    printf(failures ? "gpu_peer_gather_check: FAILED\n" : "gpu_peer_gather_check: all ok\n");
    return failures ? 1 : 0;
}
