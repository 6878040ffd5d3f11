// Host-only check of the meeting point of a multi-device context's rank threads (olavm_amd/csrc/peer_group.h): the barrier the
// ranks meet at twice per exchange, and what happens when one of them gives up.  No GPU is touched: only PeerGroup::barrier /
// fail / reset run here (the copies and events of peer_all_gather need devices and are covered by tests/test_gpu_multi.py).
//   hipcc -O1 -std=c++17 -pthread -o host_peer_group_check tests/host_peer_group_check.cpp && ./host_peer_group_check
#include <atomic>
#include <cstdio>
#include <thread>
#include <vector>

#include "../olavm_amd/csrc/peer_group.h"

using namespace ola;

static int failures = 0;
#define EXPECT(c) do { if (!(c)) { printf("FAILED: %s (line %d)\n", #c, __LINE__); failures++; } } while (0)

int main() {
    for (uint32_t world : {2u, 4u, 8u}) {
        // 1. every rank passes the same number of barriers; a shared counter proves nobody runs ahead by more than one phase
        {
            PeerGroup g;
            g.world = world;
            std::atomic<long> arrived{0};
            std::atomic<int> violations{0};
            const int rounds = 2000;
            std::vector<std::thread> ts;
            for (uint32_t r = 0; r < world; r++)
                ts.emplace_back([&, r] {
                    for (int i = 0; i < rounds; i++) {
                        arrived.fetch_add(1);
                        g.barrier();
                        if (arrived.load() < (long)(i + 1) * world) violations.fetch_add(1);   // somebody left before all had arrived
                    }
                });
            for (auto& t : ts) t.join();
            EXPECT(violations.load() == 0);
            EXPECT(arrived.load() == (long)rounds * world);
        }
        // 2. one rank fails mid-way: everybody else is released from the barrier with an error instead of waiting for ever,
        //    and after reset() the group works again
        {
            PeerGroup g;
            g.world = world;
            std::atomic<int> released{0}, completed{0};
            std::vector<std::thread> ts;
            for (uint32_t r = 0; r < world; r++)
                ts.emplace_back([&, r] {
                    try {
                        for (int i = 0; i < 100; i++) {
                            if (r == world - 1 && i == 37) { g.fail(); return; }      // this rank's prover threw
                            g.barrier();
                        }
                        completed.fetch_add(1);
                    } catch (const OlaError& e) {
                        if (e.code == -7) released.fetch_add(1);
                    }
                });
            for (auto& t : ts) t.join();
            EXPECT(completed.load() == 0);
            EXPECT(released.load() == (int)world - 1);
            bool threw = false;
            try { g.barrier(); } catch (const OlaError&) { threw = true; }           // still failed: refuses at once
            EXPECT(threw);
            g.reset();
            std::atomic<int> ok{0};
            std::vector<std::thread> t2;
            for (uint32_t r = 0; r < world; r++) t2.emplace_back([&] { try { for (int i = 0; i < 50; i++) g.barrier(); ok.fetch_add(1); } catch (...) {} });
            for (auto& t : t2) t.join();
            EXPECT(ok.load() == (int)world);
        }
        // 3. a failure while the others are NOT at the barrier yet: they see it when they arrive
        {
            PeerGroup g;
            g.world = world;
            g.fail();
            std::atomic<int> released{0};
            std::vector<std::thread> ts;
            for (uint32_t r = 0; r + 1 < world; r++)
                ts.emplace_back([&] { try { g.barrier(); } catch (const OlaError&) { released.fetch_add(1); } });
            for (auto& t : ts) t.join();
            EXPECT(released.load() == (int)world - 1);
        }
    }
    printf(failures ? "host_peer_group_check: %d FAILED\n" : "host_peer_group_check: all ok\n", failures);
    return failures ? 1 : 0;
}
