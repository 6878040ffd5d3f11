// The ranks of ONE context that spans several GPUs (ola_gpu_init_multi, SURVEY 8(b) Threading: "one ctx spans 1/2/4/8 devices").
//
// The reference's GPU precedent is a single process with process-wide device state (plonky2/field/src/cfft/ntt/mod.rs:14-17,
// 48-50) and its caller proves once per process (client/src/main.rs:174-214); a multi-device context therefore lives in that one
// process: rank r is a worker thread driving GPU devices[r] on its own stream, all ranks run the same prover on the same traces
// (the transcripts agree without exchanging challenges, see ShardInfo) and the exchanges of the coset partition are done HERE,
// by the library, over xGMI:
//
//   all_gather(send, recv, bytes) on rank r
//     1. publish (send, recv) and record event ready[r] on r's stream          -- "my block is complete when this fires"
//     2. host barrier among the rank threads                                    (every rank has published)
//     3. for every peer j: r's stream waits for ready[j], then PULLS j's block with hipMemcpyPeerAsync into recv + j*bytes
//        (xGMI is point to point: the G-1 pulls of a rank travel over G-1 different links at once, and no rank relays)
//     4. record event done[r] on r's stream, host barrier, then r's stream waits for every done[j]
//        -- r's later kernels may overwrite `send` only after all peers have pulled it.
//
// Nothing synchronises a device with the host: the collective is ordered by events on the ranks' own streams
// (OLA_SHARD_STREAM_ORDERED), the host threads only meet each other.  Logical ranks may alias one physical device (tests on a
// one-GPU box, or oversubscription): the pulls are then device-local copies and the event logic is the same.
#pragma once
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <mutex>
#include <string>
#include <vector>

#include "device_ctx.h"

namespace ola {

struct PeerGroup;
struct PeerRank {
    PeerGroup* group = nullptr;
    uint32_t rank = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ready = nullptr, done = nullptr;
    const void* send = nullptr;
    void* recv = nullptr;
    size_t bytes = 0;
};

struct PeerGroup {
    uint32_t world = 1;
    std::vector<PeerRank> ranks;
    std::mutex mu;
    std::condition_variable cv;
    uint32_t arrived = 0;
    uint64_t generation = 0;
    bool failed = false;
    uint64_t exchanges = 0, bytes_moved = 0;   // rank 0's count (observability / tests)

    void reset() { std::lock_guard<std::mutex> lk(mu); arrived = 0; failed = false; }
    // a rank gave up (its prover threw): release everybody who waits, now or later
    void fail() { std::lock_guard<std::mutex> lk(mu); failed = true; cv.notify_all(); }
    void barrier() {
        std::unique_lock<std::mutex> lk(mu);
        if (failed) throw OlaError(-7, "a peer rank of the multi-device context failed");
        const uint64_t gen = generation;
        if (++arrived == world) { arrived = 0; generation++; cv.notify_all(); return; }
        cv.wait(lk, [&] { return generation != gen || failed; });
        if (generation == gen) throw OlaError(-7, "a peer rank of the multi-device context failed");
    }
    ~PeerGroup() {
        for (PeerRank& r : ranks) {
            if (r.ready) (void)hipEventDestroy(r.ready);
            if (r.done) (void)hipEventDestroy(r.done);
        }
    }
};

// ShardInfo::all_gather of a multi-device context; user = the calling rank's PeerRank
inline int32_t peer_all_gather(void* user, const void* send_dev, void* recv_dev, size_t bytes) {
    PeerRank& me = *static_cast<PeerRank*>(user);
    PeerGroup& g = *me.group;
    try {
        me.send = send_dev; me.recv = recv_dev; me.bytes = bytes;
        HIP_CHECK(hipEventRecord(me.ready, me.stream));
        g.barrier();
        for (uint32_t k = 0; k < g.world; k++) {
            const uint32_t j = (me.rank + k) % g.world;          // start with the own block, then walk the ring: spreads the pulls
            const PeerRank& src = g.ranks[j];
            if (src.bytes != bytes) throw OlaError(-7, "multi-device all-gather: ranks disagree on the block size");
            char* dst = static_cast<char*>(recv_dev) + (size_t)j * bytes;
            if (j == me.rank) {
                if (dst != send_dev) HIP_CHECK(hipMemcpyAsync(dst, send_dev, bytes, hipMemcpyDeviceToDevice, me.stream));
                continue;
            }
            HIP_CHECK(hipStreamWaitEvent(me.stream, src.ready, 0));
            if (src.device == me.device) HIP_CHECK(hipMemcpyAsync(dst, src.send, bytes, hipMemcpyDeviceToDevice, me.stream));
            else HIP_CHECK(hipMemcpyPeerAsync(dst, me.device, src.send, src.device, bytes, me.stream));
        }
        HIP_CHECK(hipEventRecord(me.done, me.stream));
        if (me.rank == 0) { g.exchanges++; g.bytes_moved += (size_t)(g.world - 1) * bytes * g.world; }
        g.barrier();
        for (uint32_t j = 0; j < g.world; j++)
            if (j != me.rank) HIP_CHECK(hipStreamWaitEvent(me.stream, g.ranks[j].done, 0));
        return 0;
    } catch (const OlaError&) {
        g.fail();
        return 1;
    }
}

}  // namespace ola
