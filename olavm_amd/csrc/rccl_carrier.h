// RCCL as the carrier of a multi-device context's exchanges (ola_gpu_init_multi, OLA_COLLECTIVE=rccl).
//
// The coset partition's collective is an all-gather of device buffers (ShardInfo::all_gather).  A multi-device context has two
// ways of moving the bytes: the library's own event-ordered peer pulls (peer_group.h, the default) and -- this file -- RCCL:
// one communicator per rank from ncclCommInitAll over the context's devices, ncclAllGather on the rank's own stream.  RCCL is
// NOT a link dependency: librccl.so is dlopen'ed when the carrier is asked for, so a host without RCCL (or a one-GPU box) loads
// libola_gpu.so as before; the function types come from <rccl/rccl.h>, so a signature drift is a compile error here.
//
// Threading: rank r's worker thread (prove_with_traces_multi) calls ncclAllGather on ITS communicator -- the usage RCCL documents
// for one thread per device (rccl.h: collectives on different communicators "must be called by different threads/processes or
// use ncclGroupStart/ncclGroupEnd").  Before every collective the rank threads meet at the group's host barrier: a rank whose
// prover threw never leaves its peers waiting inside a device-side collective (the barrier throws for everybody instead).
// A rank whose ncclAllGather itself fails AFTER the barrier has peers that already enqueued theirs: it aborts every
// communicator of the group (ncclCommAbort releases device-side waits), and the group refuses further use.
// Ordering: the collective is enqueued on the rank's stream, after the kernels that produced `send`, before the ones that read
// `recv` (OLA_SHARD_STREAM_ORDERED) -- no host synchronisation, as with the peer carrier.
//
// Refusals (the context then keeps the peer carrier and says why, ola_gpu_collective): librccl.so not found, a symbol missing,
// ncclCommInitAll failing -- which is what happens when logical ranks alias one physical device ("Duplicate GPU detected").
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
// The declarations only (nothing of RCCL is linked).  On a ROCm install without the RCCL development headers the five entry
// points used here are declared locally, as rccl.h 2.x declares them, so that the library still builds (and still refuses the
// carrier at run time when librccl.so is absent).
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1 } ncclDataType_t;
ncclResult_t ncclCommInitAll(ncclComm_t* comm, int ndev, const int* devlist);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclCommAbort(ncclComm_t comm);
ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream);
const char* ncclGetErrorString(ncclResult_t result);
ncclResult_t ncclGetVersion(int* version);
}
#endif

#include <atomic>

#include <memory>
#include <string>
#include <vector>

#include "peer_group.h"

namespace ola {

struct RcclApi {
    void* so = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;        // optional: absent in very old builds
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    // the process-wide handle; nullptr + `why` when RCCL cannot be used
    static RcclApi* get(std::string& why) {
        static RcclApi api;
        static std::string err;
        static const bool ok = [] {
            for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                api.so = dlopen(name, RTLD_NOW | RTLD_LOCAL);
                if (api.so) break;
            }
            if (!api.so) { const char* e = dlerror(); err = std::string("librccl.so could not be loaded: ") + (e ? e : "?"); return false; }
            auto sym = [](const char* n) { void* p = dlsym(api.so, n); if (!p) err += std::string(err.empty() ? "" : ", ") + n; return p; };
            api.CommInitAll = reinterpret_cast<decltype(api.CommInitAll)>(sym("ncclCommInitAll"));
            api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
            api.CommAbort = reinterpret_cast<decltype(api.CommAbort)>(dlsym(api.so, "ncclCommAbort"));
            api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
            api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
            api.GetVersion = reinterpret_cast<decltype(api.GetVersion)>(sym("ncclGetVersion"));
            if (!err.empty()) { err = "librccl.so lacks " + err; return false; }
            return true;
        }();
        if (!ok) { why = err; return nullptr; }
        return &api;
    }
};

struct RcclGroup;
struct RcclRank {
    RcclGroup* group = nullptr;
    PeerGroup* meet = nullptr;     // host barrier + failure release of the rank threads
    uint32_t rank = 0;
    hipStream_t stream = nullptr;
};
struct RcclGroup {
    RcclApi* api = nullptr;
    std::vector<ncclComm_t> comms;
    std::vector<RcclRank> ranks;
    int version = 0;
    std::atomic<bool> dead{false};          // a collective failed half-way: the communicators were aborted
    void abort_all() {
        if (dead.exchange(true)) return;
        if (api && api->CommAbort) for (ncclComm_t c : comms) if (c) (void)api->CommAbort(c);
    }
    ~RcclGroup() {
        if (api && !dead.load()) for (ncclComm_t c : comms) if (c) (void)api->CommDestroy(c);
    }
};

// One communicator per device of the context.  Returns nullptr and the reason when RCCL cannot carry this context.
inline std::unique_ptr<RcclGroup> rccl_group_create(const std::vector<int>& devices, const std::vector<hipStream_t>& streams, PeerGroup* meet, std::string& why) {
    for (size_t a = 0; a < devices.size(); a++)
        for (size_t b = a + 1; b < devices.size(); b++)
            if (devices[a] == devices[b]) {
                why = "ranks " + std::to_string(a) + " and " + std::to_string(b) + " share device " + std::to_string(devices[a]) +
                      ": RCCL needs one physical GPU per rank (ncclCommInitAll: duplicate GPU)";
                return nullptr;
            }
    RcclApi* api = RcclApi::get(why);
    if (!api) return nullptr;
    std::unique_ptr<RcclGroup> g(new RcclGroup());
    g->api = api;
    g->comms.assign(devices.size(), nullptr);
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) { prev = -1; (void)hipGetLastError(); }
    const ncclResult_t rc = api->CommInitAll(g->comms.data(), (int)devices.size(), devices.data());
    if (prev >= 0) (void)hipSetDevice(prev);
    (void)hipGetLastError();
    if (rc != ncclSuccess) {
        why = std::string("ncclCommInitAll over ") + std::to_string(devices.size()) + " device(s) failed: " + api->GetErrorString(rc);
        g->comms.clear();
        return nullptr;
    }
    (void)api->GetVersion(&g->version);
    g->ranks.resize(devices.size());
    for (size_t r = 0; r < devices.size(); r++) { g->ranks[r].group = g.get(); g->ranks[r].meet = meet; g->ranks[r].rank = (uint32_t)r; g->ranks[r].stream = streams[r]; }
    return g;
}

// ShardInfo::all_gather over RCCL; user = the calling rank's RcclRank
inline int32_t rccl_all_gather(void* user, const void* send_dev, void* recv_dev, size_t bytes) {
    RcclRank& me = *static_cast<RcclRank*>(user);
    RcclGroup& g = *me.group;
    try {
        if (g.dead.load()) throw OlaError(-5, "the RCCL communicators of this context were aborted after a failed collective");
        if (me.meet) me.meet->barrier();          // everybody is about to enqueue the same collective (or somebody failed: throws)
        const ncclResult_t rc = g.api->AllGather(send_dev, recv_dev, bytes, ncclUint8, g.comms[me.rank], me.stream);
        if (rc != ncclSuccess) {
            g.abort_all();                        // the peers are past the barrier and have enqueued theirs: release them
            throw OlaError(-5, std::string("ncclAllGather: ") + g.api->GetErrorString(rc));
        }
        if (me.meet && me.rank == 0) { me.meet->exchanges++; me.meet->bytes_moved += (size_t)(g.comms.size() - 1) * bytes * g.comms.size(); }
        return 0;
    } catch (const OlaError&) {
        if (me.meet) me.meet->fail();
        return 1;
    }
}

}  // namespace ola
