// C ABI of the backend (include/ola_gpu.h) -- the main translation unit of libola_gpu.so (lookup.hip and the generated
// quotient kernels are compiled separately).
// Everything below the `extern "C"` layer is C++/HIP; errors are caught here and turned into status codes.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <memory>
#include <thread>
#include <string>
#include <vector>

#include "../../include/ola_gpu.h"
#include "device_ctx.h"
#include "gl.cuh"
#include "poseidon_host.h"
#include "lookup.h"
#include "peer_group.h"
#include "rccl_carrier.h"

// unity build: device code shares the __constant__ Poseidon tables
#include "ntt.hip"
#include "ntt2.hip"
#include "merkle.hip"
#include "batch.hip"
#include "fri.hip"
#include "stark.hip"
#include "selftest.hip"

using namespace ola;

static thread_local std::string g_last_error;

struct OlaCtx {
    DeviceCtx dev;
    NttTables* tables = nullptr;
    OlaGpuConfig cfg;
    std::vector<uint8_t> pending_proof;   // an AllProof that did not fit the caller's buffer (ola_take_pending_proof)
    bool first_proof_done = false;        // the first whole proof reserves its large blocks on a helper thread (prove_all)
    // A context that spans several GPUs (ola_gpu_init_multi): this object is rank 0, `peers` are ranks 1..n-1 (owned), `group`
    // their meeting point (peer_group.h).  Empty / null for a single-device context.
    std::vector<OlaCtx*> peers;
    std::unique_ptr<PeerGroup> group;
    // who moves the bytes of the partition's exchanges (ola_gpu_collective): the library's peer pulls, or RCCL when OLA_COLLECTIVE=rccl
    // asked for it at ola_gpu_init_multi and RCCL could be set up over the context's devices (rccl_carrier.h); else the reason why not
    std::unique_ptr<RcclGroup> rccl;
    std::string collective_note;
    ~OlaCtx() {
        if (tables) ntt_tables_destroy(tables);
        // every rank's stream drains, then the communicators go (they enqueued on those streams), then the ranks with their streams
        for (OlaCtx* p : peers) {
            (void)hipSetDevice(p->dev.device);
            (void)hipStreamSynchronize(p->dev.stream);
        }
        if (!peers.empty()) { (void)hipSetDevice(dev.device); if (dev.stream) (void)hipStreamSynchronize(dev.stream); }
        rccl.reset();
        for (OlaCtx* p : peers) {
            (void)hipSetDevice(p->dev.device);
            delete p;
        }
    }
};

// Kernel launches report configuration errors (grid / LDS limits, missing code object) only through the sticky "last
// error"; it is read once per entry point so that such a failure surfaces as OLA_E_HIP instead of a silently wrong result.
// Only while a context exists: the host-only entry points (challenger, ola_air_kernels_available) also work on machines
// without a HIP device, where the runtime answers every query with an error.
static std::atomic<int> g_live_contexts{0};
static void check_launch_errors() {
    if (g_live_contexts.load() <= 0) return;
    const hipError_t e = hipGetLastError();
    // hipErrorNotReady is what event / stream queries answer while work is in flight, not a failure
    if (e != hipSuccess && e != hipErrorNotReady) throw OlaError(OLA_E_HIP, std::string("kernel launch failed: ") + hipGetErrorString(e));
}
// The error slot is per host thread and shared with whatever else uses HIP in the process (the caller's framework polls
// events, for instance): forget what was there before this call so that only this call's launches are judged.
static void clear_stale_errors() {
    if (g_live_contexts.load() > 0) (void)hipGetLastError();
}
#define OLA_TRY try { clear_stale_errors();
#define OLA_CATCH                                                   \
        check_launch_errors();                                      \
    }                                                               \
    catch (const OlaError& e) { g_last_error = e.what(); return e.code; } \
    catch (const std::bad_alloc&) { g_last_error = "host out of memory"; return OLA_E_OOM; } \
    catch (const std::exception& e) { g_last_error = e.what(); return OLA_E_INTERNAL; } \
    return OLA_OK;

static void require(bool ok, const char* what) {
    if (!ok) throw OlaError(OLA_E_INVALID_ARG, std::string("invalid argument: ") + what);
}

// The HIP current device is a property of the calling host thread: every entry point that works on a context makes the
// context's device current for its duration (and puts the caller's back), so that hipMalloc and constant uploads land on the
// right GPU when the host drives several contexts or calls from a thread that never called hipSetDevice.
struct DeviceGuard {
    int prev = -1, want = -1;
    std::atomic<int>* calls = nullptr;   // the context's count of running C-ABI calls (DeviceCtx::calls: no block hand-over out of a busy context)
    explicit DeviceGuard(OlaCtx* ctx);
    // Restores whatever the caller had, also when the body moved on to other devices (the peer loops of ola_gpu_sync /
    // ola_gpu_trim on a multi-device context end on the last peer): compare with the device that is current NOW, not with `want`.
    ~DeviceGuard() {
        if (calls) calls->fetch_sub(1);
        if (prev < 0) return;
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess) { cur = -1; (void)hipGetLastError(); }
        if (cur != prev) (void)hipSetDevice(prev);
    }
};
#define OLA_ON_DEVICE(ctx) DeviceGuard ola_device_guard_(ctx)

DeviceGuard::DeviceGuard(OlaCtx* ctx) {
    if (!ctx) return;
    want = ctx->dev.device;
    if (hipGetDevice(&prev) != hipSuccess) { prev = -1; (void)hipGetLastError(); }
    if (prev != want) HIP_CHECK(hipSetDevice(want));
    calls = &ctx->dev.calls;          // last: a constructor that threw has no destructor to undo it
    calls->fetch_add(1);
}

// the configuration a context runs with: the caller's, or StarkConfig::standard_fast_config, validated
static OlaGpuConfig resolve_config(const OlaGpuConfig* cfg) {
    OlaGpuConfig d = {};
    d.device = -1; d.stream = nullptr; d.rate_bits = 3; d.cap_height = 4; d.proof_of_work_bits = 16;
    d.fri_arity_bits = 4; d.fri_final_poly_bits = 5; d.num_query_rounds = 28; d.num_challenges = 2;
    const OlaGpuConfig c = cfg ? *cfg : d;
    // the structure parameters every later call relies on (a zero arity would loop forever in fri_arities, zero proof-of-work
    // bits shift by 64 in the grinding kernel, ...)
    require(c.rate_bits >= 1 && c.rate_bits <= 8, "rate_bits must be in 1..8");
    require(c.cap_height <= 16, "cap_height must be at most 16");
    require(c.proof_of_work_bits >= 1 && c.proof_of_work_bits <= 40, "proof_of_work_bits must be in 1..40");
    require(c.fri_arity_bits >= 1 && c.fri_arity_bits <= 8, "fri_arity_bits must be in 1..8");
    require(c.fri_final_poly_bits <= 16, "fri_final_poly_bits must be at most 16");
    require(c.num_query_rounds >= 1 && c.num_query_rounds <= 1024, "num_query_rounds must be in 1..1024");
    require(c.num_challenges == 2, "num_challenges must be 2 (circuits/src/stark/config.rs)");
    require(c.hasher == OLA_HASH_POSEIDON || c.hasher == OLA_HASH_BLAKE3, "hasher must be OLA_HASH_POSEIDON or OLA_HASH_BLAKE3");
    return c;
}

// one device's context: stream, Poseidon constants, transform tables (device < 0: the calling thread's current device)
static std::unique_ptr<OlaCtx> create_device_ctx(const OlaGpuConfig& cfg, int device, void* stream) {
    std::unique_ptr<OlaCtx> c(new OlaCtx());
    c->cfg = cfg;
    c->cfg.device = device;
    c->cfg.stream = stream;
    c->dev.hasher = (int)cfg.hasher;
    { const char* t = getenv("OLA_TIMING"); c->dev.timing = t && *t && *t != '0'; c->dev.acct.on = c->dev.timing; }   // OLA_TIMING also prints the partition accounting
    // where ola_gpu_init's time goes (OLA_TIMING): the first HIP call of a process opens the runtime and the device, the first
    // touch of a symbol loads this library's code object for the device; everything of ours is lazy (tables, pinned ring, pool)
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!c->dev.timing) return;
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[ola-timing] init: %-52s %9.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    };
    if (device >= 0) HIP_CHECK(hipSetDevice(device));
    HIP_CHECK(hipGetDevice(&c->dev.device));
    lap("hipSetDevice / hipGetDevice");
    if (stream) { c->dev.stream = (hipStream_t)stream; c->dev.owns_stream = false; }
    else { HIP_CHECK(hipStreamCreateWithFlags(&c->dev.stream, hipStreamNonBlocking)); c->dev.owns_stream = true; }
    lap("stream (first real device work: opens the device)");
    poseidon_init(&c->dev);
    lap("Poseidon constants (loads the code object)");
    c->tables = ntt_tables_create(&c->dev);
    lap("transform table registry (tables are built on use)");
    return c;
}

// prove_with_traces on a context that spans several GPUs: one worker thread per rank (the caller's thread is rank 0), every rank
// runs the whole prover on the same traces with the coset partition switched on and the library's own xGMI all-gather
// (peer_group.h) as its collective.  All ranks finish with the same AllProof bytes; rank 0's are returned.
static void prove_with_traces_multi(OlaCtx* ctx, const u64* airset, size_t airset_words, const TraceSource* traces, const uint32_t* log_n,
                                    const u64* params, const u64* compress, std::vector<uint8_t>& bytes) {
    const uint32_t world = (uint32_t)ctx->peers.size() + 1;
    PeerGroup& g = *ctx->group;
    g.reset();
    struct Result { std::vector<uint8_t> bytes; int code = 0; std::string msg; };
    std::vector<Result> res(world);
    const auto t0 = std::chrono::steady_clock::now();
    auto run = [&](uint32_t r) {
        OlaCtx* c = r == 0 ? ctx : ctx->peers[r - 1];
        try {
            HIP_CHECK(hipSetDevice(c->dev.device));
            (void)hipGetLastError();
            ShardInfo sh;
            sh.rank = r; sh.world = world;
            while ((1u << sh.log_world) < world) sh.log_world++;
            if (ctx->rccl) { sh.all_gather = rccl_all_gather; sh.user = &ctx->rccl->ranks[r]; }
            else { sh.all_gather = peer_all_gather; sh.user = &g.ranks[r]; }
            sh.stream_ordered = true;
            c->dev.shard = sh;
            c->dev.acct.on = ctx->dev.acct.on;
            c->dev.acct.begin_proof();
            c->dev.scopes.on = ctx->dev.scopes.on && r == 0;
            c->dev.scopes_begin();
            prove_with_traces(&c->dev, *c->tables, c->cfg, airset, airset_words, traces, log_n, params, compress, res[r].bytes);
            HIP_CHECK(hipStreamSynchronize(c->dev.stream));
            c->dev.scopes_collect();
            c->dev.acct.collect(&c->dev.scopes);
            const hipError_t e = hipGetLastError();
            if (e != hipSuccess && e != hipErrorNotReady) throw OlaError(OLA_E_HIP, std::string("kernel launch failed: ") + hipGetErrorString(e));
        } catch (const OlaError& e) { res[r].code = e.code; res[r].msg = e.what(); g.fail(); }
        catch (const std::bad_alloc&) { res[r].code = OLA_E_OOM; res[r].msg = "host out of memory"; g.fail(); }
        catch (const std::exception& e) { res[r].code = OLA_E_INTERNAL; res[r].msg = e.what(); g.fail(); }
        c->dev.shard = ShardInfo();
    };
    std::vector<std::thread> workers;
    try {
        for (uint32_t r = 1; r < world; r++) workers.emplace_back(run, r);
    } catch (const std::exception& e) {          // the process cannot start another thread: release the ranks that did start
        g.fail();
        for (std::thread& t : workers) t.join();
        throw OlaError(OLA_E_INTERNAL, std::string("could not start the rank threads of the multi-device context: ") + e.what());
    }
    run(0);
    for (std::thread& t : workers) t.join();
    ctx->dev.acct.wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    // the rank that failed first-hand explains more than the ones that were released from a barrier because of it
    int bad = -1;
    for (uint32_t r = 0; r < world; r++)
        if (res[r].code != 0 && (bad < 0 || res[bad].msg.find("a peer rank") != std::string::npos)) bad = (int)r;
    if (bad >= 0) throw OlaError(res[bad].code, "rank " + std::to_string(bad) + " (device " + std::to_string(g.ranks[bad].device) + "): " + res[bad].msg);
    for (uint32_t r = 1; r < world; r++)
        if (res[r].bytes != res[0].bytes) throw OlaError(OLA_E_INTERNAL, "ranks of the multi-device context produced different proofs (rank " + std::to_string(r) + ")");
    bytes = std::move(res[0].bytes);
}

// ola_gpu_warmup: the process-wide start-up work done AHEAD of the first context, on a helper thread, where the reference brings
// its own GPU state up -- OlaStark::default() calls plonky2::field::cfft::ntt::init_gpu() (circuits/src/stark/ola_stark.rs:47,
// plonky2/field/src/cfft/ntt/mod.rs:53-99) before prove() generates the traces (client/src/main.rs:191-200).  What costs time in
// ola_gpu_init is not ours to shrink (HIP runtime start-up, opening the device, loading the code objects: OLA_TIMING prints the
// split), but none of it depends on the configuration, so it can run while the host generates traces.  The state is never
// destroyed (the thread may still be running at exit).
struct WarmState {
    std::mutex mu;
    std::condition_variable cv;
    bool started = false, done = false;
    int device = -1;
    double ms = 0;
    std::string error;
    void* ring = nullptr;          // a pinned staging ring for the first context created on `device` (upload.h adopts it)
    size_t ring_bytes = 0;
    OlaCtx* ready = nullptr;       // the context the priming proofs ran on (tables built, pool filled): the first ola_gpu_init on `device` takes it
    static WarmState& get() { static WarmState* w = new WarmState(); return *w; }
};
// The priming instance: every table all zeros, heights that send the large tables through the three- and two-pass transforms, the
// small ones through the single-pass kernels and one through the interpreter quotient kernel -- the size classes a real proof
// uses.  The traces satisfy no AIR; DeviceCtx::priming switches the divisibility check off and the bytes are thrown away.
static void prime_context(OlaCtx* c, const std::vector<u64>& airset) {
    const std::vector<size_t> widths = airset_widths(airset.data(), airset.size());
    const size_t nt = widths.size();
    static const uint32_t pattern[12] = {12, 12, 18, 3, 16, 10, 10, 10, 10, 10, 10, 10};
    std::vector<uint32_t> log_n(nt);
    std::vector<std::vector<u64>> zeros(nt);
    std::vector<TraceSource> src(nt);
    for (size_t t = 0; t < nt; t++) {
        log_n[t] = pattern[t % 12];
        zeros[t].assign(widths[t] << log_n[t], 0);          // written, not calloc'ed: the upload path reads real pages
        src[t].base = zeros[t].data();
    }
    c->dev.priming = true;
    const bool acct = c->dev.acct.on, timing = c->dev.timing;
    c->dev.acct.on = false; c->dev.timing = false;
    for (uint32_t hasher : {OLA_HASH_POSEIDON, OLA_HASH_BLAKE3}) {
        c->cfg.hasher = hasher; c->dev.hasher = (int)hasher;
        std::vector<uint8_t> bytes;
        try {
            c->dev.acct.begin_proof();
            c->dev.scopes_begin();
            prove_with_traces(&c->dev, *c->tables, c->cfg, airset.data(), airset.size(), src.data(), log_n.data(), nullptr, nullptr, bytes);
            HIP_CHECK(hipStreamSynchronize(c->dev.stream));
            c->dev.acct.collect(nullptr);
        } catch (const std::exception&) { (void)hipGetLastError(); (void)hipStreamSynchronize(c->dev.stream); }
    }
    c->dev.priming = false;
    c->dev.acct.on = acct; c->dev.timing = timing;
    c->cfg.hasher = OLA_HASH_POSEIDON; c->dev.hasher = 0;
}

static void warmup_body(WarmState& w, int device, uint32_t flags, std::vector<u64> airset) {
    const auto t0 = std::chrono::steady_clock::now();
    const bool timing = [] { const char* t = getenv("OLA_TIMING"); return t && *t && *t != '0'; }();
    auto t_lap = t0;
    auto lap = [&](const char* what) {
        if (!timing) return;
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[ola-timing] warm-up thread: %-44s %9.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t_lap).count());
        t_lap = t1;
    };
    std::string err;
    void* ring = nullptr;
    size_t ring_bytes = 0;
    OlaCtx* ready = nullptr;
    int dev = device;
    try {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) throw OlaError(OLA_E_NO_DEVICE, "no HIP device visible");
        lap("HIP runtime start-up (hipGetDeviceCount)");
        if (device >= ndev) throw OlaError(OLA_E_INVALID_ARG, "device index out of range");
        if (device >= 0) HIP_CHECK(hipSetDevice(device));
        HIP_CHECK(hipGetDevice(&dev));
        HIP_CHECK(hipFree(nullptr));                          // opens the device
        lap("device open (hipFree(0))");
        poseidon_upload_constants();                          // first symbol of the main code object: loads it
        lap("main code object (constants upload)");
        for (const AirKernelEntry* e : AIR_KERNELS) {         // the generated quotient kernels are one code object each
            hipFuncAttributes fa;
            if (hipFuncGetAttributes(&fa, (const void*)e->kernel) != hipSuccess) (void)hipGetLastError();
        }
        lap("quotient-kernel code objects");
        if (flags & OLA_WARMUP_PINNED_RING) {
            const size_t want = (size_t)128 << 20;            // upload.h's default ring: 8 slots of 16 MB
            if (hipHostMalloc(&ring, want, hipHostMallocDefault) == hipSuccess) ring_bytes = want;
            else { (void)hipGetLastError(); ring = nullptr; }
            lap("pinned staging ring (128 MB)");
        }
        if (!airset.empty()) {
            // a context with the default configuration, primed by one throw-away proof per hash configuration: every kernel of the
            // proof path has been launched once, the transform tables exist, the pool holds the small blocks.  ola_gpu_init takes it
            // over (cfg and hasher are plain fields); the ring goes with it
            std::unique_ptr<OlaCtx> c = create_device_ctx(resolve_config(nullptr), dev, nullptr);
            if (ring) { c->dev.staging = ring; c->dev.staging_bytes = ring_bytes; ring = nullptr; ring_bytes = 0; }
            prime_context(c.get(), airset);
            ready = c.release();
            lap("priming proofs (all-zero instance, both hash configurations)");
        }
        HIP_CHECK(hipDeviceSynchronize());
    } catch (const std::exception& e) { err = e.what(); (void)hipGetLastError(); }
    std::lock_guard<std::mutex> lk(w.mu);
    w.ready = ready;
    w.device = dev; w.error = err; w.ring = ring; w.ring_bytes = ring_bytes;
    w.ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    w.done = true;
    w.cv.notify_all();
}
// every context creation waits for a warm-up that is under way (the runtime would serialise the two anyway)
static void warmup_join() {
    WarmState& w = WarmState::get();
    std::unique_lock<std::mutex> lk(w.mu);
    if (w.started) w.cv.wait(lk, [&] { return w.done; });
}
// the primed context, if the warm-up made one for this device and the caller does not bring a stream of its own
static OlaCtx* take_warm_ctx(const OlaGpuConfig& c, int device) {
    if (c.stream != nullptr) return nullptr;
    WarmState& w = WarmState::get();
    std::lock_guard<std::mutex> lk(w.mu);
    if (!w.done || !w.ready) return nullptr;
    int cur = -1;
    if (device < 0 && hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if ((device >= 0 ? device : cur) != w.device) return nullptr;
    OlaCtx* r = w.ready;
    w.ready = nullptr;
    r->cfg = c;
    r->cfg.device = w.device;
    r->dev.hasher = (int)c.hasher;
    return r;
}
namespace ola {
// upload.h: the ring a warm-up pinned for this device, once
void* take_warm_ring(int device, size_t want, size_t* got) {
    WarmState& w = WarmState::get();
    std::lock_guard<std::mutex> lk(w.mu);
    if (!w.done || !w.ring || w.device != device || w.ring_bytes < want) return nullptr;
    void* r = w.ring;
    *got = w.ring_bytes;
    w.ring = nullptr; w.ring_bytes = 0;
    return r;
}
}  // namespace ola

extern "C" {

const char* ola_gpu_last_error(void) { return g_last_error.c_str(); }

int32_t ola_gpu_warmup(int32_t device, uint32_t flags, const uint64_t* airset, size_t airset_words) {
    try {
        WarmState& w = WarmState::get();
        std::lock_guard<std::mutex> lk(w.mu);
        if (w.started) return OLA_OK;
        std::vector<u64> set;
        if (airset && airset_words) set.assign((const u64*)airset, (const u64*)airset + airset_words);   // the caller's copy need not outlive the call
        w.started = true;
        std::thread([&w, device, flags, set] { warmup_body(w, device, flags, set); }).detach();
    } catch (const std::exception& e) { g_last_error = e.what(); return OLA_E_INTERNAL; }
    return OLA_OK;
}

int32_t ola_gpu_warmup_wait(double* ms_out) {
    WarmState& w = WarmState::get();
    std::unique_lock<std::mutex> lk(w.mu);
    if (!w.started) { g_last_error = "invalid argument: ola_gpu_warmup was not called"; return OLA_E_INVALID_ARG; }
    w.cv.wait(lk, [&] { return w.done; });
    if (ms_out) *ms_out = w.ms;
    if (!w.error.empty()) { g_last_error = "warm-up failed: " + w.error; return OLA_E_HIP; }
    return OLA_OK;
}

int32_t ola_gpu_init(const OlaGpuConfig* cfg, OlaCtx** out_ctx) {
    OLA_TRY
    require(out_ctx != nullptr, "out_ctx is NULL");
    warmup_join();
    int ndev = 0;
    const auto t_rt = std::chrono::steady_clock::now();
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        throw OlaError(OLA_E_NO_DEVICE, "no HIP device visible (the backend has no CPU fallback)");
    { const char* t = getenv("OLA_TIMING");
      if (t && *t && *t != '0') fprintf(stderr, "[ola-timing] init: %-52s %9.3f ms\n", "hipGetDeviceCount (HIP runtime start-up)",
                                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_rt).count()); }
    const OlaGpuConfig c = resolve_config(cfg);
    require(c.device < ndev, "device index out of range");
    OlaCtx* warm = take_warm_ctx(c, c.device);
    *out_ctx = warm ? warm : create_device_ctx(c, c.device, c.stream).release();
    (*out_ctx)->dev.join_pool_registry();
    g_live_contexts.fetch_add(1);
    OLA_CATCH
}

int32_t ola_gpu_init_multi(const OlaGpuConfig* cfg, const int32_t* devices, uint32_t n_devices, OlaCtx** out_ctx) {
    OLA_TRY
    require(out_ctx != nullptr, "out_ctx is NULL");
    require(n_devices == 1 || n_devices == 2 || n_devices == 4 || n_devices == 8, "a context spans 1, 2, 4 or 8 devices");
    warmup_join();
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        throw OlaError(OLA_E_NO_DEVICE, "no HIP device visible (the backend has no CPU fallback)");
    const OlaGpuConfig c = resolve_config(cfg);
    require(n_devices <= (1u << c.rate_bits) && n_devices <= (1u << c.cap_height), "more devices than LDE cosets or Merkle cap entries");
    std::vector<int> dv(n_devices);
    for (uint32_t r = 0; r < n_devices; r++) {
        dv[r] = devices ? devices[r] : (int)r;
        require(dv[r] >= 0 && dv[r] < ndev, "device index out of range");
    }
    require(n_devices == 1 || c.stream == nullptr, "a multi-device context creates its own streams (cfg.stream must be NULL)");
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) { prev = -1; (void)hipGetLastError(); }
    struct Restore { int d; ~Restore() { if (d >= 0) (void)hipSetDevice(d); } } restore{prev};
    OlaCtx* warm = take_warm_ctx(c, dv[0]);
    std::unique_ptr<OlaCtx> root = warm ? std::unique_ptr<OlaCtx>(warm) : create_device_ctx(c, dv[0], c.stream);
    if (warm) HIP_CHECK(hipSetDevice(dv[0]));
    if (n_devices > 1) {
        for (uint32_t r = 1; r < n_devices; r++) root->peers.push_back(create_device_ctx(c, dv[r], nullptr).release());
        // xGMI peer access between every pair of distinct devices (the all-gather pulls from the peers' memory)
        for (uint32_t a = 0; a < n_devices; a++)
            for (uint32_t b = 0; b < n_devices; b++) {
                if (dv[a] == dv[b]) continue;
                int can = 0;
                HIP_CHECK(hipDeviceCanAccessPeer(&can, dv[a], dv[b]));
                if (!can) throw OlaError(OLA_E_HIP, "device " + std::to_string(dv[a]) + " cannot access device " + std::to_string(dv[b]) + " (no peer access)");
                HIP_CHECK(hipSetDevice(dv[a]));
                const hipError_t e = hipDeviceEnablePeerAccess(dv[b], 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) HIP_CHECK(e);
                (void)hipGetLastError();
            }
        root->group.reset(new PeerGroup());
        PeerGroup& g = *root->group;
        g.world = n_devices;
        g.ranks.resize(n_devices);
        for (uint32_t r = 0; r < n_devices; r++) {
            OlaCtx* rc = r == 0 ? root.get() : root->peers[r - 1];
            PeerRank& pr = g.ranks[r];
            pr.group = &g; pr.rank = r; pr.device = rc->dev.device; pr.stream = rc->dev.stream;
            HIP_CHECK(hipSetDevice(pr.device));
            HIP_CHECK(hipEventCreateWithFlags(&pr.ready, hipEventDisableTiming));
            HIP_CHECK(hipEventCreateWithFlags(&pr.done, hipEventDisableTiming));
        }
    }
    // OLA_COLLECTIVE = peer (default) | rccl: the carrier of this context's exchanges.  RCCL is loaded on request only; when it
    // cannot carry the context (library absent, ranks aliased onto one GPU) the context keeps the peer carrier and records why.
    {
        const char* e = getenv("OLA_COLLECTIVE");
        const std::string want = e ? e : "peer";
        if (want == "rccl") {
            std::vector<hipStream_t> streams(n_devices);
            for (uint32_t r = 0; r < n_devices; r++) streams[r] = r == 0 ? root->dev.stream : root->peers[r - 1]->dev.stream;
            std::string why;
            root->rccl = rccl_group_create(dv, streams, root->group.get(), why);
            if (root->rccl) root->collective_note = "RCCL " + std::to_string(root->rccl->version) + ": ncclAllGather on " + std::to_string(n_devices) + " communicator(s) of one ncclCommInitAll";
            else root->collective_note = "OLA_COLLECTIVE=rccl refused, peer carrier kept: " + why;
        } else if (want != "peer") {
            throw OlaError(OLA_E_INVALID_ARG, "OLA_COLLECTIVE must be peer or rccl");
        }
    }
    *out_ctx = root.release();
    if (n_devices == 1) (*out_ctx)->dev.join_pool_registry();     // the ranks of a multi-device context keep their pools to themselves
    g_live_contexts.fetch_add(1);
    OLA_CATCH
}

int32_t ola_gpu_collective(OlaCtx* ctx, uint32_t* carrier, uint32_t* ranks, char* note, size_t note_cap) {
    OLA_TRY
    require(ctx != nullptr, "ctx");
    if (carrier) *carrier = ctx->rccl ? OLA_COLLECTIVE_RCCL : (ctx->group ? OLA_COLLECTIVE_PEER : OLA_COLLECTIVE_NONE);
    if (ranks) *ranks = ctx->rccl ? (uint32_t)ctx->rccl->comms.size() : (uint32_t)ctx->peers.size() + 1;
    if (note && note_cap) { snprintf(note, note_cap, "%s", ctx->collective_note.c_str()); }
    OLA_CATCH
}

// The context's all-gather by itself, on every rank of the context: each rank fills a block of `bytes_per_rank` bytes with its
// own pattern, `reps` gathers run back to back on the ranks' streams, every byte of every rank's result is compared.
int32_t ola_gpu_all_gather_check(OlaCtx* ctx, uint32_t carrier, size_t bytes_per_rank, uint32_t reps, double* ms_per_gather, uint64_t* mismatches) {
    OLA_TRY
    OLA_ON_DEVICE(ctx);
    require(ctx && mismatches, "null pointer");
    require(bytes_per_rank > 0 && bytes_per_rank % 8 == 0 && reps >= 1, "bytes_per_rank must be a positive multiple of 8, reps >= 1");
    require(carrier == OLA_COLLECTIVE_PEER || carrier == OLA_COLLECTIVE_RCCL, "carrier must be OLA_COLLECTIVE_PEER or OLA_COLLECTIVE_RCCL");
    if (carrier == OLA_COLLECTIVE_RCCL && !ctx->rccl)
        throw OlaError(OLA_E_INVALID_ARG, "this context has no RCCL carrier" + (ctx->collective_note.empty() ? std::string(" (create it with OLA_COLLECTIVE=rccl)") : ": " + ctx->collective_note));
    if (carrier == OLA_COLLECTIVE_PEER && !ctx->group) throw OlaError(OLA_E_INVALID_ARG, "a single-device context has no peers to gather from");
    const uint32_t world = carrier == OLA_COLLECTIVE_RCCL ? (uint32_t)ctx->rccl->comms.size() : (uint32_t)ctx->peers.size() + 1;
    if (ctx->group) ctx->group->reset();
    std::vector<uint64_t> bad(world, 0);
    std::vector<double> ms(world, 0.0);
    std::vector<int> code(world, 0);
    std::vector<std::string> msg(world);
    auto run = [&](uint32_t r) {
        OlaCtx* c = r == 0 ? ctx : ctx->peers[r - 1];
        DeviceCtx* d = &c->dev;
        void *send = nullptr, *recv = nullptr;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        try {
            HIP_CHECK(hipSetDevice(d->device));
            send = d->alloc(bytes_per_rank);
            recv = d->alloc(bytes_per_rank * world);
            HIP_CHECK(hipMemsetAsync(send, (int)(0x11 * (r + 1)), bytes_per_rank, d->stream));
            HIP_CHECK(hipMemsetAsync(recv, 0, bytes_per_rank * world, d->stream));
            HIP_CHECK(hipEventCreate(&e0));
            HIP_CHECK(hipEventCreate(&e1));
            void* user = carrier == OLA_COLLECTIVE_RCCL ? (void*)&ctx->rccl->ranks[r] : (void*)&ctx->group->ranks[r];
            auto gather = carrier == OLA_COLLECTIVE_RCCL ? rccl_all_gather : peer_all_gather;
            if (gather(user, send, recv, bytes_per_rank) != 0) throw OlaError(OLA_E_INTERNAL, "all-gather failed (warm-up)");
            HIP_CHECK(hipEventRecord(e0, d->stream));
            for (uint32_t i = 0; i < reps; i++)
                if (gather(user, send, recv, bytes_per_rank) != 0) throw OlaError(OLA_E_INTERNAL, "all-gather failed");
            HIP_CHECK(hipEventRecord(e1, d->stream));
            HIP_CHECK(hipStreamSynchronize(d->stream));
            float t = 0;
            HIP_CHECK(hipEventElapsedTime(&t, e0, e1));
            ms[r] = t / reps;
            std::vector<unsigned char> host(bytes_per_rank * world);
            HIP_CHECK(hipMemcpy(host.data(), recv, host.size(), hipMemcpyDeviceToHost));
            for (uint32_t j = 0; j < world; j++)
                for (size_t k = 0; k < bytes_per_rank; k++) bad[r] += host[(size_t)j * bytes_per_rank + k] != (unsigned char)(0x11 * (j + 1));
        } catch (const OlaError& e) { code[r] = e.code; msg[r] = e.what(); if (ctx->group) ctx->group->fail(); }
        catch (const std::exception& e) { code[r] = OLA_E_INTERNAL; msg[r] = e.what(); if (ctx->group) ctx->group->fail(); }
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        (void)hipStreamSynchronize(d->stream);
        d->free(send);
        d->free(recv);
    };
    std::vector<std::thread> workers;
    for (uint32_t r = 1; r < world; r++) workers.emplace_back(run, r);
    run(0);
    for (std::thread& t : workers) t.join();
    for (uint32_t r = 0; r < world; r++)
        if (code[r] != 0) throw OlaError(code[r], "rank " + std::to_string(r) + ": " + msg[r]);
    uint64_t total = 0;
    double worst = 0;
    for (uint32_t r = 0; r < world; r++) { total += bad[r]; worst = std::max(worst, ms[r]); }
    *mismatches = total;
    if (ms_per_gather) *ms_per_gather = worst;
    OLA_CATCH
}

int32_t ola_gpu_abi_version(size_t* challenger_size, size_t* config_size) {
    if (challenger_size) *challenger_size = sizeof(OlaChallenger);
    if (config_size) *config_size = sizeof(OlaGpuConfig);
    return OLA_GPU_ABI_VERSION;
}

int32_t ola_gpu_device_count(OlaCtx* ctx, uint32_t* n_devices) {
    OLA_TRY
    require(ctx && n_devices, "null pointer");
    *n_devices = (uint32_t)ctx->peers.size() + 1;
    OLA_CATCH
}

int32_t ola_gpu_free(OlaCtx* ctx) {
    OLA_TRY
    if (ctx) {
        int prev = -1;
        if (hipGetDevice(&prev) != hipSuccess) { prev = -1; (void)hipGetLastError(); }
        (void)hipSetDevice(ctx->dev.device);
        (void)hipStreamSynchronize(ctx->dev.stream);
        delete ctx;
        if (prev >= 0) (void)hipSetDevice(prev);
        g_live_contexts.fetch_sub(1);
    }
    OLA_CATCH
}

int32_t ola_gpu_sync(OlaCtx* ctx) {
    OLA_TRY
    OLA_ON_DEVICE(ctx);
    require(ctx, "ctx");
    HIP_CHECK(hipStreamSynchronize(ctx->dev.stream));
    for (OlaCtx* p : ctx->peers) { HIP_CHECK(hipSetDevice(p->dev.device)); HIP_CHECK(hipStreamSynchronize(p->dev.stream)); }
    OLA_CATCH
}

// ------------------------------------------------------------------------------------------------ NTT
static void ntt_dev(OlaCtx* ctx, int32_t op, const u64* in, u64* out, u64* scratch, uint32_t log_n, uint32_t batch,
                    u64 shift, uint32_t blowup_log) {
    require(log_n <= 32 - (op == OLA_NTT_COSET_LDE || op == OLA_NTT_COSET_LDE_LEAF_ORDER ? blowup_log : 0), "log_n too large");
    if (batch == 0) return;
    const size_t n = (size_t)1 << log_n;
    NttTables& t = *ctx->tables;
    DevBuf own(&ctx->dev);   // a scratch buffer of our own is released on every path
    auto need_scratch = [&](size_t elems) {
        if (!scratch) scratch = own.alloc(elems);
    };
    switch (op) {
        case OLA_NTT_EVALUATE:
            if (log_n > 13) need_scratch(n * batch);
            ntt_evaluate(t, in, out, scratch, log_n, batch);
            break;
        case OLA_NTT_INTERPOLATE:
            if (log_n > 13) need_scratch(n * batch);
            ntt_interpolate(t, in, out, scratch, log_n, batch);
            break;
        case OLA_NTT_COSET_LDE_LEAF_ORDER:
            require(gl_canon(shift) == GL_GENERATOR, "leaf-order LDE is defined for the coset shift 7");
            ntt_lde_leaf_order(t, in, out, log_n, blowup_log, batch);
            break;
        case OLA_NTT_COSET_LDE: {
            // natural order: P(shift * g^m).  Computed as blowup cosets in leaf order, then un-bit-reversed rows.
            require(gl_canon(shift) == GL_GENERATOR || blowup_log == 0, "LDE with blowup > 1 is defined for the coset shift 7");
            if (blowup_log == 0) {
                if (log_n > 13) need_scratch(n * batch);
                ntt_coset_evaluate(t, in, out, scratch, log_n, batch, gl_canon(shift), true);
            } else {
                const size_t N = n << blowup_log;
                need_scratch(N * batch);
                ntt_lde_leaf_order(t, in, scratch, log_n, blowup_log, batch);
                launch_bitrev_rows(&ctx->dev, scratch, out, log_n + blowup_log, batch);
            }
            break;
        }
        case OLA_NTT_COSET_INTERPOLATE: {
            if (log_n > 13) need_scratch(n * batch);
            ntt_coset_interpolate(t, in, out, scratch, log_n, batch, gl_canon(shift));
            break;
        }
        default: require(false, "unknown NTT op");
    }
}

int32_t ola_ntt_batch_dev(OlaCtx* ctx, int32_t op, const uint64_t* in_dev, uint64_t* out_dev, uint64_t* scratch_dev,
                          uint32_t log_n, uint32_t batch, uint64_t shift, uint32_t blowup_log) {
    OLA_TRY
    OLA_ON_DEVICE(ctx);
    require(ctx && in_dev && out_dev, "null pointer");
    ntt_dev(ctx, op, (const u64*)in_dev, (u64*)out_dev, (u64*)scratch_dev, log_n, batch, shift, blowup_log);
    HIP_CHECK(hipStreamSynchronize(ctx->dev.stream));
    OLA_CATCH
}

int32_t ola_ntt_batch(OlaCtx* ctx, int32_t op, const uint64_t* in, uint64_t* out, uint32_t log_n, uint32_t batch,
                      uint64_t shift, uint32_t blowup_log) {
    OLA_TRY
    OLA_ON_DEVICE(ctx);
    require(ctx && in && out, "null pointer");
    require(log_n <= 32 && blowup_log <= 8, "log_n too large");          // before any size arithmetic
    const size_t n = (size_t)1 << log_n;
    const bool grows = (op == OLA_NTT_COSET_LDE || op == OLA_NTT_COSET_LDE_LEAF_ORDER);
    const size_t in_elems = n * batch, out_elems = grows ? (n << blowup_log) * batch : n * batch;
    if (batch == 0) return OLA_OK;
    DevBuf mem(&ctx->dev);   // released on every path, also when a later allocation throws
    u64* d_in = (u64*)mem.alloc_bytes(in_elems * 8);
    u64* d_out = (u64*)mem.alloc_bytes(out_elems * 8);
    HIP_CHECK(hipMemcpyAsync(d_in, in, in_elems * 8, hipMemcpyHostToDevice, ctx->dev.stream));
    ntt_dev(ctx, op, d_in, d_out, nullptr, log_n, batch, shift, blowup_log);
    HIP_CHECK(hipMemcpyAsync(out, d_out, out_elems * 8, hipMemcpyDeviceToHost, ctx->dev.stream));
    HIP_CHECK(hipStreamSynchronize(ctx->dev.stream));
    OLA_CATCH
}

// ------------------------------------------------------------------------------------------------ hashing
int32_t ola_poseidon_permute(OlaCtx* ctx, uint64_t* states, size_t n) {
    OLA_TRY
    OLA_ON_DEVICE(ctx);
    require(ctx && (states || n == 0), "null pointer");
    if (n == 0) return OLA_OK;
    DevBuf mem(&ctx->dev);   // released on every path, also when a later allocation throws
    u64* d = (u64*)mem.alloc_bytes(n * 96);
    HIP_CHECK(hipMemcpyAsync(d, states, n * 96, hipMemcpyHostToDevice, ctx->dev.stream));
    launch_poseidon_states(&ctx->dev, d, n);
    HIP_CHECK(hipMemcpyAsync(states, d, n * 96, hipMemcpyDeviceToHost, ctx->dev.stream));
    HIP_CHECK(hipStreamSynchronize(ctx->dev.stream));
    OLA_CATCH
}

int32_t ola_hash_rows(OlaCtx* ctx, const uint64_t* rows, size_t num_rows, size_t row_len, uint64_t* digests) {
    OLA_TRY
    OLA_ON_DEVICE(ctx);
    require(ctx && digests && (rows || num_rows * row_len == 0), "null pointer");
    if (num_rows == 0) return OLA_OK;
    DevBuf mem(&ctx->dev);   // released on every path, also when a later allocation throws
    u64* d_rows = (u64*)mem.alloc_bytes(num_rows * row_len * 8);
    u64* d_dig = (u64*)mem.alloc_bytes(num_rows * 32);
    if (row_len) HIP_CHECK(hipMemcpyAsync(d_rows, rows, num_rows * row_len * 8, hipMemcpyHostToDevice, ctx->dev.stream));
    launch_leaf_hash_rowmajor(&ctx->dev, d_rows, row_len, num_rows, d_dig);
    HIP_CHECK(hipMemcpyAsync(digests, d_dig, num_rows * 32, hipMemcpyDeviceToHost, ctx->dev.stream));
    HIP_CHECK(hipStreamSynchronize(ctx->dev.stream));
    OLA_CATCH
}

int32_t ola_merkle_cap(OlaCtx* ctx, const uint64_t* leaves, size_t num_leaves, size_t leaf_len, uint32_t cap_height,
                       uint64_t* cap_out) {
    OLA_TRY
    OLA_ON_DEVICE(ctx);
    require(ctx && leaves && cap_out, "null pointer");
    require(num_leaves && (num_leaves & (num_leaves - 1)) == 0, "num_leaves must be a power of two");
    require(((size_t)1 << cap_height) <= num_leaves, "cap height should be at most log2(leaves.len())");
    DevBuf mem(&ctx->dev);   // released on every path, also when a later allocation throws
    u64* d_rows = (u64*)mem.alloc_bytes(num_leaves * leaf_len * 8);
    u64* heap = (u64*)mem.alloc_bytes(2 * num_leaves * 32);
    HIP_CHECK(hipMemcpyAsync(d_rows, leaves, num_leaves * leaf_len * 8, hipMemcpyHostToDevice, ctx->dev.stream));
    launch_leaf_hash_rowmajor(&ctx->dev, d_rows, leaf_len, num_leaves, heap + 4 * num_leaves);
    launch_merkle_build(&ctx->dev, heap, num_leaves, cap_height);
    const size_t len_cap = (size_t)1 << cap_height;
    HIP_CHECK(hipMemcpyAsync(cap_out, heap + 4 * len_cap, len_cap * 32, hipMemcpyDeviceToHost, ctx->dev.stream));
    HIP_CHECK(hipStreamSynchronize(ctx->dev.stream));
    OLA_CATCH
}

int32_t ola_pow(OlaCtx* ctx, const uint64_t h[4], uint32_t bits, uint64_t* witness) {
    OLA_TRY
    OLA_ON_DEVICE(ctx);
    require(ctx && h && witness && bits >= 1 && bits <= 40, "bad argument");
    u64 hh[4] = {gl_canon(h[0]), gl_canon(h[1]), gl_canon(h[2]), gl_canon(h[3])};
    *witness = run_pow(&ctx->dev, hh, bits);
    OLA_CATCH
}

// ------------------------------------------------------------------------------------------------ commitment
static int32_t commit_common(OlaCtx* ctx, const uint64_t* const* cols_host, const uint64_t* cols_dev, uint32_t ncols,
                             uint32_t log_n, bool from_values, OlaBatch** out_batch, uint64_t* cap_out, uint32_t shard_rank = 0,
                             uint32_t shard_world = 1) {
    OLA_TRY
    require(shard_world >= 1 && (shard_world & (shard_world - 1)) == 0 && shard_rank < shard_world, "shard rank / world");
    uint32_t log_world = 0;
    while ((1u << log_world) < shard_world) log_world++;
    require(ctx && out_batch && cap_out && (cols_host || cols_dev), "null pointer");
    require(ncols >= 1, "ncols");
    require(log_n + ctx->cfg.rate_bits <= 32, "log_n too large");
    require(log_n + ctx->cfg.rate_bits >= ctx->cfg.cap_height, "cap height should be at most log2(leaves.len())");
    std::unique_ptr<OlaBatch> b(batch_commit(&ctx->dev, *ctx->tables, cols_host, (const u64*)cols_dev, ncols, log_n,
                                             ctx->cfg.rate_bits, ctx->cfg.cap_height, from_values, shard_rank, log_world));
    batch_read_cap(&ctx->dev, *b, (u64*)cap_out);
    *out_batch = b.release();
    OLA_CATCH
}
int32_t ola_commit_values(OlaCtx* ctx, const uint64_t* const* cols, uint32_t ncols, uint32_t log_n, OlaBatch** ob, uint64_t* cap) {
    return commit_common(ctx, cols, nullptr, ncols, log_n, true, ob, cap);
}
int32_t ola_commit_coeffs(OlaCtx* ctx, const uint64_t* const* cols, uint32_t ncols, uint32_t log_n, OlaBatch** ob, uint64_t* cap) {
    return commit_common(ctx, cols, nullptr, ncols, log_n, false, ob, cap);
}
int32_t ola_commit_values_dev(OlaCtx* ctx, const uint64_t* cols_dev, uint32_t ncols, uint32_t log_n, OlaBatch** ob, uint64_t* cap) {
    return commit_common(ctx, nullptr, cols_dev, ncols, log_n, true, ob, cap);
}
int32_t ola_commit_coeffs_dev(OlaCtx* ctx, const uint64_t* cols_dev, uint32_t ncols, uint32_t log_n, OlaBatch** ob, uint64_t* cap) {
    return commit_common(ctx, nullptr, cols_dev, ncols, log_n, false, ob, cap);
}
int32_t ola_commit_values_shard(OlaCtx* ctx, const uint64_t* const* cols, uint32_t ncols, uint32_t log_n, uint32_t rank,
                                uint32_t world, OlaBatch** ob, uint64_t* cap_slice) {
    return commit_common(ctx, cols, nullptr, ncols, log_n, true, ob, cap_slice, rank, world);
}
int32_t ola_commit_values_shard_dev(OlaCtx* ctx, const uint64_t* cols_dev, uint32_t ncols, uint32_t log_n, uint32_t rank,
                                    uint32_t world, OlaBatch** ob, uint64_t* cap_slice) {
    return commit_common(ctx, nullptr, cols_dev, ncols, log_n, true, ob, cap_slice, rank, world);
}
int32_t ola_batch_free(OlaCtx* ctx, OlaBatch* batch) {
    OLA_TRY
    OLA_ON_DEVICE(ctx);
    require(ctx, "ctx");
    if (batch) { HIP_CHECK(hipStreamSynchronize(ctx->dev.stream)); batch_destroy(&ctx->dev, batch); }
    OLA_CATCH
}
int32_t ola_batch_shape(const OlaBatch* b, uint32_t* ncols, uint32_t* log_n, uint32_t* rate_bits) {
    OLA_TRY
    require(b, "batch");
    if (ncols) *ncols = b->ncols;
    if (log_n) *log_n = b->log_n;
    if (rate_bits) *rate_bits = b->rate_bits;
    OLA_CATCH
}
int32_t ola_batch_get_coeffs(OlaCtx* ctx, const OlaBatch* b, uint64_t* out) {
    OLA_TRY
    OLA_ON_DEVICE(ctx);
    require(ctx && b && out, "null pointer");
    HIP_CHECK(hipMemcpyAsync(out, b->coeffs, ((size_t)b->ncols << b->log_n) * 8, hipMemcpyDeviceToHost, ctx->dev.stream));
    HIP_CHECK(hipStreamSynchronize(ctx->dev.stream));
    OLA_CATCH
}
int32_t ola_batch_get_leaf(OlaCtx* ctx, const OlaBatch* b, size_t leaf_index, uint64_t* row_out, uint64_t* siblings_out) {
    OLA_TRY
    OLA_ON_DEVICE(ctx);
    require(ctx && b && row_out, "null pointer");
    require(leaf_index < b->num_leaves(), "leaf index out of range");
    batch_get_leaf(&ctx->dev, *b, leaf_index, (u64*)row_out, (u64*)siblings_out, &*ctx->tables);
    OLA_CATCH
}
static void require_full(const OlaBatch* b) {
    if (b && b->is_shard()) throw OlaError(OLA_E_INVALID_ARG, "this entry point needs a complete commitment, not one GPU's coset share");
}

int32_t ola_batch_get_lde_row(OlaCtx* ctx, const OlaBatch* b, size_t index, size_t step, uint64_t* row_out) {
    OLA_TRY
    OLA_ON_DEVICE(ctx);
    require(ctx && b && row_out, "null pointer");
    require_full(b);   // natural-order LDE rows interleave the cosets of all shards
    require(step == 0 || index <= b->num_leaves() / step, "row index out of range");
    const size_t nat = index * step;
    require(nat < b->num_leaves(), "row index out of range");
    batch_get_leaf(&ctx->dev, *b, bitrev32((u32)nat, b->log_n + b->rate_bits), (u64*)row_out, nullptr, &*ctx->tables);
    OLA_CATCH
}

// ------------------------------------------------------------------------------------------------ transcript
int32_t ola_challenger_init(OlaChallenger* ch) {
    OLA_TRY
    require(ch, "challenger");
    challenger_init(*ch, OLA_HASH_POSEIDON);
    OLA_CATCH
}
int32_t ola_challenger_init_hasher(OlaChallenger* ch, uint32_t hasher) {
    OLA_TRY
    require(ch, "challenger");
    require(hasher == OLA_HASH_POSEIDON || hasher == OLA_HASH_BLAKE3, "hasher must be OLA_HASH_POSEIDON or OLA_HASH_BLAKE3");
    challenger_init(*ch, hasher);
    OLA_CATCH
}
int32_t ola_challenger_observe_cap(OlaChallenger* ch, const uint64_t* digests, size_t n) {
    OLA_TRY
    require(ch && (digests || n == 0), "null pointer");
    challenger_observe_cap(*ch, (const u64*)digests, n);
    OLA_CATCH
}
int32_t ola_blake3_hash_elements(const uint64_t* elems, size_t n, uint64_t out[4]) {
    OLA_TRY
    require(elems && out && n >= 1 && n <= 4096, "1..4096 elements");
    std::vector<u64> w(n);
    for (size_t i = 0; i < n; i++) w[i] = gl_canon(elems[i]);
    b3_hash_bytes32_host(w.data(), (u32)n, (u64*)out);
    OLA_CATCH
}
int32_t ola_challenger_observe(OlaChallenger* ch, const uint64_t* e, size_t n) {
    OLA_TRY
    require(ch && (e || n == 0), "null pointer");
    challenger_observe(*ch, (const u64*)e, n);
    OLA_CATCH
}
int32_t ola_challenger_get(OlaChallenger* ch, uint64_t* out, size_t n) {
    OLA_TRY
    require(ch && (out || n == 0), "null pointer");
    for (size_t i = 0; i < n; i++) out[i] = challenger_get(*ch);
    OLA_CATCH
}
int32_t ola_challenger_compact(OlaChallenger* ch) {
    OLA_TRY
    require(ch, "challenger");
    challenger_compact(*ch);
    OLA_CATCH
}

int32_t ola_open_and_prove(OlaCtx* ctx, const OlaBatch* trace, const OlaBatch* zs, const OlaBatch* quotient,
                           uint32_t num_permutation_zs, OlaChallenger* challenger, uint8_t* out, size_t cap,
                           size_t* out_len, size_t* openings_len) {
    OLA_TRY
    OLA_ON_DEVICE(ctx);
    require(ctx && trace && zs && quotient && challenger && out_len, "null pointer");
    require_full(trace); require_full(zs); require_full(quotient);
    require(trace->log_n == zs->log_n && trace->log_n == quotient->log_n, "degree mismatch between commitments");
    require(num_permutation_zs <= zs->ncols, "num_permutation_zs");
    require(challenger->hasher == ctx->cfg.hasher, "the challenger was not initialised for this context's hasher (ola_challenger_init_hasher)");
    OlaChallenger ch = *challenger;  // only committed on success
    std::vector<uint8_t> bytes;
    size_t olen = 0;
    open_and_prove(&ctx->dev, *ctx->tables, ctx->cfg, *trace, *zs, *quotient, num_permutation_zs, ch, bytes, olen);
    *out_len = bytes.size();
    if (openings_len) *openings_len = olen;
    if (bytes.size() > cap || !out) throw OlaError(OLA_E_INVALID_ARG, "output buffer too small");
    memcpy(out, bytes.data(), bytes.size());
    *challenger = ch;
    OLA_CATCH
}

// ---- the same, one step per call (fri.hip "one step per call")
int32_t ola_open(OlaCtx* ctx, const OlaBatch* trace, const OlaBatch* zs, const OlaBatch* quotient, uint32_t num_permutation_zs, const uint64_t zeta[2],
                 uint8_t* out, size_t cap, size_t* out_len, OlaFri** fri_out) {
    OLA_TRY
    OLA_ON_DEVICE(ctx);
    require(ctx && trace && zs && quotient && zeta && out_len && fri_out, "null pointer");
    require(ctx->peers.empty() && ctx->dev.shard.world <= 1, "the step-wise opening entry points run on single-device contexts");
    require_full(trace); require_full(zs); require_full(quotient);
    require(trace->log_n == zs->log_n && trace->log_n == quotient->log_n, "degree mismatch between commitments");
    require(num_permutation_zs <= zs->ncols, "num_permutation_zs");
    std::unique_ptr<OlaFri> f(new OlaFri(&ctx->dev, &*ctx->tables, ctx->cfg, trace, zs, quotient, num_permutation_zs));
    f->owner = ctx;
    std::vector<uint8_t> bytes;
    fri_steps_open(*f, (const u64*)zeta, bytes);
    *out_len = bytes.size();
    if (bytes.size() > cap || !out) throw OlaError(OLA_E_INVALID_ARG, "output buffer too small");
    memcpy(out, bytes.data(), bytes.size());
    *fri_out = f.release();
    OLA_CATCH
}
int32_t ola_fri_plan(const OlaFri* fri, uint32_t* arity_bits, uint32_t cap, uint32_t* n_layers, uint32_t* final_poly_len) {
    OLA_TRY
    require(fri && n_layers, "null pointer");
    *n_layers = (uint32_t)fri->arities.size();
    if (final_poly_len) {
        int d = fri->degree_bits;
        for (int a : fri->arities) d -= a;
        *final_poly_len = 1u << d;
    }
    require(!arity_bits || cap >= fri->arities.size(), "arity_bits too small");
    if (arity_bits) for (size_t i = 0; i < fri->arities.size(); i++) arity_bits[i] = (uint32_t)fri->arities[i];
    OLA_CATCH
}
int32_t ola_fri_commit_begin(OlaFri* fri, const uint64_t alpha[2]) {
    OLA_TRY
    require(fri && alpha, "null pointer");
    OLA_ON_DEVICE((OlaCtx*)fri->owner);
    fri_steps_begin(*fri, (const u64*)alpha);
    OLA_CATCH
}
int32_t ola_fri_commit_next_layer(OlaFri* fri, const uint64_t* beta, uint64_t* cap_out) {
    OLA_TRY
    require(fri && cap_out, "null pointer");
    OLA_ON_DEVICE((OlaCtx*)fri->owner);
    fri_steps_next_layer(*fri, (const u64*)beta, (u64*)cap_out);
    OLA_CATCH
}
int32_t ola_fri_commit_finish(OlaFri* fri, const uint64_t* beta, uint64_t* final_poly_out, size_t cap_elems, size_t* n_out) {
    OLA_TRY
    require(fri && final_poly_out, "null pointer");
    OLA_ON_DEVICE((OlaCtx*)fri->owner);
    const size_t got = fri_steps_finish(*fri, (const u64*)beta, (u64*)final_poly_out, cap_elems);
    if (n_out) *n_out = got;
    OLA_CATCH
}
int32_t ola_fri_query(OlaFri* fri, const uint64_t* x_index, uint32_t n, uint8_t* out, size_t cap, size_t* out_len) {
    OLA_TRY
    require(fri && (x_index || n == 0) && out_len, "null pointer");
    OLA_ON_DEVICE((OlaCtx*)fri->owner);
    std::vector<uint8_t> bytes;
    fri_steps_query(*fri, (const u64*)x_index, n, bytes);
    *out_len = bytes.size();
    if (bytes.size() > cap || !out) throw OlaError(OLA_E_INVALID_ARG, "output buffer too small");
    memcpy(out, bytes.data(), bytes.size());
    OLA_CATCH
}
int32_t ola_fri_free(OlaFri* fri) {
    OLA_TRY
    if (fri) {
        OLA_ON_DEVICE((OlaCtx*)fri->owner);
        delete fri;
    }
    OLA_CATCH
}

// the body of both whole-proof entry points: `src[t]` says where table t's columns are
static void prove_all(OlaCtx* ctx, const uint64_t* airset, size_t airset_words, const std::vector<TraceSource>& src, const uint32_t* log_n,
                      const uint64_t* params, const uint64_t* compress_challenges, uint8_t* out, size_t cap, size_t* out_len) {
    std::vector<uint8_t> bytes;
    if (ctx->peers.empty()) {
        const auto t0 = std::chrono::steady_clock::now();
        // The first whole proof of a context finds an empty pool: a helper thread starts allocating every large block the proof
        // will ask for (reserve_for_proof, ola_gpu_reserve's list: the ones needed first come first) while this thread uploads
        // and commits -- hipMalloc of tens of GB, and the driver's scrubbing of previously used VRAM inside it, then overlap the
        // first commitments instead of stalling each later phase when it asks for a size the pool has not seen (the memory
        // table's quotient commitment: 104 ms cold against 41 ms warm).  OLA_AUTO_RESERVE=0 switches it off.
        if (!ctx->first_proof_done) {
            ctx->first_proof_done = true;
            static const bool auto_reserve = [] { const char* e = getenv("OLA_AUTO_RESERVE"); return !(e && *e == '0'); }();
            size_t cached;
            { std::lock_guard<std::mutex> lk(ctx->dev.mu); cached = ctx->dev.cached_bytes + ctx->dev.pending.size(); }
            if (auto_reserve && cached < ((size_t)1 << 30)) reserve_for_proof(&ctx->dev, ctx->cfg, (const u64*)airset, airset_words, log_n);
        }
        ctx->dev.acct.begin_proof();
        ctx->dev.scopes_begin();
        prove_with_traces(&ctx->dev, *ctx->tables, ctx->cfg, (const u64*)airset, airset_words, src.data(), log_n,
                          (const u64*)params, (const u64*)compress_challenges, bytes);
        if (ctx->dev.scopes.on) { HIP_CHECK(hipStreamSynchronize(ctx->dev.stream)); ctx->dev.scopes_collect(); }
        ctx->dev.acct.collect(&ctx->dev.scopes);
        ctx->dev.acct.wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (ctx->dev.timing && ctx->dev.acct.on) {
            const WorkAcct& a = ctx->dev.acct;
            const double sharded = a.sharded_ms[1] + a.sharded_ms[2] + a.sharded_ms[3];
            fprintf(stderr, "[ola-timing] coset partition: sharded_kernel_ms %.2f (divides by up to 2 / 4 / 8 ranks: %.2f / %.2f / %.2f), replicated_ms %.2f of %.2f, "
                            "exchange_bytes %llu in %u exchanges\n", sharded, a.sharded_ms[1], a.sharded_ms[2], a.sharded_ms[3], a.wall_ms - sharded, a.wall_ms,
                    (unsigned long long)a.exchange_bytes, a.exchanges);
        }
    } else {
        prove_with_traces_multi(ctx, (const u64*)airset, airset_words, src.data(), log_n, (const u64*)params,
                                (const u64*)compress_challenges, bytes);
    }
    *out_len = bytes.size();
    if (bytes.size() > cap || !out) {
        ctx->pending_proof = std::move(bytes);          // the work is not lost: ola_take_pending_proof hands it over
        throw OlaError(OLA_E_INVALID_ARG, "output buffer too small");
    }
    ctx->pending_proof.clear();
    memcpy(out, bytes.data(), bytes.size());
}

int32_t ola_prove_with_traces(OlaCtx* ctx, const uint64_t* airset, size_t airset_words, const uint64_t* const* traces,
                              const uint32_t* log_n, const uint64_t* params, const uint64_t* compress_challenges, uint8_t* out,
                              size_t cap, size_t* out_len) {
    OLA_TRY
    OLA_ON_DEVICE(ctx);
    require(ctx && airset && traces && log_n && out_len, "null pointer");
    const size_t nt = airset_widths((const u64*)airset, airset_words).size();
    std::vector<TraceSource> src(nt);
    for (size_t t = 0; t < nt; t++) { require(traces[t] != nullptr, "traces[t] is NULL"); src[t].base = (const u64*)traces[t]; }
    prove_all(ctx, airset, airset_words, src, log_n, params, compress_challenges, out, cap, out_len);
    OLA_CATCH
}

int32_t ola_prove_with_traces_cols(OlaCtx* ctx, const uint64_t* airset, size_t airset_words, const uint64_t* const* const* cols,
                                   const uint32_t* log_n, const uint64_t* params, const uint64_t* compress_challenges, uint8_t* out,
                                   size_t cap, size_t* out_len) {
    OLA_TRY
    OLA_ON_DEVICE(ctx);
    require(ctx && airset && cols && log_n && out_len, "null pointer");
    const std::vector<size_t> widths = airset_widths((const u64*)airset, airset_words);
    const size_t nt = widths.size();
    std::vector<TraceSource> src(nt);
    for (size_t t = 0; t < nt; t++) {
        require(cols[t] != nullptr, "cols[t] is NULL");
        for (size_t c = 0; c < widths[t]; c++) require(cols[t][c] != nullptr, "cols[t][c] is NULL");
        src[t].cols = (const u64* const*)cols[t];
    }
    prove_all(ctx, airset, airset_words, src, log_n, params, compress_challenges, out, cap, out_len);
    OLA_CATCH
}

int32_t ola_take_pending_proof(OlaCtx* ctx, uint8_t* out, size_t cap, size_t* out_len) {
    OLA_TRY
    OLA_ON_DEVICE(ctx);
    require(ctx && out_len, "null pointer");
    require(!ctx->pending_proof.empty(), "no proof is pending");
    *out_len = ctx->pending_proof.size();
    if (ctx->pending_proof.size() > cap || !out) throw OlaError(OLA_E_INVALID_ARG, "output buffer too small");
    memcpy(out, ctx->pending_proof.data(), ctx->pending_proof.size());
    ctx->pending_proof.clear();
    ctx->pending_proof.shrink_to_fit();
    OLA_CATCH
}

int32_t ola_prove_single_table(OlaCtx* ctx, const uint64_t* airset, size_t airset_words, uint32_t table,
                               const uint64_t* const* trace_cols, const OlaBatch* trace_commitment, const uint64_t* trace_cap,
                               const uint64_t* ctl_challenges, const uint64_t* params, OlaChallenger* challenger, uint8_t* out,
                               size_t cap, size_t* out_len) {
    OLA_TRY
    OLA_ON_DEVICE(ctx);
    require(ctx && airset && trace_cols && trace_commitment && trace_cap && ctl_challenges && challenger && out_len, "null pointer");
    require(challenger->hasher == ctx->cfg.hasher, "the challenger was not initialised for this context's hasher (ola_challenger_init_hasher)");
    std::vector<uint8_t> bytes;
    OlaChallenger ch = *challenger;          // the caller's transcript only advances when the proof was produced
    prove_single_table_host(&ctx->dev, *ctx->tables, ctx->cfg, (const u64*)airset, airset_words, table, (const u64* const*)trace_cols,
                            *trace_commitment, (const u64*)trace_cap, (const u64*)ctl_challenges, (const u64*)params, ch, bytes);
    *out_len = bytes.size();
    if (bytes.size() > cap || !out) throw OlaError(OLA_E_INVALID_ARG, "output buffer too small");
    memcpy(out, bytes.data(), bytes.size());
    *challenger = ch;
    OLA_CATCH
}

int32_t ola_table_shape(OlaCtx* ctx, const uint64_t* airset, size_t airset_words, uint32_t table, uint32_t out[6]) {
    OLA_TRY
    require(ctx && airset && out, "null pointer");
    table_shape_host(ctx->cfg, (const u64*)airset, airset_words, table, out);
    OLA_CATCH
}

static int32_t ola_zs_phase(OlaCtx* ctx, const uint64_t* airset, size_t airset_words, uint32_t table, uint32_t log_n,
                            const uint64_t* const* trace_cols, const uint64_t* perm_ch, const uint64_t* ctl_ch, int which, uint64_t* z_out) {
    OLA_TRY
    OLA_ON_DEVICE(ctx);
    require(ctx && airset && trace_cols && z_out, "null pointer");
    require(log_n + ctx->cfg.rate_bits <= 32, "log_n too large");
    std::vector<u64> cols;
    phase_zs_host(&ctx->dev, *ctx->tables, ctx->cfg, (const u64*)airset, airset_words, table, log_n, (const u64* const*)trace_cols,
                  (const u64*)perm_ch, (const u64*)ctl_ch, which, cols);
    if (!cols.empty()) memcpy(z_out, cols.data(), cols.size() * 8);
    OLA_CATCH
}
int32_t ola_perm_z(OlaCtx* ctx, const uint64_t* airset, size_t airset_words, uint32_t table, uint32_t log_n,
                   const uint64_t* const* trace_cols, const uint64_t* perm_challenges, uint64_t* z_out) {
    if (!perm_challenges) { g_last_error = "invalid argument: null pointer"; return OLA_E_INVALID_ARG; }
    return ola_zs_phase(ctx, airset, airset_words, table, log_n, trace_cols, perm_challenges, nullptr, 0, z_out);
}
int32_t ola_ctl_z(OlaCtx* ctx, const uint64_t* airset, size_t airset_words, uint32_t table, uint32_t log_n,
                  const uint64_t* const* trace_cols, const uint64_t* ctl_challenges, uint64_t* z_out) {
    if (!ctl_challenges) { g_last_error = "invalid argument: null pointer"; return OLA_E_INVALID_ARG; }
    return ola_zs_phase(ctx, airset, airset_words, table, log_n, trace_cols, nullptr, ctl_challenges, 1, z_out);
}

int32_t ola_quotient(OlaCtx* ctx, const uint64_t* airset, size_t airset_words, uint32_t table, const OlaBatch* trace, const OlaBatch* zs,
                     const uint64_t* perm_challenges, const uint64_t* ctl_challenges, const uint64_t* alphas, const uint64_t* params,
                     uint64_t* chunks_out) {
    OLA_TRY
    OLA_ON_DEVICE(ctx);
    require(ctx && airset && trace && zs && ctl_challenges && alphas && chunks_out, "null pointer");
    std::vector<u64> chunks;
    phase_quotient_host(&ctx->dev, *ctx->tables, ctx->cfg, (const u64*)airset, airset_words, table, *trace, *zs, (const u64*)perm_challenges,
                        (const u64*)ctl_challenges, (const u64*)alphas, (const u64*)params, chunks);
    memcpy(chunks_out, chunks.data(), chunks.size() * 8);
    OLA_CATCH
}

int32_t ola_generate_poseidon_trace(OlaCtx* ctx, const uint64_t* inputs, const uint64_t* filters, size_t n, uint64_t* out) {
    OLA_TRY
    OLA_ON_DEVICE(ctx);
    require(ctx && inputs && out, "null pointer");
    if (n == 0) return OLA_OK;
    DevBuf mem(&ctx->dev);   // released on every path, also when a later allocation throws
    u64* d_in = (u64*)mem.alloc_bytes(12 * n * 8);
    u64* d_f = filters ? (u64*)mem.alloc_bytes(4 * n * 8) : nullptr;
    u64* d_out = (u64*)mem.alloc_bytes(134 * n * 8);
    HIP_CHECK(hipMemcpyAsync(d_in, inputs, 12 * n * 8, hipMemcpyHostToDevice, ctx->dev.stream));
    if (filters) HIP_CHECK(hipMemcpyAsync(d_f, filters, 4 * n * 8, hipMemcpyHostToDevice, ctx->dev.stream));
    launch_poseidon_trace(&ctx->dev, d_in, d_f, n, d_out);
    HIP_CHECK(hipMemcpyAsync(out, d_out, 134 * n * 8, hipMemcpyDeviceToHost, ctx->dev.stream));
    HIP_CHECK(hipStreamSynchronize(ctx->dev.stream));
    OLA_CATCH
}

int32_t ola_permuted_cols_dev(OlaCtx* ctx, const uint64_t* inputs_dev, const uint64_t* table_dev, size_t n,
                          uint64_t* permuted_inputs_dev, uint64_t* permuted_table_dev) {
    OLA_TRY
    OLA_ON_DEVICE(ctx);
    require(ctx && ((inputs_dev && table_dev && permuted_inputs_dev && permuted_table_dev) || n == 0), "null pointer");
    permuted_cols_dev(&ctx->dev, (const u64*)inputs_dev, (const u64*)table_dev, n, (u64*)permuted_inputs_dev, (u64*)permuted_table_dev);
    HIP_CHECK(hipStreamSynchronize(ctx->dev.stream));
    OLA_CATCH
}

int32_t ola_permuted_cols(OlaCtx* ctx, const uint64_t* inputs, const uint64_t* table, size_t n, uint64_t* permuted_inputs,
                      uint64_t* permuted_table) {
    OLA_TRY
    OLA_ON_DEVICE(ctx);
    require(ctx && ((inputs && table && permuted_inputs && permuted_table) || n == 0), "null pointer");
    if (n == 0) return OLA_OK;
    DevBuf mem(&ctx->dev);
    u64* d = mem.alloc(4 * n);
    HIP_CHECK(hipMemcpyAsync(d, inputs, n * 8, hipMemcpyHostToDevice, ctx->dev.stream));
    HIP_CHECK(hipMemcpyAsync(d + n, table, n * 8, hipMemcpyHostToDevice, ctx->dev.stream));
    permuted_cols_dev(&ctx->dev, d, d + n, n, d + 2 * n, d + 3 * n);
    HIP_CHECK(hipMemcpyAsync(permuted_inputs, d + 2 * n, n * 8, hipMemcpyDeviceToHost, ctx->dev.stream));
    HIP_CHECK(hipMemcpyAsync(permuted_table, d + 3 * n, n * 8, hipMemcpyDeviceToHost, ctx->dev.stream));
    HIP_CHECK(hipStreamSynchronize(ctx->dev.stream));
    OLA_CATCH
}

int32_t ola_gpu_trim(OlaCtx* ctx) {
    OLA_TRY
    OLA_ON_DEVICE(ctx);
    require(ctx, "ctx");
    ctx->dev.release_cache();
    for (OlaCtx* p : ctx->peers) { HIP_CHECK(hipSetDevice(p->dev.device)); p->dev.release_cache(); }
    OLA_CATCH
}

int32_t ola_gpu_selftest(OlaCtx* ctx, uint64_t pairs, uint64_t* mismatches) {
    OLA_TRY
    OLA_ON_DEVICE(ctx);
    require(ctx && mismatches, "null pointer");
    *mismatches = field_selftest(&ctx->dev, pairs);
    OLA_CATCH
}

int32_t ola_gpu_memory_stats(OlaCtx* ctx, uint64_t out[4], int32_t reset) {
    OLA_TRY
    OLA_ON_DEVICE(ctx);
    require(ctx && out, "null pointer");
    // a multi-device context: every rank has its own pool (also when ranks share a GPU); the figures are the LARGEST over the ranks,
    // i.e. what one GPU of the node has to hold
    for (int i = 0; i < 4; i++) out[i] = 0;
    for (size_t r = 0; r <= ctx->peers.size(); r++) {
        DeviceCtx& d = r == 0 ? ctx->dev : ctx->peers[r - 1]->dev;
        std::lock_guard<std::mutex> lk(d.mu);
        const uint64_t v[4] = {d.live_bytes, d.live_peak, d.live_bytes + d.cached_bytes, d.reserved_peak};
        for (int i = 0; i < 4; i++) out[i] = std::max<uint64_t>(out[i], v[i]);
        if (reset) { d.live_peak = d.live_bytes; d.reserved_peak = d.live_bytes + d.cached_bytes; }
    }
    OLA_CATCH
}

int32_t ola_gpu_proof_stats(OlaCtx* ctx, int32_t enable, double out[8]) {
    OLA_TRY
    require(ctx, "ctx");
    if (enable >= 0) ctx->dev.acct.on = enable != 0;
    if (out) {
        const WorkAcct& a = ctx->dev.acct;
        out[0] = a.wall_ms; out[1] = a.sharded_ms[1]; out[2] = a.sharded_ms[2]; out[3] = a.sharded_ms[3];
        out[4] = (double)a.exchange_bytes; out[5] = (double)a.exchanges;
        out[6] = ctx->group ? (double)ctx->group->exchanges : 0; out[7] = ctx->group ? (double)ctx->group->bytes_moved : 0;
    }
    OLA_CATCH
}

int32_t ola_gpu_phase_stats(OlaCtx* ctx, double* out, uint32_t n_phases) {
    OLA_TRY
    require(ctx && out, "null pointer");
    const WorkAcct& a = ctx->dev.acct;
    for (uint32_t i = 0; i < n_phases; i++) {
        out[3 * i] = i < PH_COUNT ? a.phase_ms[i] : 0;
        out[3 * i + 1] = i < PH_COUNT ? a.phase_units[i][0] : 0;
        out[3 * i + 2] = i < PH_COUNT ? a.phase_units[i][1] : 0;
        if (i >= PH_COUNT && i < 2 * PH_COUNT) {          // rows OLA_PHASE_COUNT + p: phase p's dominant scope (the one that moved the most bytes)
            out[3 * i] = a.phase_top_ms[i - PH_COUNT];
            out[3 * i + 1] = a.phase_top_bytes[i - PH_COUNT];
        }
    }
    OLA_CATCH
}

int32_t ola_gpu_upload_stats(OlaCtx* ctx, double out[8]) {
    OLA_TRY
    require(ctx && out, "null pointer");
    const UploadStats& u = ctx->dev.upload;
    out[0] = u.waited_ms; out[1] = u.total_ms; out[2] = u.first_ms; out[3] = u.bytes; out[4] = (double)u.mode; out[5] = (double)u.threads;
    out[6] = u.link_bytes; out[7] = 0;
    OLA_CATCH
}

int32_t ola_gpu_scope_times(OlaCtx* ctx, int32_t enable, OlaScopeTime* out, uint32_t cap, uint32_t* n_out) {
    OLA_TRY
    require(ctx, "ctx");
    if (enable >= 0) ctx->dev.scopes.on = enable != 0;
    const std::vector<ScopeLog::Rec>& recs = ctx->dev.scopes.recs;
    if (n_out) *n_out = (uint32_t)recs.size();
    if (out) {
        for (size_t i = 0; i < recs.size() && i < cap; i++) {
            const ScopeLog::Rec& r = recs[i];
            OlaScopeTime& o = out[i];
            memset(&o, 0, sizeof o);
            snprintf(o.name, sizeof o.name, "%s", r.name.c_str());
            o.depth = r.depth; o.ref_depth = r.ref_depth; o.table = r.table; o.is_reference_scope = r.ref ? 1 : 0;
            o.start_ms = r.start_ms; o.ms = r.ms; o.sharded_ms = r.sharded_ms;
        }
        require(recs.size() <= cap, "scope buffer too small (*n_out holds the count)");
    }
    OLA_CATCH
}

int32_t ola_gpu_ntt_pass_times(OlaCtx* ctx, int32_t enable, OlaPassTime* out, uint32_t cap, uint32_t* n_out) {
    OLA_TRY
    require(ctx != nullptr, "ctx is NULL");
    OLA_ON_DEVICE(ctx);
    NttTables& t = *ctx->tables;
    HIP_CHECK(hipStreamSynchronize(ctx->dev.stream));
    std::vector<OlaPassTime> agg;
    for (NttTables::PassRec& r : t.pass_recs) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) { (void)hipGetLastError(); ms = 0; }
        char name[sizeof(((OlaPassTime*)0)->kernel)];
        snprintf(name, sizeof(name), "ntt2t_pass_kernel<%d,%d,%s,8,%d>", r.R, r.mode, r.inv ? "true" : "false", r.lm);
        size_t i = 0;
        while (i < agg.size() && strcmp(agg[i].kernel, name) != 0) i++;
        if (i == agg.size()) { OlaPassTime e = {}; memcpy(e.kernel, name, sizeof(name)); agg.push_back(e); }
        agg[i].launches++;
        agg[i].total_ms += ms;
        agg[i].elements += (double)r.elems;
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    t.pass_recs.clear();
    if (enable >= 0) t.pass_timing = enable != 0;
    if (n_out) *n_out = (uint32_t)agg.size();
    if (out) {
        require(cap >= agg.size(), "cap is smaller than the number of pass kernels (*n_out)");
        for (size_t i = 0; i < agg.size(); i++) out[i] = agg[i];
    }
    OLA_CATCH
}

int32_t ola_set_shard_options(OlaCtx* ctx, uint32_t flags) {
    OLA_TRY
    require(ctx, "ctx");
    require((flags & ~OLA_SHARD_STREAM_ORDERED) == 0, "unknown flag");
    ctx->dev.shard.stream_ordered = (flags & OLA_SHARD_STREAM_ORDERED) != 0;
    OLA_CATCH
}

int32_t ola_gpu_get_stream(OlaCtx* ctx, void** stream_out) {
    OLA_TRY
    require(ctx && stream_out, "null pointer");
    *stream_out = (void*)ctx->dev.stream;
    OLA_CATCH
}

int32_t ola_gpu_reserve(OlaCtx* ctx, const uint64_t* airset, size_t airset_words, const uint32_t* log_n) {
    OLA_TRY
    OLA_ON_DEVICE(ctx);
    require(ctx && airset && log_n, "null pointer");
    reserve_for_proof(&ctx->dev, ctx->cfg, (const u64*)airset, airset_words, log_n);
    OLA_CATCH
}

int32_t ola_set_shard(OlaCtx* ctx, uint32_t rank, uint32_t world, ola_all_gather_fn all_gather, void* user) {
    OLA_TRY
    OLA_ON_DEVICE(ctx);
    require(ctx, "ctx");
    require(world >= 1 && world <= 8 && (world & (world - 1)) == 0 && rank < world, "world must be 1, 2, 4 or 8 and rank < world");
    require(ctx->peers.empty(), "a multi-device context partitions the proof itself; ola_set_shard is for one-device contexts of multi-process hosts");
    require(world == 1 || all_gather, "a sharded context needs an all_gather callback");
    ShardInfo sh;
    if (world > 1) {
        sh.rank = rank; sh.world = world;
        while ((1u << sh.log_world) < world) sh.log_world++;
        sh.all_gather = all_gather; sh.user = user;
    }
    ctx->dev.shard = sh;
    OLA_CATCH
}

int32_t ola_air_kernels_available(const uint64_t* airset, size_t airset_words, uint8_t* has_kernel, size_t ntables) {
    OLA_TRY
    require(airset && has_kernel, "null pointer");
    air_kernels_available((const u64*)airset, airset_words, has_kernel, ntables);
    OLA_CATCH
}

}  // extern "C"
