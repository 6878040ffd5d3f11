// Device context shared by all kernels of the backend: one HIP device, one compute stream, a persistent pool for
// tables, and an error slot the C ABI reports through ola_gpu_last_error().  No exceptions cross the C ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <mutex>
#include <set>
#include <thread>
#include <cstdlib>
#include <map>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

namespace ola {

struct OlaError : public std::runtime_error {
    int code;
    OlaError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define HIP_CHECK(expr)                                                                                   \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess)                                                                             \
            throw ::ola::OlaError(-5, std::string(#expr) + ": " + hipGetErrorString(_e) + " at " __FILE__ \
                                          ":" + std::to_string(__LINE__));                                \
    } while (0)

// All device work of a context is issued on ONE stream, so a block returned to the cache may be handed out again
// without synchronising: whatever still reads it was enqueued earlier on the same stream.  Scratch buffers of a 2^22-row
// proof are tens of GB; going back to hipMalloc/hipFree for each of them costs more than the kernels (and hipFree
// synchronises the device), hence the cache.  It is released on out-of-memory and when the context is destroyed.
// Coset partition of a proof over `world` GPUs: every rank runs the same host code on the same (replicated)
// traces, so transcripts agree without exchanging challenges; LDEs, leaf hashing, Merkle sub-trees and quotient
// evaluation are restricted to the rank's cosets, and three small exchanges go through `all_gather`: cap slices, the
// quotient values, and the opened rows / paths of the queries.  The callback is supplied by the host (torch.distributed
// over RCCL in olavm_amd/backend.py): it must gather `bytes` bytes of device memory from every rank into recv (rank
// order) and return 0 once recv is complete.
struct ShardInfo {
    uint32_t rank = 0, world = 1, log_world = 0;
    uint32_t min_log_n = 12;   // smaller tables are proven replicated: sharding them costs more latency than it saves
    int32_t (*all_gather)(void* user, const void* send_dev, void* recv_dev, size_t bytes) = nullptr;
    void* user = nullptr;
    // the callback enqueues the collective ON THE CONTEXT'S STREAM (RCCL through a torch ExternalStream, say): stream order
    // already puts it after the kernels that produced `send` and before the ones that read `recv`, no host synchronisation
    bool stream_ordered = false;
};

// Where a proof's device time goes with respect to the coset partition (ola_gpu_proof_stats): the prover brackets every piece
// of work that the partition DIVIDES among the ranks with a WorkScope; the scopes are timed with events on the context's stream
// (no synchronisation), and the bracketed time is summed per "largest world that still divides it" (a quotient that lives on
// 2 cosets stops scaling at 2 GPUs).  Everything outside a scope -- interpolations, Z columns, FRI after the first layer, small
// tables, launch gaps, host transcript -- is what every rank repeats.  exchange_bytes: the gathered payload of the exchanges a
// sharded run performs (a rank receives (G-1)/G of it); counted on single-GPU runs too, so that one GPU can project G.
// The reference's `timed!` scopes (prover.rs:111-553, fri/oracle.rs:56-90,221-225, fri/prover.rs:41-58) with DEVICE times: when
// switched on (ola_gpu_scope_times), every PhaseTimer brackets its scope with two events on the context's stream -- no
// synchronisation, the host keeps running ahead -- and after the proof's final synchronisation the list holds, per scope, its
// nesting depth, when the GPU entered it (ms since the proof began on the stream), how long the GPU stayed in it, and how much
// of that was work the coset partition divides (WorkScope spans that began inside it).  The Rust shim replays the list into
// the caller's TimingTree (integration/rust/hip_prover.rs); bench.py turns it into the breakdown of the replicated share.
struct ScopeLog {
    struct Rec { std::string name; uint32_t depth, ref_depth; int parent, table; bool ref; hipEvent_t a, b; double start_ms, ms, sharded_ms; };
    static constexpr size_t kMaxScopes = 4096;     // a 12-table proof opens about 330
    bool on = false;
    std::vector<Rec> recs;
    int open = -1;                 // innermost open scope
    uint32_t depth = 0, ref_depth = 0;
    int table = -1;                // set by prove_with_traces around each table's scopes
    hipEvent_t origin = nullptr;
    // a scope the reference has under the same name (the others are this library's grouping: "table 3 prove_single_table", ...)
    static bool is_reference_scope(const std::string& n) {
        static const char* names[] = {"compute trace commitments", "compute permutation Z(x) polys", "compute Zs commitment", "compute quotient polys",
                                      "split quotient polys", "compute quotient commitment", "compute openings proof", "IFFT", "FFT + blinding",
                                      "build Merkle tree", "fold codewords in the commitment phase", "find proof-of-work witness"};
        for (const char* s : names) if (n == s) return true;
        // format!("perform final FFT {}", len), fri/oracle.rs:223 -- not the partitioned variant, which carries a suffix
        return n.rfind("perform final FFT ", 0) == 0 && n.find('(') == std::string::npos;
    }
};

// kernel families the accounting also times one by one (ola_gpu_phase_stats): device milliseconds + two unit counters each
enum { PH_LEAF_HASH = 0, PH_MERKLE_LEVELS, PH_FRI_FOLD, PH_LDE, PH_INTT, PH_QUOTIENT, PH_OPEN_EVAL, PH_COUNT };
struct WorkAcct {
    bool on = false;
    struct Span { hipEvent_t a, b; int maxlog; int scope; double bytes = 0; };   // maxlog >= 100: a phase span, phase = maxlog - 100; scope: ScopeLog index or -1
    double phase_ms[PH_COUNT] = {};
    double phase_units[PH_COUNT][2] = {};
    // the scope of each phase that moved the most bytes (its dominant launch): device time and bytes -- a phase's total hides it behind
    // the launch-bound small ones (a proof has thirty folds, two of them large)
    double phase_top_ms[PH_COUNT] = {}, phase_top_bytes[PH_COUNT] = {};
    std::vector<Span> spans;
    std::vector<hipEvent_t> spare;
    double sharded_ms[4] = {0, 0, 0, 0};   // [k]: work that divides by min(G, 2^k), k = 1..3
    double wall_ms = 0;
    uint64_t exchange_bytes = 0;
    uint32_t exchanges = 0;
    bool shardable = false;                // the table being proven is large enough for the partition (ShardInfo::min_log_n)
    hipEvent_t get() {
        if (!spare.empty()) { hipEvent_t e = spare.back(); spare.pop_back(); return e; }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        return e;
    }
    // Spans are only meaningful between begin_proof() and collect(): entry points that run a PhaseScope outside a proof (a host
    // driving ola_commit_* / ola_ntt_* directly with the accounting on) would otherwise grow `spans` without bound and have their
    // time charged to the next proof.  begin_proof() drops whatever is outstanding (events go back to `spare`), and outside a
    // proof the list is capped: past the cap the oldest spans are recycled unread.
    static constexpr size_t kMaxIdleSpans = 256;
    void recycle_spans() {
        for (Span& s : spans) { if (s.a) spare.push_back(s.a); if (s.b) spare.push_back(s.b); }
        spans.clear();
    }
    void add_span(const Span& s) {
        if (spans.size() >= (in_proof ? (size_t)1 << 16 : kMaxIdleSpans)) recycle_spans();   // in_proof stays set if a proof threw

        spans.push_back(s);
    }
    bool in_proof = false;
    void begin_proof() {
        recycle_spans();
        in_proof = true;
        for (double& m : sharded_ms) m = 0;
        wall_ms = 0; exchange_bytes = 0; exchanges = 0;
        for (int i = 0; i < PH_COUNT; i++) { phase_ms[i] = 0; phase_units[i][0] = phase_units[i][1] = 0; phase_top_ms[i] = phase_top_bytes[i] = 0; }
    }
    // after the stream has been synchronised
    void collect(ScopeLog* log = nullptr) {
        for (Span& s : spans) {
            float ms = 0;
            if (s.a && s.b && hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
                if (s.maxlog >= 100) {
                    phase_ms[s.maxlog - 100] += ms;
                    if (s.bytes > phase_top_bytes[s.maxlog - 100]) { phase_top_bytes[s.maxlog - 100] = s.bytes; phase_top_ms[s.maxlog - 100] = ms; }
                }
                else {
                    sharded_ms[s.maxlog] += ms;
                    if (log) for (int i = s.scope; i >= 0 && i < (int)log->recs.size(); i = log->recs[(size_t)i].parent) log->recs[(size_t)i].sharded_ms += ms;
                }
            }
            else (void)hipGetLastError();
            if (s.a) spare.push_back(s.a);
            if (s.b) spare.push_back(s.b);
        }
        spans.clear();
        in_proof = false;
    }
    ~WorkAcct() { collect(); for (hipEvent_t e : spare) (void)hipEventDestroy(e); }
};

// what the last upload did (ola_gpu_upload_stats)
struct UploadStats {
    double waited_ms = 0;       // the proving thread blocked this long for column groups
    double total_ms = 0;        // first byte asked for -> last byte on the device
    double first_ms = 0;        // until the first column group was complete
    double bytes = 0;
    double link_bytes = 0;      // bytes that crossed the link (narrow columns travel as 32-bit words)
    uint32_t mode = 0;          // 0 staged, 1 pageable
    uint32_t threads = 0;
};

struct DeviceCtx;
// The contexts of this process, for the hand-over of cached blocks between the ones that share a GPU (DeviceCtx::adopt).
struct PoolRegistry {
    std::mutex mu;
    std::vector<DeviceCtx*> all;
    static PoolRegistry& get() { static PoolRegistry* r = new PoolRegistry(); return *r; }   // never destroyed: contexts may outlive static teardown
};

struct DeviceCtx {
    ShardInfo shard;
    WorkAcct acct;
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    int hasher = 0;                            // OLA_HASH_POSEIDON / OLA_HASH_BLAKE3: GenericConfig::Hasher of the Merkle trees and the challenger
    bool timing = false;                       // OLA_TIMING=1: per-phase wall-clock on stderr (synchronises at phase edges)
    bool priming = false;                      // ola_gpu_warmup's throw-away proof of an all-zero instance: the divisibility check is off
    std::vector<void*> persistent;
    std::multimap<size_t, void*> cache;        // free blocks by size
    std::unordered_map<void*, size_t> live;    // blocks handed out
    size_t cached_bytes = 0;
    size_t live_bytes = 0, live_peak = 0, reserved_peak = 0;   // handed out now / at most; handed out + cached at most
    // Background reservation (ola_gpu_reserve): a helper thread hipMallocs the blocks a coming proof will ask for and puts
    // them into the cache, so that the driver's scrubbing of previously used VRAM (about 30 ms per GB on this stack, paid inside
    // hipMalloc) overlaps whatever the caller does between creating the context and proving.  `mu` guards cache / live / pending.
    std::mutex mu;
    std::condition_variable cv;
    std::multiset<size_t> pending;             // sizes the helper has not delivered yet
    std::thread reserver;
    // Hand-over between the contexts of one process that share a GPU (a host that keeps a Poseidon and a Blake3 context, say): a
    // context that misses in its own cache takes a fitting block out of the cache of an IDLE sibling instead of going to hipMalloc
    // (which scrubs recycled VRAM at about 30 ms per GB and would make the two pools add up).  A sibling lends only when
    //   - it takes part (`lends`: single-device contexts; OLA_POOL_SHARE=0 switches the hand-over off),
    //   - none of the C-ABI calls is running on it (`calls`: its upload / peer streams and helper threads live inside calls),
    //   - its stream has drained (hipStreamQuery; also asked when both share a stream -- the borrower's upload stream is not
    //     ordered behind it),
    // all checked under the sibling's `mu`: whatever is in its cache then was freed by finished calls whose work has completed.
    std::atomic<int> calls{0};
    std::atomic<int> pins{0};                  // siblings looking at this context's cache outside the registry lock (adopt / release_idle_siblings)
    bool lends = false;
    // pinned staging ring of the trace upload (upload.h), kept from proof to proof: hipHostMalloc costs milliseconds per 100 MB
    void* staging = nullptr;
    size_t staging_bytes = 0;
    void* staging_dev = nullptr;               // device side of the narrow-column path (upload.h)
    size_t staging_dev_bytes = 0;
    UploadStats upload;
    // Pinned host memory for the small read-backs and uploads of the proof path (caps, opening values, query rows, descriptors).
    // A copy to or from PAGEABLE host memory is staged by the runtime on the calling thread -- a read-back of 512 bytes is then a
    // hidden synchronisation of 28 us, 16 us from pinned memory (tools/ubench/roundtrip.hip) -- and the small tables of a proof
    // are bound by exactly these.  A stack: scopes (DevBuf) mark the top when they open and restore it when they close; the first
    // kPinnedScratch bytes are a fixed slot for read-backs that are waited for at once (batch_read_cap).
    // a second stream for work nothing on the main stream waits for (the proof-of-work search of a table: fri.hip PowDefer)
    hipStream_t side = nullptr;
    hipStream_t side_stream() {
        if (!side && hipStreamCreateWithFlags(&side, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); side = nullptr; }
        return side;
    }
    // device words of the side stream's searches: persistent, never pool blocks (a pool block may still be in use by earlier work
    // of the MAIN stream when it is handed out again -- safe for main-stream users only)
    unsigned long long* side_words = nullptr;
    static constexpr size_t kSideWords = 256;
    static constexpr size_t kPinnedBytes = (size_t)16 << 20, kPinnedScratch = 4096;
    char* pinned = nullptr;
    size_t pinned_top = kPinnedScratch;
    bool pinned_failed = false;
    char* pinned_base() {
        if (!pinned && !pinned_failed) {
            void* p = nullptr;
            if (hipHostMalloc(&p, kPinnedBytes, hipHostMallocDefault) == hipSuccess) pinned = (char*)p;
            else { (void)hipGetLastError(); pinned_failed = true; }
        }
        return pinned;
    }
    // nullptr when the arena is exhausted (or could not be pinned): the caller falls back to pageable memory
    void* pinned_alloc(size_t bytes) {
        bytes = (bytes + 63) / 64 * 64;
        if (!pinned_base() || pinned_top + bytes > kPinnedBytes) return nullptr;
        void* p = pinned + pinned_top;
        pinned_top += bytes;
        return p;
    }
    ScopeLog scopes;
    size_t adopted_bytes = 0, adopted_blocks = 0;
    // scopes of one proof: begin on the stream, resolve after the final synchronisation
    void scopes_begin() {
        ScopeLog& L = scopes;
        for (ScopeLog::Rec& r : L.recs) { if (r.a) acct.spare.push_back(r.a); if (r.b) acct.spare.push_back(r.b); r.a = r.b = nullptr; }
        L.recs.clear(); L.open = -1; L.depth = L.ref_depth = 0; L.table = -1;
        if (!L.on) return;
        if (!L.origin) L.origin = acct.get();
        if (L.origin && hipEventRecord(L.origin, stream) != hipSuccess) (void)hipGetLastError();
    }
    void scopes_collect() {
        ScopeLog& L = scopes;
        for (ScopeLog::Rec& r : L.recs) {
            float s = 0, m = 0;
            if (L.origin && r.a && r.b && hipEventElapsedTime(&s, L.origin, r.a) == hipSuccess && hipEventElapsedTime(&m, r.a, r.b) == hipSuccess) { r.start_ms = s; r.ms = m; }
            else (void)hipGetLastError();
            if (r.a) acct.spare.push_back(r.a);
            if (r.b) acct.spare.push_back(r.b);
            r.a = r.b = nullptr;
        }
    }
    void join_pool_registry() {
        const char* e = getenv("OLA_POOL_SHARE");
        lends = !(e && *e == '0');
        PoolRegistry& R = PoolRegistry::get();
        std::lock_guard<std::mutex> lk(R.mu);
        R.all.push_back(this);
    }
    void leave_pool_registry() {
        PoolRegistry& R = PoolRegistry::get();
        {
            std::lock_guard<std::mutex> lk(R.mu);
            for (size_t i = 0; i < R.all.size(); i++)
                if (R.all[i] == this) { R.all.erase(R.all.begin() + (long)i); break; }
        }
        // a sibling that took this context off the list before it left may still be asking its stream: wait for it (microseconds)
        while (pins.load() != 0) std::this_thread::yield();
    }
    // the siblings on this device that lend, pinned so that they stay alive after the registry lock is released: the runtime calls
    // that follow (hipStreamQuery, hipFree) must not run under the one lock every context of the process needs to come and go
    std::vector<DeviceCtx*> pin_siblings() {
        std::vector<DeviceCtx*> v;
        PoolRegistry& R = PoolRegistry::get();
        std::lock_guard<std::mutex> rk(R.mu);
        for (DeviceCtx* o : R.all)
            if (o != this && o->lends && o->device == device) { o->pins.fetch_add(1); v.push_back(o); }
        return v;
    }
    static void unpin(const std::vector<DeviceCtx*>& v) { for (DeviceCtx* o : v) o->pins.fetch_sub(1); }
    // true when `o` (locked by the caller) may hand blocks of its cache to this context right now
    bool sibling_is_idle(DeviceCtx* o) {
        if (o->calls.load() != 0) return false;
        // also when the lender runs on the borrower's own stream: the borrower's upload stream writes into fresh blocks without
        // being ordered behind that stream's pending kernels (an asynchronous entry point frees scratch with work in flight)
        const hipError_t q = hipStreamQuery(o->stream);
        if (q != hipSuccess) { (void)hipGetLastError(); return false; }
        return true;
    }
    void* adopt(size_t want, size_t& got) {
        if (!lends) return nullptr;
        const std::vector<DeviceCtx*> sib = pin_siblings();
        void* p = nullptr;
        for (DeviceCtx* o : sib) {
            std::unique_lock<std::mutex> lk(o->mu, std::try_to_lock);
            if (!lk.owns_lock()) continue;
            auto it = o->cache.lower_bound(want);
            if (it == o->cache.end() || it->first > want + want / 4) continue;
            if (!sibling_is_idle(o)) continue;
            p = it->second;
            got = it->first;
            o->cached_bytes -= got;
            o->cache.erase(it);
            break;
        }
        unpin(sib);
        return p;
    }
    // out of memory: the cached blocks of idle siblings on this device go back to the driver
    size_t release_idle_siblings() {
        if (!lends) return 0;
        size_t freed = 0;
        const std::vector<DeviceCtx*> sib = pin_siblings();
        for (DeviceCtx* o : sib) {
            std::unique_lock<std::mutex> lk(o->mu, std::try_to_lock);
            if (!lk.owns_lock() || o->cache.empty() || !sibling_is_idle(o)) continue;
            for (auto& kv : o->cache) (void)hipFree(kv.second);
            freed += o->cached_bytes;
            o->cache.clear();
            o->cached_bytes = 0;
        }
        unpin(sib);
        if (freed && timing) fprintf(stderr, "[ola-timing] device allocator: out of memory, released %.1f GB cached by idle contexts on this GPU\n", freed / 1e9);
        return freed;
    }
    void reserve_async(std::vector<size_t> sizes) {
        join_reserver();
        {
            std::lock_guard<std::mutex> lk(mu);
            for (size_t& sz : sizes) { sz = round_size(sz); pending.insert(sz); }
        }
        reserver = std::thread([this, sizes] {
            (void)hipSetDevice(device);
            for (size_t sz : sizes) {
                // an idle sibling's cached block first (the hand-over of alloc()), the driver otherwise
                size_t got = sz;
                void* p = adopt(sz, got);
                hipError_t e = hipSuccess;
                const bool adopted = p != nullptr;
                if (!p) { got = sz; e = hipMalloc(&p, sz); }
                std::lock_guard<std::mutex> lk(mu);
                if (adopted) { adopted_bytes += got; adopted_blocks++; }
                pending.erase(pending.find(sz));
                if (e == hipSuccess) { cache.emplace(got, p); cached_bytes += got; if (live_bytes + cached_bytes > reserved_peak) reserved_peak = live_bytes + cached_bytes; }
                else (void)hipGetLastError();
                cv.notify_all();
            }
        });
    }
    void join_reserver() { if (reserver.joinable()) reserver.join(); }
    void note_alloc(size_t sz) {
        live_bytes += sz;
        if (live_bytes > live_peak) live_peak = live_bytes;
        if (live_bytes + cached_bytes > reserved_peak) reserved_peak = live_bytes + cached_bytes;
    }

    static size_t round_size(size_t bytes) {
        const size_t g = bytes >= (1u << 20) ? (2u << 20) : 256;
        return ((bytes ? bytes : 8) + g - 1) / g * g;
    }
    void* alloc_persistent(size_t bytes) {
        void* p = nullptr;
        HIP_CHECK(hipMalloc(&p, bytes ? bytes : 8));
        persistent.push_back(p);
        return p;
    }
    void release_cache() {
        join_reserver();
        std::lock_guard<std::mutex> lk(mu);
        if (cache.empty()) return;
        if (timing) fprintf(stderr, "[ola-timing] device allocator: out of memory, releasing %.1f GB of cached blocks\n", cached_bytes / 1e9);
        (void)hipStreamSynchronize(stream);
        for (auto& kv : cache) (void)hipFree(kv.second);
        cache.clear();
        cached_bytes = 0;
    }
    void* alloc(size_t bytes) {
        const size_t want = round_size(bytes);
        {
            std::unique_lock<std::mutex> lk(mu);
            for (;;) {
                auto it = cache.lower_bound(want);
                if (it != cache.end() && it->first <= want + want / 4) {   // close enough fit
                    void* p = it->second;
                    live[p] = it->first;
                    cached_bytes -= it->first;
                    note_alloc(it->first);
                    cache.erase(it);
                    return p;
                }
                auto pd = pending.lower_bound(want);                       // a fitting block is on its way: wait for it
                if (pd != pending.end() && *pd <= want + want / 4) { cv.wait(lk); continue; }
                break;
            }
        }
        void* p = nullptr;
        {
            size_t got = 0;
            p = adopt(want, got);
            if (p) {
                std::lock_guard<std::mutex> lk(mu);
                live[p] = got;
                note_alloc(got);
                adopted_bytes += got;
                adopted_blocks++;
                return p;
            }
        }
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            release_cache();
            e = hipMalloc(&p, want);
        }
        if (e != hipSuccess && release_idle_siblings() > 0) {
            (void)hipGetLastError();
            e = hipMalloc(&p, want);
        }
        if (e != hipSuccess) throw OlaError(-3, std::string("hipMalloc(") + std::to_string(want) + "): " + hipGetErrorString(e));
        std::lock_guard<std::mutex> lk(mu);
        live[p] = want;
        note_alloc(want);
        return p;
    }
    void free(void* p) {
        if (!p) return;
        std::lock_guard<std::mutex> lk(mu);
        auto it = live.find(p);
        if (it == live.end()) { (void)hipFree(p); return; }
        cache.emplace(it->second, p);
        cached_bytes += it->second;
        live_bytes -= it->second;
        live.erase(it);
    }
    ~DeviceCtx() {
        for (ScopeLog::Rec& r : scopes.recs) { if (r.a) acct.spare.push_back(r.a); if (r.b) acct.spare.push_back(r.b); }
        if (scopes.origin) acct.spare.push_back(scopes.origin);      // ~WorkAcct destroys the pool of events
        leave_pool_registry();
        join_reserver();
        if (stream) (void)hipStreamSynchronize(stream);
        for (auto& kv : cache) (void)hipFree(kv.second);
        for (auto& kv : live) (void)hipFree(kv.first);
        for (void* p : persistent) (void)hipFree(p);
        if (staging) (void)hipHostFree(staging);
        if (side) { (void)hipStreamSynchronize(side); (void)hipStreamDestroy(side); }
        if (pinned) (void)hipHostFree(pinned);
        if (staging_dev) (void)hipFree(staging_dev);
        if (owns_stream && stream) (void)hipStreamDestroy(stream);
    }
};

// Brackets work that the coset partition divides among up to 2^maxlog ranks (WorkAcct); active only while accounting is on and
// the table being proven is on the partition.
struct WorkScope {
    DeviceCtx* ctx;
    hipEvent_t a = nullptr;
    int maxlog, scope = -1;
    WorkScope(DeviceCtx* c, int maxlog_) : ctx(c), maxlog(maxlog_ < 1 ? 0 : (maxlog_ > 3 ? 3 : maxlog_)) {
        if (!ctx->acct.on || !ctx->acct.shardable || maxlog == 0) { maxlog = 0; return; }
        if (ctx->scopes.on) scope = ctx->scopes.open;
        a = ctx->acct.get();
        if (a && hipEventRecord(a, ctx->stream) != hipSuccess) { (void)hipGetLastError(); ctx->acct.spare.push_back(a); a = nullptr; }
    }
    ~WorkScope() {
        if (!a) return;
        hipEvent_t b = ctx->acct.get();
        if (b && hipEventRecord(b, ctx->stream) != hipSuccess) { (void)hipGetLastError(); ctx->acct.spare.push_back(b); b = nullptr; }
        ctx->acct.add_span({a, b, maxlog, scope});
    }
};
// Times one kernel family of a proof (WorkAcct::phase_ms) and counts what it processed; nests freely with WorkScope.
struct PhaseScope {
    DeviceCtx* ctx;
    hipEvent_t a = nullptr;
    int phase;
    double bytes = 0;
    PhaseScope(DeviceCtx* c, int phase_, double units0, double units1 = 0) : ctx(c), phase(phase_) {
        if (!ctx->acct.on) return;
        static const int bytes_at[PH_COUNT] = {1, 1, 0, 0, 0, 1, 1};       // which of a phase's two unit counters is its bytes (ola_gpu.h)
        bytes = bytes_at[phase] ? units1 : units0;
        ctx->acct.phase_units[phase][0] += units0;
        ctx->acct.phase_units[phase][1] += units1;
        a = ctx->acct.get();
        if (a && hipEventRecord(a, ctx->stream) != hipSuccess) { (void)hipGetLastError(); ctx->acct.spare.push_back(a); a = nullptr; }
    }
    ~PhaseScope() {
        if (!a) return;
        hipEvent_t b = ctx->acct.get();
        if (b && hipEventRecord(b, ctx->stream) != hipSuccess) { (void)hipGetLastError(); ctx->acct.spare.push_back(b); b = nullptr; }
        ctx->acct.add_span({a, b, 100 + phase, -1, bytes});
    }
};
// an exchange of the partition: `gathered_bytes` = payload of all ranks together
inline void acct_exchange(DeviceCtx* ctx, size_t gathered_bytes) {
    if (!ctx->acct.on) return;
    ctx->acct.exchange_bytes += gathered_bytes;
    ctx->acct.exchanges++;
}

// all-gather of device buffers through the host-supplied collective (ShardInfo)
inline void shard_all_gather(DeviceCtx* ctx, const void* send_dev, void* recv_dev, size_t bytes) {
    acct_exchange(ctx, bytes * ctx->shard.world);
    if (!ctx->shard.all_gather) throw OlaError(-1, "sharded proving needs ola_set_shard with an all_gather callback");
    if (!ctx->shard.stream_ordered) HIP_CHECK(hipStreamSynchronize(ctx->stream));
    const int32_t rc = ctx->shard.all_gather(ctx->shard.user, send_dev, recv_dev, bytes);
    if (rc != 0) throw OlaError(-7, "all_gather callback failed with code " + std::to_string(rc));
}

// wall-clock of one phase of the prover (the reference's `timed!` scopes, prover.rs), printed when ctx->timing is set
struct PhaseTimer {
    DeviceCtx* ctx;
    std::string name;
    std::chrono::steady_clock::time_point t0;
    int rec = -1;
    PhaseTimer(DeviceCtx* c, const std::string& n) : ctx(c), name(n) {
        // (a bound on the list: entry points outside a whole proof open scopes too and nobody resets the list between them)
        if (ctx->scopes.on && ctx->scopes.recs.size() < ScopeLog::kMaxScopes) {
            ScopeLog& L = ctx->scopes;
            const std::string bare = n.substr(std::min(n.size(), n.find_first_not_of(' ')));
            ScopeLog::Rec r{bare, L.depth, L.ref_depth, L.open, L.table, ScopeLog::is_reference_scope(bare), ctx->acct.get(), nullptr, 0, 0, 0};
            if (r.a && hipEventRecord(r.a, ctx->stream) != hipSuccess) { (void)hipGetLastError(); ctx->acct.spare.push_back(r.a); r.a = nullptr; }
            rec = (int)L.recs.size();
            L.recs.push_back(r);
            L.open = rec; L.depth++; if (r.ref) L.ref_depth++;
        }
        if (ctx->timing) { (void)hipStreamSynchronize(ctx->stream); t0 = std::chrono::steady_clock::now(); }
    }
    ~PhaseTimer() {
        if (rec >= 0 && rec < (int)ctx->scopes.recs.size()) {
            ScopeLog& L = ctx->scopes;
            ScopeLog::Rec& r = L.recs[(size_t)rec];
            r.b = ctx->acct.get();
            if (r.b && hipEventRecord(r.b, ctx->stream) != hipSuccess) { (void)hipGetLastError(); ctx->acct.spare.push_back(r.b); r.b = nullptr; }
            L.open = r.parent; L.depth = r.depth; L.ref_depth = r.ref_depth;
        }
        if (!ctx->timing) return;
        (void)hipStreamSynchronize(ctx->stream);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        fprintf(stderr, "[ola-timing] %-44s %9.3f ms\n", name.c_str(), ms);
    }
};

}  // namespace ola
