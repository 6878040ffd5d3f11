// Trace upload of prove_with_traces: the caller's tables -- [Vec<PolynomialValues<F>>; NUM_TABLES] in the reference
// (circuits/src/stark/prover.rs:79-83; one Vec<F> per column, plonky2/field/src/polynomial/mod.rs:24-26; GoldilocksField is
// repr(transparent) u64, goldilocks_field.rs:24-26) -- go to the prover's column-major device buffers while the proving thread
// already interpolates / extends / hashes the column groups that have arrived.
//
// A column is wherever the caller's allocator put it (TraceSource::cols) or part of one contiguous block (TraceSource::base).
// Host memory is pageable; hipMemcpyAsync from pageable memory is staged by the runtime on the calling thread, one chunk in
// flight (round 4: 40 GB/s, the proving thread waited 62 - 82 ms of a 261 ms proof).  Here the staging is the library's own:
// a ring of pinned slots kept by the context, filled by a few copier threads (one core copies 29 GB/s on the test box, the
// link carries 56.8), every filled slot sent with hipMemcpyAsync right away on one of two upload streams in turn -- the
// next copy is already queued on the other DMA engine when one ends; with one stream the gap between copies costs 18 us per
// piece, 45 instead of 56 GB/s at 4 MB pieces -- and one thread that retires slots in order and publishes "columns [0, c) of
// table t have arrived".  Tables are sent in the order the caller gives (prove_with_traces: small tables first, then the large
// ones in descending size, so that what is left to do after the last byte is the smallest large table's Merkle tree).
//   OLA_UPLOAD=staged (default) | pageable (round 4's path: hipMemcpyAsync from the caller's memory, kept as the A/B control)
//   OLA_UPLOAD_THREADS (default 4), OLA_UPLOAD_PIECE_MB (16), OLA_UPLOAD_SLOTS (8), OLA_UPLOAD_STREAMS (2), OLA_UPLOAD_PACK (1), OLA_UPLOAD_SOLO_KB (512), OLA_UPLOAD_PACK8 (1)
// Narrow columns (OLA_UPLOAD_PACK, default on): most columns of an execution trace hold small values -- selectors, opcodes,
// addresses, clocks, 32-bit limbs.  While a copier thread fills a slot with a piece of ONE column it checks, block by block,
// whether every word of the piece is below 2^32; if so only the low halves go into the slot, half the bytes cross the link,
// and a kernel on the upload stream widens them into the prover's buffer (through a small device-side ring).  The first word
// of 2^32 or more ends the attempt (a column of field-sized values fails in its first block) and the piece travels as 64-bit
// words.  The device buffer holds the same canonical words either way.
// Round 6: (a) a column can only be packed when it has a piece to itself, and columns shorter than a slot used to share one -- a
// 2^20-row table (8 MB columns, two per slot) sent 8 % fewer bytes instead of 40 %: columns of OLA_UPLOAD_SOLO_KB (512 KB) and
// more now get their own piece; (b) a column whose words are all below 2^8 (selectors, flags, opcode bits: 82 % of the words of
// an executed Fibonacci trace) travels as BYTES (OLA_UPLOAD_PACK8, tried first).  README-shape proof: 1.12 -> 0.69 -> 0.3 GB over
// the link.
// Tables that are already in device memory are copied device to device without staging.
// Not built: hipHostRegister of the caller's columns -- pinning 4.29 GB took 212.8 ms on the test box (50 ms per GB, three
// times the transfer itself; tools/ubench/h2d_rates.hip, profiles/r05_h2d_rates.txt), and page-rounded registrations take in
// neighbouring heap objects, after which the runtime refuses copies that straddle the registration's edge.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "device_ctx.h"

namespace ola {

typedef unsigned long long u64;   // as in gl.cuh

// where one table's columns are: `cols[c]` (each its own allocation) or `base + c n` (one block)
void* take_warm_ring(int device, size_t want, size_t* got);   // ola_gpu.hip

struct TraceSource {
    const u64* base = nullptr;
    const u64* const* cols = nullptr;
    const u64* col(uint32_t c, size_t n) const { return cols ? cols[c] : base + (size_t)c * n; }
};

// widens the low halves a copier thread packed into 64-bit words (narrow columns)
__global__ __launch_bounds__(256) void upload_widen_kernel(const uint32_t* __restrict__ in, u64* __restrict__ out, size_t n) {
    const size_t i0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i0 + 4 <= n) {
        const uint4 v = *reinterpret_cast<const uint4*>(in + i0);
        out[i0] = v.x; out[i0 + 1] = v.y; out[i0 + 2] = v.z; out[i0 + 3] = v.w;
    } else {
        for (size_t i = i0; i < n; i++) out[i] = in[i];
    }
}

// the same for columns whose words are all below 2^8 (selectors, flags, opcode bits: 82 % of the words of an executed Fibonacci trace)
__global__ __launch_bounds__(256) void upload_widen8_kernel(const uint8_t* __restrict__ in, u64* __restrict__ out, size_t n) {
    const size_t i0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i0 + 8 <= n) {
        const uint2 v = *reinterpret_cast<const uint2*>(in + i0);
        ulonglong2* o = reinterpret_cast<ulonglong2*>(out + i0);
        o[0] = make_ulonglong2(v.x & 0xFF, (v.x >> 8) & 0xFF);
        o[1] = make_ulonglong2((v.x >> 16) & 0xFF, v.x >> 24);
        o[2] = make_ulonglong2(v.y & 0xFF, (v.y >> 8) & 0xFF);
        o[3] = make_ulonglong2((v.y >> 16) & 0xFF, v.y >> 24);
    } else {
        for (size_t i = i0; i < n; i++) out[i] = in[i];
    }
}

class TraceUploader {
  public:
    enum Mode { STAGED = 0, PAGEABLE = 1 };
    TraceUploader(DeviceCtx* ctx, size_t ntables) : ctx_(ctx), jobs_(ntables), done_(ntables), narrow_(ntables) {
        for (auto& d : done_) d.store(0);
        const char* m = getenv("OLA_UPLOAD");
        mode_ = (m && !strcmp(m, "pageable")) ? PAGEABLE : STAGED;
        piece_bytes_ = (size_t)env_int("OLA_UPLOAD_PIECE_MB", 16, 1, 256) << 20;
        slots_ = (size_t)env_int("OLA_UPLOAD_SLOTS", 8, 2, 1024);
        nstreams_ = (size_t)env_int("OLA_UPLOAD_STREAMS", 2, 1, 4);
        pack_ = env_int("OLA_UPLOAD_PACK", 1, 0, 1) != 0;
        solo_bytes_ = (size_t)env_int("OLA_UPLOAD_SOLO_KB", 512, 0, 1 << 20) << 10;
        pack8_ = env_int("OLA_UPLOAD_PACK8", 1, 0, 1) != 0;
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        // the ranks of a multi-device context upload side by side: keep the copier threads of all of them within the machine
        const unsigned share = std::max(1u, hw / std::max(1u, ctx->shard.world));
        nthreads_ = (unsigned)env_int("OLA_UPLOAD_THREADS", (int)std::min(4u, std::max(1u, share - 1)), 1, 64);
    }
    ~TraceUploader() {
        cancel_.store(true);
        { std::lock_guard<std::mutex> lk(mu_); cv_.notify_all(); }
        join_all();
        // error path: copies and widen kernels already queued still write into the destination buffers, which the unwinding caller
        // is about to hand back to the context's cache -- drain the upload streams before that
        for (hipStream_t st : streams_) if (st) (void)hipStreamSynchronize(st);
        for (hipEvent_t e : events_) if (e) (void)hipEventDestroy(e);
        if (start_ev_) (void)hipEventDestroy(start_ev_);
        for (hipStream_t st : streams_) if (st) (void)hipStreamDestroy(st);
    }
    // columns [first, first + ncols) of `src` -> dst (column c of the job at dst + c n)
    void add(size_t t, const TraceSource& src, uint32_t first, u64* dst, uint32_t ncols, size_t n) {
        const size_t target = (size_t)64 << 20;
        uint32_t cc = (uint32_t)std::max<size_t>(1, target / (n * 8));
        jobs_[t] = {src, first, dst, ncols, n, std::max(1u, std::min(cc, ncols))};      // ncols may be 0: a rank without columns of its own
        narrow_[t].reset(new std::atomic<char>[ncols ? ncols : 1]);
        for (uint32_t c = 0; c < ncols; c++) narrow_[t][c].store(1);
    }
    // after wait(t, c + 1): every word of column c went over the link as a 32-bit word, i.e. the column holds canonical values already
    bool column_is_narrow(size_t t, uint32_t c) const { return mode_ == STAGED && pack_ && narrow_[t] && narrow_[t][c].load() != 0; }
    // granularity the proving thread should ask in (about 64 MB of columns)
    uint32_t chunk_cols(size_t t) const { return jobs_[t].chunk; }
    // tables are sent in this order (default: as numbered)
    void set_order(const std::vector<size_t>& order) { order_ = order; }
    void start() {
        streams_.assign(mode_ == STAGED ? nstreams_ : 1, nullptr);
        for (hipStream_t& st : streams_) HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        // the destination buffers come out of the context's cache: whatever was enqueued on the context's stream before this call
        // (an asynchronous entry point that freed its scratch with kernels still pending) must have finished with them first
        HIP_CHECK(hipEventCreateWithFlags(&start_ev_, hipEventDisableTiming));
        HIP_CHECK(hipEventRecord(start_ev_, ctx_->stream));
        for (hipStream_t st : streams_) HIP_CHECK(hipStreamWaitEvent(st, start_ev_, 0));
        pieces_plan_ = mode_;
        plan();
        t_start_ = std::chrono::steady_clock::now();
        if (mode_ == STAGED && !pieces_.empty()) {
            // a host that cannot pin another 128 MB (locked-memory limit) still proves: through the runtime's own staging
            try { ensure_ring(); } catch (const OlaError&) { (void)hipGetLastError(); mode_ = PAGEABLE; }
        }
        if (mode_ == PAGEABLE && !pieces_.empty() && pieces_plan_ == STAGED) { pieces_.clear(); bytes_ = 0; pieces_plan_ = PAGEABLE; plan(); }
        if (mode_ == STAGED && !pieces_.empty()) {
            if (pack_) { try { ensure_device_ring(); } catch (const OlaError&) { (void)hipGetLastError(); pack_ = false; } }
            events_.assign(slots_, nullptr);
            for (auto& e : events_) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            issued_.reset(new std::atomic<char>[pieces_.size()]);
            for (size_t i = 0; i < pieces_.size(); i++) issued_[i].store(0);
            for (unsigned k = 0; k < nthreads_; k++) copiers_.emplace_back([this] { copier(); });
            th_ = std::thread([this] { retire(); });
        } else {
            th_ = std::thread([this] { run_direct(); });
        }
    }
    void wait(size_t t, uint32_t cols) {
        const auto t0 = std::chrono::steady_clock::now();
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return done_[t].load() >= cols || failed_; });
        waited_ms_ += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (failed_) throw OlaError(-5, "trace upload failed: " + error_);
    }
    void finish() {
        join_all();
        UploadStats& s = ctx_->upload;
        s.waited_ms = waited_ms_; s.total_ms = total_ms_; s.first_ms = first_ms_; s.bytes = (double)bytes_; s.mode = (uint32_t)mode_;
        s.threads = mode_ == STAGED ? nthreads_ : 1;
        s.link_bytes = mode_ == STAGED ? (double)link_bytes_.load() : (double)bytes_;
        if (ctx_->timing)
            fprintf(stderr, "[ola-timing] trace upload (%s, %u copier thread(s)): %.2f GB of trace (%.2f GB over the link: narrow columns travel as 32-bit words or bytes) in %.3f ms = %.1f GB/s of trace; the proving thread waited %.3f ms for column groups; first group complete after %.3f ms\n",
                    mode_ == STAGED ? "pinned staging ring" : "pageable hipMemcpyAsync", s.threads, bytes_ / 1e9, s.link_bytes / 1e9, total_ms_,
                    total_ms_ > 0 ? bytes_ / 1e6 / total_ms_ : 0.0, waited_ms_, first_ms_);
        if (failed_) throw OlaError(-5, "trace upload failed: " + error_);
    }

  private:
    struct Job { TraceSource src; uint32_t first; u64* dst; uint32_t ncols; size_t n; uint32_t chunk; };
    // one staging slot's worth: whole columns [c0, c1) of a table, or rows [row0, row0 + rows) of the single column c0
    struct Piece { uint32_t table, c0, c1; size_t row0, rows; uint32_t cols_done; bool device_src; };

    static int env_int(const char* name, int dflt, int lo, int hi) {
        const char* e = getenv(name);
        if (!e || !*e) return dflt;
        return std::max(lo, std::min(hi, atoi(e)));
    }
    static bool is_device_pointer(const void* p) {
        hipPointerAttribute_t a;
        if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }   // plain malloc memory: "invalid value"
        return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
    }
    void plan() {
        if (order_.size() != jobs_.size()) { order_.resize(jobs_.size()); for (size_t t = 0; t < jobs_.size(); t++) order_[t] = t; }
        for (size_t t : order_) {
            const Job& j = jobs_[t];
            if (!j.ncols) continue;
            const size_t col_bytes = j.n * 8;
            bytes_ += col_bytes * j.ncols;
            const bool dev = is_device_pointer(j.src.col(j.first, j.n));
            // only the staging ring cuts columns: the other paths copy whole column groups, as round 4 did
            const bool whole = dev || mode_ != STAGED;
            if (col_bytes >= piece_bytes_ && !whole) {
                const size_t rows_per = piece_bytes_ / 8;
                for (uint32_t c = 0; c < j.ncols; c++)
                    for (size_t r = 0; r < j.n; r += rows_per)
                        pieces_.push_back({(uint32_t)t, c, c + 1, r, std::min(rows_per, j.n - r), r + rows_per >= j.n ? c + 1 : c, false});
            } else {
                // small columns travel together; device-resident tables go column group by column group
                // (round 6) a column can only travel narrow when it has a piece to itself: columns of solo_bytes_ and more do, also where
                // several would fit a slot -- a 2^20-row table (8 MB columns, two per 16 MB slot) sent 8 % fewer bytes instead of 40 %
                const bool solo = !whole && pack_ && solo_bytes_ && col_bytes >= solo_bytes_ && j.n >= 4096;
                const uint32_t per = whole ? j.chunk : solo ? 1u : (uint32_t)std::max<size_t>(1, piece_bytes_ / col_bytes);
                for (uint32_t c = 0; c < j.ncols; c += per) {
                    const uint32_t c1 = std::min(j.ncols, c + per);
                    pieces_.push_back({(uint32_t)t, c, c1, 0, j.n, c1, dev});
                }
            }
        }
    }
    const u64* src_of(const Piece& p, uint32_t c) const { const Job& j = jobs_[p.table]; return j.src.col(j.first + c, j.n) + p.row0; }
    u64* dst_of(const Piece& p) const { const Job& j = jobs_[p.table]; return j.dst + (size_t)p.c0 * j.n + p.row0; }
    size_t bytes_of(const Piece& p) const { return (size_t)(p.c1 - p.c0) * p.rows * 8; }
    // the source columns [c0, c1) of a piece are one run of memory (always true for a contiguous table)
    bool contiguous(const Piece& p) const {
        for (uint32_t c = p.c0 + 1; c < p.c1; c++) if (src_of(p, c) != src_of(p, c - 1) + p.rows) return false;
        return true;
    }
    void ensure_ring() {
        const size_t want = piece_bytes_ * slots_;
        if (ctx_->staging && ctx_->staging_bytes >= want) return;
        if (ctx_->staging) { (void)hipHostFree(ctx_->staging); ctx_->staging = nullptr; ctx_->staging_bytes = 0; }
        size_t got = 0;
        if (void* warm = take_warm_ring(ctx_->device, want, &got)) { ctx_->staging = warm; ctx_->staging_bytes = got; return; }   // pinned by ola_gpu_warmup
        HIP_CHECK(hipHostMalloc(&ctx_->staging, want, hipHostMallocDefault));
        ctx_->staging_bytes = want;
    }
    // device side of the narrow-column path: one half-size slot per host slot
    void ensure_device_ring() {
        const size_t want = piece_bytes_ / 2 * slots_;
        if (ctx_->staging_dev && ctx_->staging_dev_bytes >= want) return;
        if (ctx_->staging_dev) { (void)hipFree(ctx_->staging_dev); ctx_->staging_dev = nullptr; ctx_->staging_dev_bytes = 0; }
        HIP_CHECK(hipMalloc(&ctx_->staging_dev, want));
        ctx_->staging_dev_bytes = want;
    }
    // low halves of src[0, n) into dst while every word is below 2^32; false (dst garbage) at the first larger one
    static bool pack_low_halves(uint32_t* __restrict__ dst, const u64* __restrict__ src, size_t n) {
        const size_t block = 4096;
        for (size_t b = 0; b < n; b += block) {
            const size_t e = std::min(n, b + block);
            u64 hi = 0;
            for (size_t i = b; i < e; i++) { hi |= src[i]; dst[i] = (uint32_t)src[i]; }
            if (hi >> 32) return false;
        }
        return true;
    }
    // low bytes of src[0, n) into dst while every word is below 2^8
    static bool pack_low_bytes(uint8_t* __restrict__ dst, const u64* __restrict__ src, size_t n) {
        const size_t block = 4096;
        for (size_t b = 0; b < n; b += block) {
            const size_t e = std::min(n, b + block);
            u64 hi = 0;
            for (size_t i = b; i < e; i++) { hi |= src[i]; dst[i] = (uint8_t)src[i]; }
            if (hi >> 8) return false;
        }
        return true;
    }
    void fail(hipError_t e) {
        std::lock_guard<std::mutex> lk(mu_);
        if (!failed_) { failed_ = true; error_ = hipGetErrorString(e); }
        cv_.notify_all();
    }
    void publish(const Piece& p, size_t completed) {
        std::lock_guard<std::mutex> lk(mu_);
        completed_ = completed;
        if (p.cols_done > done_[p.table].load()) {
            done_[p.table].store(p.cols_done);
            if (first_ms_ == 0) first_ms_ = since_start();
        }
        cv_.notify_all();
    }
    double since_start() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start_).count(); }

    // STAGED: copier threads take pieces in order, fill the piece's slot (free once the piece `slots_` earlier has retired) and
    // send it; retire() waits for the copies in order.
    void copier() {
        (void)hipSetDevice(ctx_->device);
        for (;;) {
            const size_t i = next_.fetch_add(1);
            if (i >= pieces_.size() || cancel_.load()) return;
            const Piece& p = pieces_[i];
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return completed_ + slots_ > i || failed_ || cancel_.load(); });
                if (failed_ || cancel_.load()) { issued_[i].store(2); cv_.notify_all(); return; }
            }
            hipError_t e = hipSuccess;
            hipStream_t st = streams_[i % streams_.size()];
            if (p.device_src) {
                std::lock_guard<std::mutex> lk(issue_mu_);
                for (uint32_t c = p.c0; c < p.c1 && e == hipSuccess; c++) {
                    narrow_[p.table][c].store(0);
                    e = hipMemcpyAsync(dst_of(p) + (size_t)(c - p.c0) * p.rows, src_of(p, c), p.rows * 8, hipMemcpyDeviceToDevice, st);
                }
                if (e == hipSuccess) e = hipEventRecord(events_[i % slots_], st);
            } else {
                char* slot = (char*)ctx_->staging + (i % slots_) * piece_bytes_;
                const bool packable = pack_ && p.c1 - p.c0 == 1 && p.rows >= 4096;
                const bool bytes = packable && pack8_ && pack_low_bytes((uint8_t*)slot, src_of(p, p.c0), p.rows);
                const bool narrow = bytes || (packable && pack_low_halves((uint32_t*)slot, src_of(p, p.c0), p.rows));
                if (!narrow) {
                    for (uint32_t c = p.c0; c < p.c1; c++) {
                        memcpy(slot + (size_t)(c - p.c0) * p.rows * 8, src_of(p, c), p.rows * 8);
                        narrow_[p.table][c].store(0);
                    }
                }
                std::lock_guard<std::mutex> lk(issue_mu_);
                if (narrow) {
                    uint32_t* d32 = (uint32_t*)((char*)ctx_->staging_dev + (i % slots_) * (piece_bytes_ / 2));
                    const size_t wire = p.rows * (bytes ? 1 : 4);
                    e = hipMemcpyAsync(d32, slot, wire, hipMemcpyHostToDevice, st);
                    if (e == hipSuccess) {
                        if (bytes) hipLaunchKernelGGL(upload_widen8_kernel, dim3((unsigned)((p.rows + 2047) / 2048)), dim3(256), 0, st, (const uint8_t*)d32, dst_of(p), p.rows);
                        else hipLaunchKernelGGL(upload_widen_kernel, dim3((unsigned)((p.rows + 1023) / 1024)), dim3(256), 0, st, d32, dst_of(p), p.rows);
                        e = hipGetLastError();
                    }
                    link_bytes_.fetch_add(wire);
                } else {
                    e = hipMemcpyAsync(dst_of(p), slot, bytes_of(p), hipMemcpyHostToDevice, st);
                    link_bytes_.fetch_add(bytes_of(p));
                }
                if (e == hipSuccess) e = hipEventRecord(events_[i % slots_], st);
            }
            if (e != hipSuccess) fail(e);
            issued_[i].store(e == hipSuccess ? 1 : 2);
            { std::lock_guard<std::mutex> lk(mu_); cv_.notify_all(); }
            if (e != hipSuccess) return;
        }
    }
    void retire() {
        (void)hipSetDevice(ctx_->device);
        for (size_t i = 0; i < pieces_.size(); i++) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return issued_[i].load() != 0 || failed_ || cancel_.load(); });
                if (issued_[i].load() != 1) return;
            }
            const hipError_t e = hipEventSynchronize(events_[i % slots_]);
            if (e != hipSuccess) { fail(e); return; }
            publish(pieces_[i], i + 1);
        }
        total_ms_ = since_start();
    }
    // PAGEABLE: one thread, one copy per column group
    void run_direct() {
        (void)hipSetDevice(ctx_->device);
        for (size_t i = 0; i < pieces_.size() && !cancel_.load(); i++) {
            const Piece& p = pieces_[i];
            hipError_t e = hipSuccess;
            if (p.c1 - p.c0 > 1 && contiguous(p)) {
                e = hipMemcpyAsync(dst_of(p), src_of(p, p.c0), bytes_of(p), hipMemcpyDefault, streams_[0]);
            } else {
                for (uint32_t c = p.c0; c < p.c1 && e == hipSuccess; c++)
                    e = hipMemcpyAsync(dst_of(p) + (size_t)(c - p.c0) * p.rows, src_of(p, c), p.rows * 8, hipMemcpyDefault, streams_[0]);
            }
            // group boundary (or every piece from pageable memory, where the call itself is the copy): wait and publish
            const bool last_of_group = p.cols_done != p.c0 && (p.cols_done % jobs_[p.table].chunk == 0 || p.cols_done == jobs_[p.table].ncols);
            if (e == hipSuccess && (last_of_group || p.device_src)) e = hipStreamSynchronize(streams_[0]);
            if (e != hipSuccess) { fail(e); return; }
            if (last_of_group || p.device_src) publish(p, i + 1);
        }
        total_ms_ = since_start();
    }
    void join_all() {
        for (std::thread& t : copiers_) if (t.joinable()) t.join();
        if (th_.joinable()) th_.join();
    }

    DeviceCtx* ctx_;
    std::vector<Job> jobs_;
    std::vector<Piece> pieces_;
    std::vector<std::atomic<uint32_t>> done_;
    std::vector<std::unique_ptr<std::atomic<char>[]>> narrow_;   // per table and column: every piece of it travelled narrow
    std::unique_ptr<std::atomic<char>[]> issued_;     // per piece: 0 not yet, 1 sent, 2 given up
    std::vector<hipEvent_t> events_;                  // per slot
    std::vector<std::thread> copiers_;
    std::thread th_;
    std::mutex mu_, issue_mu_;
    std::condition_variable cv_;
    std::atomic<size_t> next_{0};
    std::atomic<size_t> link_bytes_{0};               // what actually crossed the link
    bool pack_ = true;
    bool pack8_ = true;                               // columns of byte-sized words travel as bytes (OLA_UPLOAD_PACK8)
    size_t solo_bytes_ = 512 << 10;                   // columns of this size and more get a piece of their own (0: never)
    size_t completed_ = 0;                            // pieces retired (guarded by mu_)
    std::atomic<bool> cancel_{false};
    Mode mode_ = STAGED, pieces_plan_ = STAGED;       // pieces_plan_: the mode plan() cut the pieces for
    size_t piece_bytes_ = 0, slots_ = 0, nstreams_ = 2, bytes_ = 0;
    std::vector<size_t> order_;
    unsigned nthreads_ = 1;
    std::chrono::steady_clock::time_point t_start_;
    double waited_ms_ = 0, first_ms_ = 0, total_ms_ = 0;
    bool failed_ = false;
    std::string error_;
    std::vector<hipStream_t> streams_;
    hipEvent_t start_ev_ = nullptr;
};

}  // namespace ola
