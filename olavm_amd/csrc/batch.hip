// PolynomialBatch on the device: coefficient form + coset LDE in commitment-leaf order + Poseidon Merkle heap.
//
// Replaces PolynomialBatch::{from_values, from_coeffs, get_lde_values} and MerkleTree::{get, prove}
//   plonky2/plonky2/src/fri/oracle.rs:45-139, plonky2/plonky2/src/hash/merkle_tree/mod.rs:268-308
// The reference materialises row-major leaves (transpose + reverse_index_bits_in_place, oracle.rs:84-85); here the
// LDE stays column-major in HBM and is produced directly in leaf order by the in-place DIF transform:
//   lde[col][c*n + r] = P_col(7 * g^bitrev(c) * w_n^bitrev_n(r))  ==  natural LDE row bitrev_N(c*n + r).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <functional>
#include <vector>

#include "device_ctx.h"
#include "gl.cuh"

// A batch may be one GPU's SHARE of a commitment (coset partition): leaf blocks (cosets)
// [coset_first, coset_first + 2^rate_bits) of a 2^full_rate_bits-coset LDE.  `rate_bits` and `cap_height` then describe the
// local tree (full values minus log2 of the number of shards), whose cap is this shard's contiguous slice of the full cap.
struct OlaBatch {
    uint32_t ncols = 0, log_n = 0, rate_bits = 0, cap_height = 0;
    uint32_t full_rate_bits = 0, coset_first = 0;
    bool is_shard() const { return rate_bits != full_rate_bits; }
    // lean: the LDE is not kept.  It was produced one coset at a time while the leaves were hashed, and whoever needs values
    // again (quotient evaluation, opened rows) re-derives the coset from `coeffs` (batch_lde_slice); coefficients and digests stay.
    bool lean = false;
    ola::u64* coeffs = nullptr;  // [ncols][n], natural coefficient order
    ola::u64* lde = nullptr;     // [ncols][N], leaf order (nullptr when lean)
    ola::u64* heap = nullptr;    // 2N digests of 4 u64, heap[N + j] = leaf j, root at 1
    size_t n() const { return (size_t)1 << log_n; }
    size_t num_leaves() const { return (size_t)1 << (log_n + rate_bits); }
};

namespace ola {

__global__ __launch_bounds__(256) void bitrev_rows_kernel(const u64* __restrict__ in, u64* __restrict__ out, int bits) {
    const size_t N = (size_t)1 << bits;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const size_t col = blockIdx.y;
    out[col * N + i] = in[col * N + bitrev32((u32)i, bits)];
}
// Tiled form for bits >= 2T: the index splits as [hi : T | mid | lo : T] and reverses to [rev lo | rev mid | rev hi].  One
// workgroup moves the 2^T x 2^T tile of one `mid`: rows of 2^T consecutive elements in (fixed hi, running lo), transposed in
// LDS, rows of 2^T consecutive elements out (fixed rev lo, running rev hi) -- both sides touch whole 2^T * 8 byte segments
// instead of the 8-byte gathers of the plain kernel.
template <int T>
__global__ __launch_bounds__(256) void bitrev_rows_tiled_kernel(const u64* __restrict__ in, u64* __restrict__ out, int bits) {
    __shared__ u64 tile[1 << T][(1 << T) + 1];
    const size_t N = (size_t)1 << bits;
    const int midbits = bits - 2 * T;
    const u32 mid = blockIdx.x;
    const u32 rmid = midbits ? bitrev32(mid, midbits) : 0;
    const u64* src = in + (size_t)blockIdx.y * N + ((size_t)mid << T);
    u64* dst = out + (size_t)blockIdx.y * N + ((size_t)rmid << T);
#pragma unroll
    for (int e = threadIdx.x; e < (1 << (2 * T)); e += 256) {
        const u32 hi = e >> T, lo = e & ((1 << T) - 1);
        tile[hi][lo] = src[((size_t)hi << (bits - T)) + lo];
    }
    __syncthreads();
#pragma unroll
    for (int e = threadIdx.x; e < (1 << (2 * T)); e += 256) {
        const u32 a = e >> T, b = e & ((1 << T) - 1);          // a = rev(lo), b = rev(hi)
        dst[((size_t)a << (bits - T)) + b] = tile[bitrev32(b, T)][bitrev32(a, T)];
    }
}
void launch_bitrev_rows(DeviceCtx* ctx, const u64* in, u64* out, int bits, size_t cols) {
    const size_t N = (size_t)1 << bits;
    constexpr int T = 5;
    if (bits >= 2 * T && bits - 2 * T <= 30) {
        hipLaunchKernelGGL(bitrev_rows_tiled_kernel<T>, dim3(1u << (bits - 2 * T), (unsigned)cols), dim3(256), 0, ctx->stream, in, out, bits);
        return;
    }
    hipLaunchKernelGGL(bitrev_rows_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)cols), dim3(256), 0, ctx->stream,
                       in, out, bits);
}

// data[col][k] *= s^k  (two-level table of s)
__global__ __launch_bounds__(256) void scale_powers_kernel(u64* __restrict__ data, size_t n, const u64* __restrict__ lo,
                                                           const u64* __restrict__ hi, int h) {
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const size_t col = blockIdx.y;
    const u64 s = gl_mul(lo[k & (((size_t)1 << h) - 1)], hi[k >> h]);
    data[col * n + k] = gl_mul(data[col * n + k], s);
}

// values on shift*<w_n> (natural order) -> coefficients (cfft/serial.rs:64-78)
void ntt_coset_interpolate(NttTables& t, const u64* values, u64* coeffs, u64* scratch, int L, size_t cols, u64 shift) {
    const size_t n = (size_t)1 << L;
    ntt_interpolate(t, values, coeffs, scratch, L, cols);
    TwoLevel sc = get_shift(t, L, gl_inv(shift));
    hipLaunchKernelGGL(scale_powers_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)cols), dim3(256), 0,
                       t.ctx->stream, coeffs, n, sc.lo, sc.hi, sc.h);
}

// rows_out[q][c] = lde[c][idx[q]]
__global__ void gather_rows_kernel(const u64* __restrict__ lde, size_t col_stride, int ncols,
                                   const unsigned long long* __restrict__ idx, u64* __restrict__ rows_out) {
    const int q = blockIdx.x;
    for (int c = threadIdx.x; c < ncols; c += blockDim.x) rows_out[(size_t)q * ncols + c] = lde[(size_t)c * col_stride + idx[q]];
}
// paths_out[q][l] = sibling digest at level l (leaf level first), depth = log2(N) - cap_height
__global__ void gather_paths_kernel(const u64* __restrict__ heap, size_t N, int depth,
                                    const unsigned long long* __restrict__ idx, u64* __restrict__ paths_out) {
    const int q = blockIdx.x;
    const int l = threadIdx.x >> 2, w = threadIdx.x & 3;
    if (l >= depth) return;
    const size_t node = ((N + idx[q]) >> l) ^ 1;
    paths_out[((size_t)q * depth + l) * 4 + w] = heap[node * 4 + w];
}

__global__ __launch_bounds__(256) void canonicalize_kernel(u64* __restrict__ d, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] = gl_canon(d[i]);
}
void canonicalize(DeviceCtx* ctx, u64* d, size_t n) {
    hipLaunchKernelGGL(canonicalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d, n);
}

void batch_destroy(DeviceCtx* ctx, OlaBatch* b) {
    if (!b) return;
    ctx->free(b->coeffs);
    ctx->free(b->lde);
    ctx->free(b->heap);
    delete b;
}

// Columns of a device-resident table may still be arriving (prove_with_traces uploads the traces from a helper thread):
// the interpolation then runs in groups of `chunk_cols` columns and calls before_chunk(c0, c1) -- wait for the upload,
// canonicalise -- ahead of each group.
struct ColumnFeed {
    uint32_t chunk_cols = 0;
    std::function<void(uint32_t, uint32_t)> before_chunk;
};

OlaBatch* batch_commit(DeviceCtx* ctx, NttTables& t, const uint64_t* const* cols_host, const u64* cols_dev, uint32_t ncols,
                       uint32_t log_n, uint32_t rate_bits, uint32_t cap_height, bool from_values, uint32_t shard_rank = 0,
                       uint32_t shard_log_world = 0, const ColumnFeed* feed = nullptr, bool lean = false) {
    if (shard_log_world > rate_bits || shard_log_world > cap_height || shard_rank >= (1u << shard_log_world))
        throw OlaError(-1, "shard count must divide both the number of cosets and the cap");
    OlaBatch* b = new OlaBatch();
    b->ncols = ncols; b->log_n = log_n; b->full_rate_bits = rate_bits;
    const uint32_t full_rate_bits = rate_bits;
    rate_bits -= shard_log_world;          // from here on: the local tree
    cap_height -= shard_log_world;
    b->rate_bits = rate_bits; b->cap_height = cap_height;
    b->coset_first = shard_rank << rate_bits;
    const size_t n = b->n(), N = b->num_leaves();
    u64* tmp = nullptr;
    bool lde_done = false;
    if (lean && rate_bits > 0) {
        // Memory-lean commitment (fri/oracle.rs:66-99 with the LDE streamed): values -> coefficients, then for every coset of the
        // rank: extend (one n-point coset transform per column), hash its n leaves into the digest heap, drop the values.
        // Peak: coefficients + one coset + the heap, instead of coefficients + 2^rate_bits cosets + the heap.
        b->lean = true;
        u64* stage = nullptr;
        try {
            b->coeffs = (u64*)ctx->alloc((size_t)ncols * n * 8);
            b->heap = (u64*)ctx->alloc(2 * N * 32);
            tmp = (u64*)ctx->alloc((size_t)ncols * n * 8);      // transform scratch, then the coset being hashed
            if (from_values) {
                const u64* vals = cols_dev;
                if (cols_host) {
                    stage = (u64*)ctx->alloc((size_t)ncols * n * 8);
                    for (uint32_t c = 0; c < ncols; c++)
                        HIP_CHECK(hipMemcpyAsync(stage + (size_t)c * n, cols_host[c], n * 8, hipMemcpyHostToDevice, ctx->stream));
                    vals = stage;
                }
                if (feed && feed->chunk_cols && !cols_host) {
                    for (uint32_t c0 = 0; c0 < ncols; c0 += feed->chunk_cols) {
                        const uint32_t c1 = std::min(ncols, c0 + feed->chunk_cols);
                        feed->before_chunk(c0, c1);
                        ntt_interpolate(t, vals + (size_t)c0 * n, b->coeffs + (size_t)c0 * n, tmp, log_n, c1 - c0);
                    }
                } else {
                    ntt_interpolate(t, vals, b->coeffs, tmp, log_n, ncols);
                }
            } else {
                if (cols_host) {
                    for (uint32_t c = 0; c < ncols; c++)
                        HIP_CHECK(hipMemcpyAsync(b->coeffs + (size_t)c * n, cols_host[c], n * 8, hipMemcpyHostToDevice, ctx->stream));
                } else {
                    HIP_CHECK(hipMemcpyAsync(b->coeffs, cols_dev, (size_t)ncols * n * 8, hipMemcpyDeviceToDevice, ctx->stream));
                }
                canonicalize(ctx, b->coeffs, (size_t)ncols * n);
            }
            {
                WorkScope ws(ctx, (int)std::min(full_rate_bits, b->cap_height + shard_log_world));
                for (size_t c = 0; c < ((size_t)1 << rate_bits); c++) {
                    ntt_lde_leaf_order(t, b->coeffs, tmp, log_n, full_rate_bits, ncols, b->coset_first + c, 1);
                    launch_leaf_hash_colmajor(ctx, tmp, n, (int)ncols, n, b->heap + 4 * N + 4 * c * n);
                }
                launch_merkle_build(ctx, b->heap, N, cap_height);
            }
            HIP_CHECK(hipStreamSynchronize(ctx->stream));
            ctx->free(tmp);
            if (stage) ctx->free(stage);
        } catch (...) {
            if (tmp) ctx->free(tmp);
            if (stage) ctx->free(stage);
            batch_destroy(ctx, b);
            throw;
        }
        return b;
    }
    try {
        b->coeffs = (u64*)ctx->alloc((size_t)ncols * n * 8);
        b->lde = (u64*)ctx->alloc((size_t)ncols * N * 8);
        b->heap = (u64*)ctx->alloc(2 * N * 32);
        // staging inside the (not yet written) LDE buffer: values at [0, ncols*n), transform scratch after it
        u64* stage = b->lde;
        u64* scratch = b->lde + (size_t)ncols * n;
        if (rate_bits == 0) { tmp = (u64*)ctx->alloc((size_t)ncols * n * 8); scratch = tmp; }
        if (from_values) {
            const u64* vals = cols_dev;
            if (cols_host) {
                for (uint32_t c = 0; c < ncols; c++)
                    HIP_CHECK(hipMemcpyAsync(stage + (size_t)c * n, cols_host[c], n * 8, hipMemcpyHostToDevice, ctx->stream));
                vals = stage;
            }
            if (feed && feed->chunk_cols && !cols_host) {
                // interpolate AND extend each group as it arrives, so that only the leaf hashing waits for the last column
                // (the transform scratch cannot live inside the LDE buffer here: earlier groups already wrote theirs)
                if (!tmp) tmp = (u64*)ctx->alloc((size_t)feed->chunk_cols * n * 8);
                PhaseTimer tp(ctx, "      IFFT + FFT + blinding (column groups as the upload delivers them)");
                for (uint32_t c0 = 0; c0 < ncols; c0 += feed->chunk_cols) {
                    const uint32_t c1 = std::min(ncols, c0 + feed->chunk_cols);
                    feed->before_chunk(c0, c1);
                    ntt_interpolate(t, vals + (size_t)c0 * n, b->coeffs + (size_t)c0 * n, tmp, log_n, c1 - c0);
                    WorkScope ws(ctx, (int)std::min(full_rate_bits, b->cap_height + shard_log_world));
                    ntt_lde_leaf_order(t, b->coeffs + (size_t)c0 * n, b->lde + (size_t)c0 * N, log_n, full_rate_bits, c1 - c0, b->coset_first,
                                       (size_t)1 << rate_bits);
                }
                lde_done = true;
            } else {
                PhaseTimer tp(ctx, "      IFFT");
                ntt_interpolate(t, vals, b->coeffs, scratch, log_n, ncols);
            }
        } else {
            if (cols_host) {
                for (uint32_t c = 0; c < ncols; c++)
                    HIP_CHECK(hipMemcpyAsync(b->coeffs + (size_t)c * n, cols_host[c], n * 8, hipMemcpyHostToDevice, ctx->stream));
            } else {
                HIP_CHECK(hipMemcpyAsync(b->coeffs, cols_dev, (size_t)ncols * n * 8, hipMemcpyDeviceToDevice, ctx->stream));
            }
            canonicalize(ctx, b->coeffs, (size_t)ncols * n);
        }
        // (OLA_TIMING scopes carry the reference's `timed!` names, fri/oracle.rs:56-90; "transpose LDEs" has no counterpart:
        // the LDE is produced in leaf order)
        WorkScope ws(ctx, (int)std::min(full_rate_bits, b->cap_height + shard_log_world));   // from here on the rank's cosets only
        if (!lde_done) { PhaseTimer tp(ctx, "      FFT + blinding"); ntt_lde_leaf_order(t, b->coeffs, b->lde, log_n, full_rate_bits, ncols, b->coset_first, (size_t)1 << rate_bits); }
        {
            PhaseTimer tp(ctx, "      build Merkle tree");
            launch_leaf_hash_colmajor(ctx, b->lde, N, (int)ncols, N, b->heap + 4 * N);
            launch_merkle_build(ctx, b->heap, N, cap_height);
        }
        // the caller's host columns must not be read after return; device inputs need no wait here (whoever reads the cap waits)
        if (cols_host) HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (tmp) ctx->free(tmp);
    } catch (...) {
        if (tmp) ctx->free(tmp);
        batch_destroy(ctx, b);
        throw;
    }
    return b;
}

void batch_read_cap(DeviceCtx* ctx, const OlaBatch& b, u64* cap_out) {
    const size_t len_cap = (size_t)1 << b.cap_height;
    // merkle_tree/mod.rs:218-226: the cap is the level with 2^cap_height nodes (the leaf digests if the tree is all cap)
    // through the context's pinned scratch slot: a read-back into pageable memory is staged by the runtime and costs 28 us
    // instead of 16 (tools/ubench/roundtrip.hip); every commitment of every table waits for this one
    char* slot = len_cap * 32 <= DeviceCtx::kPinnedScratch ? ctx->pinned_base() : nullptr;
    HIP_CHECK(hipMemcpyAsync(slot ? (void*)slot : (void*)cap_out, b.heap + 4 * len_cap, len_cap * 32, hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (slot) memcpy(cap_out, slot, len_cap * 32);
}

// Coset `coset` (local numbering) of the batch's LDE in leaf order, [ncols][n], re-derived from the coefficients (lean batches)
void batch_lde_slice(DeviceCtx* ctx, NttTables& t, const OlaBatch& b, size_t coset, u64* out) {
    (void)ctx;
    ntt_lde_leaf_order(t, b.coeffs, out, b.log_n, b.full_rate_bits, b.ncols, b.coset_first + coset, 1);
}

// rows_out: nq x ncols, paths_out: nq x depth x 4 (either may be null)
void batch_get_leaves(DeviceCtx* ctx, const OlaBatch& b, const size_t* idx, size_t nq, u64* rows_out, u64* paths_out, NttTables* tables = nullptr) {
    const size_t N = b.num_leaves();
    const int depth = (int)(b.log_n + b.rate_bits - b.cap_height);
    if (b.lean && rows_out) {
        // the opened rows of a lean batch: one coset at a time, only the cosets a query falls into
        if (!tables) throw OlaError(-1, "rows of a memory-lean commitment need the transform tables");
        const size_t n = b.n();
        u64* slice = (u64*)ctx->alloc((size_t)b.ncols * n * 8);
        unsigned long long* d_idx = (unsigned long long*)ctx->alloc(nq * 8);
        u64* d_rows = (u64*)ctx->alloc(nq * b.ncols * 8);
        try {
            for (size_t c = 0; c < ((size_t)1 << b.rate_bits); c++) {
                std::vector<unsigned long long> loc;
                std::vector<size_t> pos;
                for (size_t q = 0; q < nq; q++)
                    if ((idx[q] >> b.log_n) == c) { loc.push_back(idx[q] & (n - 1)); pos.push_back(q); }
                if (loc.empty()) continue;
                batch_lde_slice(ctx, *tables, b, c, slice);
                HIP_CHECK(hipMemcpyAsync(d_idx, loc.data(), loc.size() * 8, hipMemcpyHostToDevice, ctx->stream));
                hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)loc.size()), dim3(64), 0, ctx->stream, slice, n, (int)b.ncols, d_idx, d_rows);
                std::vector<u64> h(loc.size() * b.ncols);
                HIP_CHECK(hipMemcpyAsync(h.data(), d_rows, h.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
                HIP_CHECK(hipStreamSynchronize(ctx->stream));
                for (size_t k = 0; k < pos.size(); k++) std::copy(h.begin() + k * b.ncols, h.begin() + (k + 1) * b.ncols, rows_out + pos[k] * b.ncols);
            }
        } catch (...) { ctx->free(slice); ctx->free(d_idx); ctx->free(d_rows); throw; }
        ctx->free(slice); ctx->free(d_idx); ctx->free(d_rows);
        rows_out = nullptr;          // the paths below come from the resident heap
        if (!paths_out) return;
    }
    std::vector<unsigned long long> h_idx(idx, idx + nq);
    unsigned long long* d_idx = (unsigned long long*)ctx->alloc(nq * 8);
    u64* d_rows = (u64*)ctx->alloc(nq * b.ncols * 8);
    u64* d_paths = (u64*)ctx->alloc(nq * (size_t)(depth > 0 ? depth : 1) * 32);
    try {
        HIP_CHECK(hipMemcpyAsync(d_idx, h_idx.data(), nq * 8, hipMemcpyHostToDevice, ctx->stream));
        if (rows_out) {
            hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)nq), dim3(64), 0, ctx->stream, b.lde, N, (int)b.ncols, d_idx, d_rows);
            HIP_CHECK(hipMemcpyAsync(rows_out, d_rows, nq * b.ncols * 8, hipMemcpyDeviceToHost, ctx->stream));
        }
        if (paths_out && depth > 0) {
            hipLaunchKernelGGL(gather_paths_kernel, dim3((unsigned)nq), dim3(((depth * 4 + 63) / 64) * 64), 0, ctx->stream,
                               b.heap, N, depth, d_idx, d_paths);
            HIP_CHECK(hipMemcpyAsync(paths_out, d_paths, nq * (size_t)depth * 32, hipMemcpyDeviceToHost, ctx->stream));
        }
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    } catch (...) { ctx->free(d_idx); ctx->free(d_rows); ctx->free(d_paths); throw; }
    ctx->free(d_idx); ctx->free(d_rows); ctx->free(d_paths);
}
void batch_get_leaf(DeviceCtx* ctx, const OlaBatch& b, size_t leaf, u64* row_out, u64* sib_out, NttTables* tables = nullptr) {
    batch_get_leaves(ctx, b, &leaf, 1, row_out, sib_out, tables);
}

}  // namespace ola
