// Poseidon-Goldilocks sponge + Merkle tree kernels for gfx950 (and the Blake3 leaves / nodes of Blake3GoldilocksConfig, blake3.cuh).
//
// Replaces (reference, relative to plonky2/plonky2/src/hash):
//   hashing.rs:84-107            hash_n_to_m_no_pad  -- overwrite-mode sponge, rate 8, 4-element digest
//   hashing.rs:66-74             compress / two_to_one = permute(l || r || 0)[0..4]
//   merkle_tree/mod.rs:180-226   MerkleTree::new_v2: every leaf hashed with hash_no_pad; heap-ordered nodes;
//                                cap = level with 2^cap_height nodes
// Layout in HBM: `heap` holds 2N digests of 4 u64: heap[N + j] = digest of leaf j, heap[i] = H(heap[2i], heap[2i+1]),
// root at 1 (heap[0] unused).  Sibling paths for MerkleTree::prove (mod.rs:273-308) are read straight off the heap, so
// the reference's per-cap-subtree `digests` relayout (mod.rs:228-259) is never materialised.
// Leaves are read column-major ([col][leaf]) so consecutive lanes read consecutive addresses.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "device_ctx.h"
#include "gl.cuh"
#include "poseidon.cuh"
#include "blake3.cuh"

namespace ola {

// batches up to this many states / nodes / leaves use the quad-cooperative (latency-oriented) kernels
static const size_t QUAD_MAX = 8192;

// leaf j = (cols[0][j], cols[1][j], ...), column c at base + c*col_stride
__global__ __launch_bounds__(256) void leaf_hash_colmajor_kernel(const u64* __restrict__ base, size_t col_stride,
                                                                  int ncols, size_t num_leaves, u64* __restrict__ out) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= num_leaves) return;
    u64 s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = 0;
    for (int c0 = 0; c0 < ncols; c0 += 8) {
        const int len = min(8, ncols - c0);
#pragma unroll
        for (int i = 0; i < 8; i++)
            if (i < len) s[i] = gl_canon(base[(size_t)(c0 + i) * col_stride + j]);
        poseidon_permute(s);
    }
    ulonglong2* o = reinterpret_cast<ulonglong2*>(out + j * 4);
    o[0] = make_ulonglong2(s[0], s[1]);
    o[1] = make_ulonglong2(s[2], s[3]);
}

// leaf j = row j of a row-major matrix (rows of row_len elements)
__global__ __launch_bounds__(256) void leaf_hash_rowmajor_kernel(const u64* __restrict__ rows, size_t row_len,
                                                                  size_t num_leaves, u64* __restrict__ out) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= num_leaves) return;
    const u64* r = rows + j * row_len;
    u64 s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = 0;
    for (size_t c0 = 0; c0 < row_len; c0 += 8) {
        const int len = (int)min((size_t)8, row_len - c0);
#pragma unroll
        for (int i = 0; i < 8; i++)
            if (i < len) s[i] = gl_canon(r[c0 + i]);
        poseidon_permute(s);
    }
    ulonglong2* o = reinterpret_cast<ulonglong2*>(out + j * 4);
    o[0] = make_ulonglong2(s[0], s[1]);
    o[1] = make_ulonglong2(s[2], s[3]);
}

// FRI commit-phase leaves (fri/prover.rs:90-96): leaf j = flatten(16 consecutive bit-reversed extension values) =
// (a[16j], b[16j], a[16j+1], b[16j+1], ...) with the extension field stored as two planes.
__global__ __launch_bounds__(256) void leaf_hash_ext_kernel(const u64* __restrict__ plane_a,
                                                            const u64* __restrict__ plane_b, int arity,
                                                            size_t num_leaves, u64* __restrict__ out) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= num_leaves) return;
    u64 s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = 0;
    for (int k0 = 0; k0 < arity; k0 += 4) {  // 4 extension elements = 8 sponge lanes per permutation
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (k0 + k < arity) {
                s[2 * k] = plane_a[j * arity + k0 + k];
                s[2 * k + 1] = plane_b[j * arity + k0 + k];
            }
        }
        poseidon_permute(s);
    }
    ulonglong2* o = reinterpret_cast<ulonglong2*>(out + j * 4);
    o[0] = make_ulonglong2(s[0], s[1]);
    o[1] = make_ulonglong2(s[2], s[3]);
}

// ---- quad-cooperative variants (4 threads per leaf / node, poseidon.cuh): same digests, used when the batch is too small
// to fill the chip, where the latency of a permutation is what matters.  Lane q of a quad owns sponge lanes 3q..3q+2.
#define QUAD_PROLOGUE(count)                                            \
    const size_t g_ = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   \
    const size_t j_ = g_ >> 2;                                         \
    const int q = (int)(g_ & 3);                                       \
    const bool live = j_ < (count);                                    \
    const size_t j = live ? j_ : (count) - 1; /* surplus lanes redo the last item so that quads stay converged */
__device__ __forceinline__ void quad_store_digest(u64* __restrict__ out4, const u64 (&x)[3], int q, bool live) {
    if (!live) return;
    if (q == 0) { out4[0] = x[0]; out4[1] = x[1]; out4[2] = x[2]; }
    if (q == 1) out4[3] = x[0];
}

__global__ __launch_bounds__(256) void leaf_hash_colmajor_quad_kernel(const u64* __restrict__ base, size_t col_stride, int ncols,
                                                                       size_t num_leaves, u64* __restrict__ out) {
    QUAD_PROLOGUE(num_leaves)
    u64 x[3] = {0, 0, 0};
    for (int c0 = 0; c0 < ncols; c0 += 8) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int e = 3 * q + k;   // sponge lane
            if (e < 8 && c0 + e < ncols) x[k] = gl_canon(base[(size_t)(c0 + e) * col_stride + j]);
        }
        poseidon_permute_quad(x, q);
    }
    quad_store_digest(out + j * 4, x, q, live);
}

__global__ __launch_bounds__(256) void leaf_hash_rowmajor_quad_kernel(const u64* __restrict__ rows, size_t row_len, size_t num_leaves,
                                                                       u64* __restrict__ out) {
    QUAD_PROLOGUE(num_leaves)
    const u64* r = rows + j * row_len;
    u64 x[3] = {0, 0, 0};
    for (size_t c0 = 0; c0 < row_len; c0 += 8) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const size_t e = 3 * q + k;
            if (e < 8 && c0 + e < row_len) x[k] = gl_canon(r[c0 + e]);
        }
        poseidon_permute_quad(x, q);
    }
    quad_store_digest(out + j * 4, x, q, live);
}

__global__ __launch_bounds__(256) void leaf_hash_ext_quad_kernel(const u64* __restrict__ plane_a, const u64* __restrict__ plane_b, int arity,
                                                                  size_t num_leaves, u64* __restrict__ out) {
    QUAD_PROLOGUE(num_leaves)
    u64 x[3] = {0, 0, 0};
    for (int k0 = 0; k0 < arity; k0 += 4) {   // sponge lane e = 2*kk (+1): extension element k0 + e/2, plane e & 1
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int e = 3 * q + k;
            if (e < 8 && k0 + e / 2 < arity) x[k] = ((e & 1) ? plane_b : plane_a)[j * arity + k0 + e / 2];
        }
        poseidon_permute_quad(x, q);
    }
    quad_store_digest(out + j * 4, x, q, live);
}

// node i = H(heap[2i] || heap[2i+1]) for i in [first, first + count)
__device__ __forceinline__ void quad_hash_node(u64* __restrict__ heap, size_t i, int q, bool live) {
    u64 x[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int e = 3 * q + k;
        x[k] = e < 8 ? heap[8 * i + e] : 0;   // the two child digests are adjacent
    }
    poseidon_permute_quad(x, q);
    quad_store_digest(heap + 4 * i, x, q, live);
}
__global__ __launch_bounds__(256) void merkle_level_quad_kernel(u64* __restrict__ heap, size_t first, size_t count) {
    QUAD_PROLOGUE(count)
    quad_hash_node(heap, first + j, q, live);
}

// parents [first, first+count) of a heap-ordered digest array
__global__ __launch_bounds__(256) void merkle_level_kernel(u64* __restrict__ heap, size_t first, size_t count) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const size_t i = first + t;
    const ulonglong2* ch = reinterpret_cast<const ulonglong2*>(heap + 8 * i);  // children 2i, 2i+1 are adjacent
    const ulonglong2 c0 = ch[0], c1 = ch[1], c2 = ch[2], c3 = ch[3];
    u64 s[12] = {c0.x, c0.y, c1.x, c1.y, c2.x, c2.y, c3.x, c3.y, 0, 0, 0, 0};
    poseidon_permute(s);
    ulonglong2* o = reinterpret_cast<ulonglong2*>(heap + 4 * i);
    o[0] = make_ulonglong2(s[0], s[1]);
    o[1] = make_ulonglong2(s[2], s[3]);
}

// The top of a tree in one launch: levels of `first` nodes and fewer (4 * first <= blockDim.x), down to the level of `last`
// nodes, one workgroup of quads, a barrier between levels.
__global__ __launch_bounds__(1024) void merkle_top_kernel(u64* __restrict__ heap, size_t first, size_t last) {
    const size_t t = threadIdx.x >> 2;
    const int q = threadIdx.x & 3;
    const size_t wave_first = (threadIdx.x & ~63u) >> 2;   // first node index of this wavefront's 16 quads
    for (size_t level = first; level >= last && level >= 1; level >>= 1) {
        if (wave_first < level) {   // wavefronts without a node of this level only wait at the barrier
            const bool live = t < level;
            quad_hash_node(heap, level + (live ? t : level - 1), q, live);
        }
        __threadfence_block();
        __syncthreads();
    }
}

// Rows of the Poseidon STARK table (builtins/poseidon/columns.rs; layout of generation/poseidon.rs:5-80) from the
// permutation inputs: the reference's executor records the S-box inputs of every round (core/src/util/poseidon_utils.rs)
// and generate_poseidon_trace lays them out; here one thread recomputes them for its row.  Columns: 4 lookup filters,
// input[12], output[12], S-box inputs of full rounds 1..3 [3x12], of the 22 partial rounds (lane 0) [22], of full rounds
// 26..29 [4x12].  The S-box argument of a round does not depend on how the linear layers are factored, so the dense
// lane-0 form of poseidon.cuh yields exactly the values the reference's sparse form records.
__global__ __launch_bounds__(256) void poseidon_trace_kernel(const u64* __restrict__ inputs, const u64* __restrict__ filters, size_t n,
                                                             u64* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    auto col = [&](int c) -> u64& { return out[(size_t)c * n + i]; };
    for (int f = 0; f < 4; f++) col(f) = filters ? gl_canon(filters[(size_t)f * n + i]) : 0;
    u64 s[12];
    for (int k = 0; k < 12; k++) { s[k] = gl_canon(inputs[(size_t)k * n + i]); col(4 + k) = s[k]; }
    for (int r = 0; r < 4; r++) {
        for (int k = 0; k < 12; k++) {
            s[k] = gl_add(gl_canon(s[k]), c_rc[r * 12 + k]);
            if (r > 0) col(28 + (r - 1) * 12 + k) = s[k];
            s[k] = sbox7_weak(s[k]);
        }
        mds_weak(s);
    }
    for (int r = 0; r < 22; r++) {
        const u64 a = gl_add(gl_canon(s[0]), c_lane0[r]);
        col(64 + r) = a;
        s[0] = sbox7_weak(a);
        mds_weak(s);
    }
    for (int r = 0; r < 4; r++) {
        for (int k = 0; k < 12; k++) {
            s[k] = gl_add(gl_canon(s[k]), r == 0 ? c_round26[k] : c_rc[(26 + r) * 12 + k]);
            col(86 + r * 12 + k) = s[k];
            s[k] = sbox7_weak(s[k]);
        }
        mds_weak(s);
    }
    for (int k = 0; k < 12; k++) col(16 + k) = gl_canon(s[k]);
}
void launch_poseidon_trace(DeviceCtx* ctx, const u64* inputs, const u64* filters, size_t n, u64* out) {
    if (n) hipLaunchKernelGGL(poseidon_trace_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, inputs, filters, n, out);
}

// quad-cooperative permutation of whole states (4 threads per state); used for small n
__global__ __launch_bounds__(256) void poseidon_states_quad_kernel(u64* __restrict__ states, size_t n) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t t = g >> 2;
    const int q = (int)(g & 3);
    const size_t tt = t < n ? t : n - 1;   // keep whole quads converged; the surplus lanes redo the last state
    u64 x[3];
#pragma unroll
    for (int k = 0; k < 3; k++) x[k] = gl_canon(states[tt * 12 + 3 * q + k]);
    poseidon_permute_quad(x, q);
    if (t < n) {
#pragma unroll
        for (int k = 0; k < 3; k++) states[t * 12 + 3 * q + k] = x[k];
    }
}

__global__ __launch_bounds__(256) void poseidon_states_kernel(u64* __restrict__ states, size_t n) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    u64 s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl_canon(states[t * 12 + i]);
    poseidon_permute(s);
#pragma unroll
    for (int i = 0; i < 12; i++) states[t * 12 + i] = s[i];
}

// FRI proof of work (fri/prover.rs:126-148): each thread tries nonce = start + global id; the minimum satisfying
// nonce of the batch is kept with an atomicMin.
__global__ __launch_bounds__(256) void pow_kernel(u64 h0, u64 h1, u64 h2, u64 h3, u64 start, u32 bits,
                                                  unsigned long long* __restrict__ best) {
    const u64 nonce = start + (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 s[12] = {h0, h1, h2, h3, nonce, 0, 0, 0, 0, 0, 0, 0};
    poseidon_permute(s);
    if ((s[0] >> (64 - bits)) == 0) atomicMin(best, (unsigned long long)nonce);
}

// ---- Blake3GoldilocksConfig (blake3.cuh): the same three leaf shapes and the same heap, one thread per leaf / node ----
// MULTI = leaves of more than one 1024-byte chunk (more than 128 elements: the 134-column Poseidon table); the single-chunk
// kernels need no stack of chaining values and therefore no scratch memory.
template <bool MULTI, class WordFn>
__device__ __forceinline__ void b3_leaf(WordFn word, u32 nwords, u32 (&d)[8]) {
    if (MULTI) b3_hash_words(word, nwords, d);
    else b3_chunk_cv(word, 0, nwords, 0, true, d);
}
template <bool MULTI>
__global__ __launch_bounds__(256) void leaf_b3_colmajor_kernel(const u64* __restrict__ base, size_t col_stride, int ncols,
                                                               size_t num_leaves, u64* __restrict__ out) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= num_leaves) return;
    u32 d[8];
    b3_leaf<MULTI>([&](u32 i) { return gl_canon(base[(size_t)i * col_stride + j]); }, (u32)ncols, d);
    b3_store_digest(out + j * 4, d);
}
template <bool MULTI>
__global__ __launch_bounds__(256) void leaf_b3_rowmajor_kernel(const u64* __restrict__ rows, size_t row_len, size_t num_leaves,
                                                               u64* __restrict__ out) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= num_leaves) return;
    const u64* r = rows + j * row_len;
    u32 d[8];
    b3_leaf<MULTI>([&](u32 i) { return gl_canon(r[i]); }, (u32)row_len, d);
    b3_store_digest(out + j * 4, d);
}
template <bool MULTI>
__global__ __launch_bounds__(256) void leaf_b3_ext_kernel(const u64* __restrict__ plane_a, const u64* __restrict__ plane_b, int arity,
                                                          size_t num_leaves, u64* __restrict__ out) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= num_leaves) return;
    u32 d[8];
    b3_leaf<MULTI>([&](u32 i) { return gl_canon(((i & 1) ? plane_b : plane_a)[j * arity + (i >> 1)]); }, (u32)(2 * arity), d);
    b3_store_digest(out + j * 4, d);
}
// The same leaves for arity 16, through LDS (round 6).  A thread of leaf_b3_ext_kernel reads 16 + 16 words 128 bytes apart from its
// neighbour's: a wave touches 64 lines per load and the lines are gone from the vector cache before their other words are asked
// for -- 17.1 GB fetched per 2^22-row proof for 1.2 GB of values (profiles/r05_proof_pmc_blake3.txt).  Here a workgroup copies
// its 128 leaves x 16 values of both planes with 16-byte loads on consecutive addresses and every thread hashes its leaf out of
// LDS (rows padded to 17 words: conflict-free for 8-byte reads).
#define B3X_LEAVES 128
__global__ __launch_bounds__(B3X_LEAVES) void leaf_b3_ext16_kernel(const u64* __restrict__ plane_a, const u64* __restrict__ plane_b, size_t num_leaves,
                                                                   u64* __restrict__ out) {
    __shared__ u64 tile[2][B3X_LEAVES * 17];
    const size_t j0 = (size_t)blockIdx.x * B3X_LEAVES;
    const size_t base = j0 * 16;
    const size_t avail = (num_leaves - j0 < (size_t)B3X_LEAVES ? num_leaves - j0 : (size_t)B3X_LEAVES) * 16;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const size_t e = 2 * ((size_t)threadIdx.x + B3X_LEAVES * i);
        if (e < avail) {
            const ulonglong2 va = *reinterpret_cast<const ulonglong2*>(plane_a + base + e);
            const ulonglong2 vb = *reinterpret_cast<const ulonglong2*>(plane_b + base + e);
            const size_t at = (e >> 4) * 17 + (e & 15);
            tile[0][at] = va.x; tile[0][at + 1] = va.y;
            tile[1][at] = vb.x; tile[1][at + 1] = vb.y;
        }
    }
    __syncthreads();
    const size_t j = j0 + threadIdx.x;
    if (j >= num_leaves) return;
    u32 d[8];
    b3_leaf<false>([&](u32 i) { return gl_canon(tile[i & 1][threadIdx.x * 17 + (i >> 1)]); }, 32u, d);
    b3_store_digest(out + j * 4, d);
}
__global__ __launch_bounds__(256) void merkle_level_b3_kernel(u64* __restrict__ heap, size_t first, size_t count) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    b3_node(heap + 8 * (first + t), heap + 4 * (first + t));
}
// `nlev` (<= 8) consecutive levels in one launch (round 6): a workgroup computes 256 nodes of the level of `first` nodes from their
// children in the heap, then the 128, 64, ... nodes above them out of LDS -- every level is still written to the heap (sibling paths
// read it), but only the bottom one is read back.  Measured on the 2^22-row Blake3 proof (profiles/r06_merkle_levels_per_launch.txt,
// 314 level launches and 20.6 GB per proof before): eight levels per launch are SLOWER (6.3 -> 8.4 ms: the upper levels run on 128, 64,
// ... 1 lanes of a workgroup that keeps its slot for eight compressions in a row), two or three are a little faster (5.8 ms), four 6.0;
// the default is two (OLA_MERKLE_FUSED_LEVELS, 0 = one launch per level).
__global__ __launch_bounds__(256) void merkle_levels_b3_kernel(u64* __restrict__ heap, size_t first, int nlev) {
    __shared__ u64 sh[2][256 * 4];
    const int t = threadIdx.x;
    u64 d[4];
    {
        const size_t node = first + (size_t)blockIdx.x * 256 + t;
        b3_node(heap + 8 * node, d);
#pragma unroll
        for (int k = 0; k < 4; k++) { heap[4 * node + k] = d[k]; sh[0][4 * t + k] = d[k]; }
    }
    for (int l = 1; l < nlev; l++) {
        __syncthreads();
        const int count = 256 >> l;
        if (t < count) {
            const size_t node = (first >> l) + (size_t)blockIdx.x * count + t;
            b3_node(&sh[(l - 1) & 1][8 * t], d);
#pragma unroll
            for (int k = 0; k < 4; k++) { heap[4 * node + k] = d[k]; sh[l & 1][4 * t + k] = d[k]; }
        }
    }
}
// levels of `first` nodes and fewer (first <= blockDim.x) down to the level of `last` nodes in one workgroup
__global__ __launch_bounds__(256) void merkle_top_b3_kernel(u64* __restrict__ heap, size_t first, size_t last) {
    for (size_t level = first; level >= last && level >= 1; level >>= 1) {
        if (threadIdx.x < level) b3_node(heap + 8 * (level + threadIdx.x), heap + 4 * (level + threadIdx.x));
        __threadfence_block();
        __syncthreads();
    }
}
static inline bool is_b3(const DeviceCtx* ctx) { return ctx->hasher == 1; }
static void require_leaf_width(size_t words) {
    if (words == 0 || words > 4096) throw OlaError(-1, "Blake3 leaves hold 1..4096 field elements");
}

// ---- host launchers ----
void poseidon_init(DeviceCtx*) { poseidon_upload_constants(); }

// hash invocations per leaf of `words` field elements: sponge permutations (rate 8), or Blake3 compressions (64-byte blocks,
// plus the parent compressions of a multi-chunk leaf)
static double leaf_hash_calls(const DeviceCtx* ctx, size_t words) {
    if (ctx->hasher == 1) { const size_t blocks = (words * 8 + 63) / 64, chunks = (words * 8 + 1023) / 1024; return (double)(blocks + chunks - 1); }
    return (double)((words + 7) / 8);
}
void launch_leaf_hash_colmajor(DeviceCtx* ctx, const u64* base, size_t col_stride, int ncols, size_t num_leaves,
                               u64* out) {
    PhaseScope ph(ctx, PH_LEAF_HASH, (double)num_leaves * leaf_hash_calls(ctx, (size_t)ncols), (double)num_leaves * ((size_t)ncols * 8 + 32));
    if (is_b3(ctx)) {
        require_leaf_width((size_t)ncols);
        if (num_leaves) {
            const dim3 grid((unsigned)((num_leaves + 255) / 256));
            if ((size_t)ncols > (size_t)B3_CHUNK_WORDS) hipLaunchKernelGGL(leaf_b3_colmajor_kernel<true>, grid, dim3(256), 0, ctx->stream, base, col_stride, ncols, num_leaves, out);
            else hipLaunchKernelGGL(leaf_b3_colmajor_kernel<false>, grid, dim3(256), 0, ctx->stream, base, col_stride, ncols, num_leaves, out);
        }
        return;
    }
    if (num_leaves <= QUAD_MAX) {
        hipLaunchKernelGGL(leaf_hash_colmajor_quad_kernel, dim3((unsigned)((4 * num_leaves + 255) / 256)), dim3(256), 0, ctx->stream, base,
                           col_stride, ncols, num_leaves, out);
        return;
    }
    const unsigned blocks = (unsigned)((num_leaves + 255) / 256);
    hipLaunchKernelGGL(leaf_hash_colmajor_kernel, dim3(blocks), dim3(256), 0, ctx->stream, base, col_stride, ncols,
                       num_leaves, out);
}
void launch_leaf_hash_rowmajor(DeviceCtx* ctx, const u64* rows, size_t row_len, size_t num_leaves, u64* out) {
    if (is_b3(ctx)) {
        require_leaf_width(row_len);
        if (num_leaves) {
            const dim3 grid((unsigned)((num_leaves + 255) / 256));
            if (row_len > (size_t)B3_CHUNK_WORDS) hipLaunchKernelGGL(leaf_b3_rowmajor_kernel<true>, grid, dim3(256), 0, ctx->stream, rows, row_len, num_leaves, out);
            else hipLaunchKernelGGL(leaf_b3_rowmajor_kernel<false>, grid, dim3(256), 0, ctx->stream, rows, row_len, num_leaves, out);
        }
        return;
    }
    if (num_leaves <= QUAD_MAX) {
        hipLaunchKernelGGL(leaf_hash_rowmajor_quad_kernel, dim3((unsigned)((4 * num_leaves + 255) / 256)), dim3(256), 0, ctx->stream, rows,
                           row_len, num_leaves, out);
        return;
    }
    const unsigned blocks = (unsigned)((num_leaves + 255) / 256);
    hipLaunchKernelGGL(leaf_hash_rowmajor_kernel, dim3(blocks), dim3(256), 0, ctx->stream, rows, row_len, num_leaves, out);
}
void launch_leaf_hash_ext(DeviceCtx* ctx, const u64* pa, const u64* pb, int arity, size_t num_leaves, u64* out) {
    PhaseScope ph(ctx, PH_LEAF_HASH, (double)num_leaves * leaf_hash_calls(ctx, (size_t)(2 * arity)), (double)num_leaves * ((size_t)arity * 16 + 32));
    if (is_b3(ctx)) {
        require_leaf_width((size_t)(2 * arity));
        if (num_leaves) {
            const dim3 grid((unsigned)((num_leaves + 255) / 256));
            static const bool staged = !(getenv("OLA_LEAF_EXT_STAGED") && !strcmp(getenv("OLA_LEAF_EXT_STAGED"), "0"));
            if (arity == 16 && staged && num_leaves >= 1024 && (((uintptr_t)pa | (uintptr_t)pb) & 15) == 0) {
                hipLaunchKernelGGL(leaf_b3_ext16_kernel, dim3((unsigned)((num_leaves + B3X_LEAVES - 1) / B3X_LEAVES)), dim3(B3X_LEAVES), 0, ctx->stream, pa, pb,
                                   num_leaves, out);
                return;
            }
            if ((size_t)(2 * arity) > (size_t)B3_CHUNK_WORDS) hipLaunchKernelGGL(leaf_b3_ext_kernel<true>, grid, dim3(256), 0, ctx->stream, pa, pb, arity, num_leaves, out);
            else hipLaunchKernelGGL(leaf_b3_ext_kernel<false>, grid, dim3(256), 0, ctx->stream, pa, pb, arity, num_leaves, out);
        }
        return;
    }
    if (num_leaves <= QUAD_MAX) {
        hipLaunchKernelGGL(leaf_hash_ext_quad_kernel, dim3((unsigned)((4 * num_leaves + 255) / 256)), dim3(256), 0, ctx->stream, pa, pb, arity,
                           num_leaves, out);
        return;
    }
    const unsigned blocks = (unsigned)((num_leaves + 255) / 256);
    hipLaunchKernelGGL(leaf_hash_ext_kernel, dim3(blocks), dim3(256), 0, ctx->stream, pa, pb, arity, num_leaves, out);
}
// heap[N..2N) must hold the leaf digests; fills heap[2^cap_height .. N): like the reference (merkle_tree/mod.rs:228-233)
// nothing above the cap is ever hashed
void launch_merkle_build(DeviceCtx* ctx, u64* heap, size_t num_leaves, uint32_t cap_height) {
    const size_t last = (size_t)1 << cap_height, top = 256;   // levels of <= `top` nodes share one launch (1024 threads)
    PhaseScope ph(ctx, PH_MERKLE_LEVELS, num_leaves > last ? (double)(num_leaves - last) : 0.0, num_leaves > last ? (double)(num_leaves - last) * 96 : 0.0);
    size_t level = num_leaves / 2;
    if (is_b3(ctx)) {
        static const int fused = getenv("OLA_MERKLE_FUSED_LEVELS") ? atoi(getenv("OLA_MERKLE_FUSED_LEVELS")) : 2;
        if (fused > 1 && last <= top && (level & (level - 1)) == 0) {
            while (level > top) {          // level is a power of two >= 512: up to `fused` levels per launch, the last of them still above `top`
                int nlev = 0;
                while (nlev < fused && nlev < 8 && (level >> nlev) > top) nlev++;
                hipLaunchKernelGGL(merkle_levels_b3_kernel, dim3((unsigned)(level / 256)), dim3(256), 0, ctx->stream, heap, level, nlev);
                level >>= nlev;
            }
        }
        for (; level >= last && level > top; level /= 2)
            hipLaunchKernelGGL(merkle_level_b3_kernel, dim3((unsigned)((level + 255) / 256)), dim3(256), 0, ctx->stream, heap, level, level);
        if (level >= last && level >= 1) hipLaunchKernelGGL(merkle_top_b3_kernel, dim3(1), dim3(256), 0, ctx->stream, heap, level, last);
        return;
    }
    for (; level >= last && level > top; level /= 2) {
        if (level <= QUAD_MAX) {
            hipLaunchKernelGGL(merkle_level_quad_kernel, dim3((unsigned)((4 * level + 255) / 256)), dim3(256), 0, ctx->stream, heap, level, level);
        } else {
            hipLaunchKernelGGL(merkle_level_kernel, dim3((unsigned)((level + 255) / 256)), dim3(256), 0, ctx->stream, heap, level, level);
        }
    }
    if (level >= last && level >= 1) hipLaunchKernelGGL(merkle_top_kernel, dim3(1), dim3(1024), 0, ctx->stream, heap, level, last);
}
void launch_poseidon_states(DeviceCtx* ctx, u64* states, size_t n) {
    if (n == 0) return;
    if (n <= QUAD_MAX) {
        hipLaunchKernelGGL(poseidon_states_quad_kernel, dim3((unsigned)((4 * n + 255) / 256)), dim3(256), 0, ctx->stream, states, n);
        return;
    }
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(poseidon_states_kernel, dim3(blocks), dim3(256), 0, ctx->stream, states, n);
}
// minimal nonce with `bits` leading zeros; scans batches of 2^20 nonces in increasing order
u64 run_pow(DeviceCtx* ctx, const u64 h[4], u32 bits) {
    unsigned long long* d_best = (unsigned long long*)ctx->alloc(8);
    const unsigned long long none = ~0ull;
    unsigned long long best = none;
    // expected witness ~2^bits: batches of 4 * 2^bits nonces find it in the first launch 98 % of the time
    const u64 batch = std::max<u64>((u64)1 << 14, (u64)4 << bits);
    for (u64 start = 0; best == none; start += batch) {
        HIP_CHECK(hipMemcpyAsync(d_best, &none, 8, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(pow_kernel, dim3((unsigned)(batch / 256)), dim3(256), 0, ctx->stream, h[0], h[1], h[2], h[3],
                           start, bits, d_best);
        HIP_CHECK(hipMemcpyAsync(&best, d_best, 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    ctx->free(d_best);
    return best;
}

}  // namespace ola
