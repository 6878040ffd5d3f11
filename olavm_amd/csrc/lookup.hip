// Device-side `permuted_cols`: the permuted input / permuted table columns of the Halo2-style lookup
// argument that the range-check, bitwise and program tables carry (reference: circuits/src/stark/lookup.rs:68-132, called
// from generation/builtin.rs:121-200 and generation/prog.rs).  Own translation unit: rocPRIM (radix sort, scans) is only
// needed here.
//
// The reference walks the two SORTED columns with one sequential merge loop that keeps a stack of "unused" table values:
//   table value not wanted by any input  -> push;   repeated input (its table entry is already taken) -> pop, or, with
//   an empty stack, remember the slot;   at the end the remembered slots and the inputs left over when the table ran
//   out are filled, in order, with what is still on the stack (bottom first).
// Restated as data-parallel steps:
//   1. canonicalise and radix-sort both columns;
//   2. classify every element with binary searches: input i of value a with rank r among the inputs equal to a is MATCHED
//      iff r < (number of table entries equal to a); otherwise it is a POP if some table entry is > a (the merge loop is
//      still running when it is reached) and a TAIL slot if not.  Table entry j is matched iff its rank among equals is
//      below the number of inputs of that value, else it is a PUSH.  A value has surplus inputs or surplus table entries,
//      never both, so ordering the pushes and pops by (value, index) reproduces the order in which the loop meets them;
//      the position of an event in that sequence follows from two exclusive scans;
//   3. stack discipline = bracket matching: S = running sum of +1 / -1, m = running minimum of min(S, 0); a pop that
//      lowers m found the stack empty; the stack height after event k is d = S - m.  A stable sort of the events by the
//      height they push to / pop from puts every pop right behind the push it takes;
//   4. pushes nobody took (in event order) fill the slots of the unmatched pops (in event order) followed by the tail
//      slots (in index order).
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "device_ctx.h"
#include "gl.cuh"
#include "lookup.h"

namespace ola {

namespace {

enum : u32 { K_MATCHED = 0, K_EVENT = 1, K_TAIL = 2 };

struct Scratch {
    DeviceCtx* ctx;
    std::vector<void*> ptrs;
    explicit Scratch(DeviceCtx* c) : ctx(c) {}
    template <typename T>
    T* alloc(size_t elems) {
        void* p = ctx->alloc(elems * sizeof(T));
        ptrs.push_back(p);
        return (T*)p;
    }
    ~Scratch() {
        (void)hipStreamSynchronize(ctx->stream);
        for (void* p : ptrs) ctx->free(p);
    }
};

__global__ __launch_bounds__(256) void canon_kernel(const u64* __restrict__ in, u64* __restrict__ out, u32 n) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = gl_canon(in[i]);
}

__device__ __forceinline__ u32 lower_bound_u64(const u64* __restrict__ a, u32 n, u64 v) {
    u32 lo = 0, hi = n;
    while (lo < hi) {
        const u32 mid = lo + ((hi - lo) >> 1);
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ u32 upper_bound_u64(const u64* __restrict__ a, u32 n, u64 v) {
    u32 lo = 0, hi = n;
    while (lo < hi) {
        const u32 mid = lo + ((hi - lo) >> 1);
        if (a[mid] <= v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// step 2: thread t < n classifies input t, thread n + t classifies table entry t.  other[t] = first position of the
// opposite column whose value is not below this one (where events of smaller values end).
__global__ __launch_bounds__(256) void classify_kernel(const u64* __restrict__ si, const u64* __restrict__ st, u32 n,
                                                       u32* __restrict__ kind_in, u32* __restrict__ kind_tab,
                                                       u32* __restrict__ other_in, u32* __restrict__ other_tab,
                                                       u64* __restrict__ permuted_table) {
    const u32 t = blockIdx.x * 256 + threadIdx.x;
    if (t < n) {
        const u64 a = si[t];
        const u32 rank = t - lower_bound_u64(si, n, a);
        const u32 lb = lower_bound_u64(st, n, a), ub = upper_bound_u64(st, n, a);
        other_in[t] = lb;
        if (rank < ub - lb) {
            kind_in[t] = K_MATCHED;
            permuted_table[t] = a;
        } else {
            kind_in[t] = ub < n ? K_EVENT : K_TAIL;
        }
    } else if (t < 2 * n) {
        const u32 j = t - n;
        const u64 b = st[j];
        const u32 rank = j - lower_bound_u64(st, n, b);
        const u32 lb = lower_bound_u64(si, n, b), ub = upper_bound_u64(si, n, b);
        other_tab[j] = lb;
        kind_tab[j] = rank < ub - lb ? K_MATCHED : K_EVENT;
    }
}

__global__ __launch_bounds__(256) void flag_kernel(const u32* __restrict__ kind, u32 n, u32 which, u32* __restrict__ flag) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i <= n) flag[i] = (i < n && kind[i] == which) ? 1u : 0u;      // n + 1 entries: the scan's last one is the total
}

// events in loop order: delta[e] = +1 (push) / -1 (pop), src[e] = table index / input index
__global__ __launch_bounds__(256) void place_events_kernel(u32 n, const u32* __restrict__ kind_in, const u32* __restrict__ kind_tab,
                                                           const u32* __restrict__ other_in, const u32* __restrict__ other_tab,
                                                           const u32* __restrict__ pop_before, const u32* __restrict__ push_before,
                                                           int* __restrict__ delta, u32* __restrict__ src) {
    const u32 t = blockIdx.x * 256 + threadIdx.x;
    if (t < n) {
        if (kind_in[t] != K_EVENT) return;
        const u32 e = pop_before[t] + push_before[other_in[t]];
        delta[e] = -1;
        src[e] = t;
    } else if (t < 2 * n) {
        const u32 j = t - n;
        if (kind_tab[j] != K_EVENT) return;
        const u32 e = push_before[j] + pop_before[other_tab[j]];
        delta[e] = 1;
        src[e] = j;
    }
}

__global__ __launch_bounds__(256) void clamp_min_kernel(const int* __restrict__ s, u32 count, int* __restrict__ m) {
    const u32 e = blockIdx.x * 256 + threadIdx.x;
    if (e < count) m[e] = s[e] < 0 ? s[e] : 0;
}

// level key of every event: pushes sort under the height they create, matched pops under the height they remove;
// pops that found the stack empty get key 0 and an `empty_pop` flag.
__global__ __launch_bounds__(256) void level_kernel(const int* __restrict__ delta, const int* __restrict__ s, const int* __restrict__ runmin,
                                                    u32 count, u32* __restrict__ key, u32* __restrict__ ident, u32* __restrict__ empty_pop) {
    const u32 e = blockIdx.x * 256 + threadIdx.x;
    if (e > count) return;
    if (e == count) { empty_pop[e] = 0; return; }
    const int m = runmin[e], m_prev = e ? runmin[e - 1] : 0;
    const int height = s[e] - m;                      // stack height after the event
    const bool unmatched = delta[e] < 0 && s[e] < m_prev;
    ident[e] = e;
    empty_pop[e] = unmatched ? 1u : 0u;
    key[e] = delta[e] > 0 ? (u32)height : (unmatched ? 0u : (u32)height + 1u);
}

// after the stable sort by level a matched pop sits right behind its push
__global__ __launch_bounds__(256) void pair_kernel(const u32* __restrict__ key_sorted, const u32* __restrict__ ev_sorted, u32 count,
                                                   const int* __restrict__ delta, const u32* __restrict__ src, const u64* __restrict__ st,
                                                   u64* __restrict__ permuted_table, u32* __restrict__ free_push) {
    const u32 q = blockIdx.x * 256 + threadIdx.x;
    if (q > count) return;
    if (q == count) { free_push[count] = 0; return; }
    const u32 e = ev_sorted[q];
    if (delta[e] > 0) {
        const bool taken = q + 1 < count && key_sorted[q + 1] == key_sorted[q] && delta[ev_sorted[q + 1]] < 0;
        free_push[e] = taken ? 0u : 1u;
    } else {
        free_push[e] = 0;
        if (key_sorted[q] != 0) permuted_table[src[e]] = st[src[ev_sorted[q - 1]]];
    }
}

// step 4: slot list (unmatched pops, then tail inputs) and value list (free pushes), then the fill
__global__ __launch_bounds__(256) void gather_lists_kernel(u32 n, u32 count, const int* __restrict__ delta, const u32* __restrict__ src,
                                                           const u32* __restrict__ empty_pop, const u32* __restrict__ empty_before,
                                                           const u32* __restrict__ free_push, const u32* __restrict__ free_before,
                                                           const u32* __restrict__ kind_in, const u32* __restrict__ tail_before,
                                                           const u64* __restrict__ st, u32* __restrict__ slots, u64* __restrict__ values) {
    const u32 t = blockIdx.x * 256 + threadIdx.x;
    if (t < count) {
        if (empty_pop[t]) slots[empty_before[t]] = src[t];
        if (free_push[t]) values[free_before[t]] = st[src[t]];
    }
    if (t < n && kind_in[t] == K_TAIL) slots[empty_before[count] + tail_before[t]] = t;
}

__global__ __launch_bounds__(256) void fill_kernel(const u32* __restrict__ slots, const u64* __restrict__ values,
                                                   const u32* __restrict__ free_total, u64* __restrict__ permuted_table) {
    const u32 k = blockIdx.x * 256 + threadIdx.x;
    if (k < *free_total) permuted_table[slots[k]] = values[k];
}

inline unsigned blocks(size_t n) { return (unsigned)((n + 255) / 256); }

void sort_u64(Scratch& mem, hipStream_t stream, const u64* in, u64* out, size_t n) {
    size_t bytes = 0;
    HIP_CHECK(rocprim::radix_sort_keys(nullptr, bytes, in, out, n, 0, 64, stream));
    void* tmp = mem.alloc<unsigned char>(bytes);
    HIP_CHECK(rocprim::radix_sort_keys(tmp, bytes, in, out, n, 0, 64, stream));
}

template <typename T, typename Op>
void scan_exclusive(Scratch& mem, hipStream_t stream, const T* in, T* out, T init, size_t n, Op op) {
    size_t bytes = 0;
    HIP_CHECK(rocprim::exclusive_scan(nullptr, bytes, in, out, init, n, op, stream));
    void* tmp = mem.alloc<unsigned char>(bytes);
    HIP_CHECK(rocprim::exclusive_scan(tmp, bytes, in, out, init, n, op, stream));
}
template <typename T, typename Op>
void scan_inclusive(Scratch& mem, hipStream_t stream, const T* in, T* out, size_t n, Op op) {
    size_t bytes = 0;
    HIP_CHECK(rocprim::inclusive_scan(nullptr, bytes, in, out, n, op, stream));
    void* tmp = mem.alloc<unsigned char>(bytes);
    HIP_CHECK(rocprim::inclusive_scan(tmp, bytes, in, out, n, op, stream));
}

}  // namespace

void permuted_cols_dev(DeviceCtx* ctx, const u64* inputs, const u64* table, size_t n_, u64* permuted_inputs, u64* permuted_table) {
    if (n_ == 0) return;
    if (n_ >= ((size_t)1 << 30)) throw OlaError(-2, "permuted_cols: more than 2^30 rows");
    const u32 n = (u32)n_;
    hipStream_t stream = ctx->stream;
    Scratch mem(ctx);
    // ---- 1. canonical, sorted
    u64* canon = mem.alloc<u64>(n);
    u64* st = mem.alloc<u64>(n);
    u64* si = permuted_inputs;
    hipLaunchKernelGGL(canon_kernel, dim3(blocks(n)), dim3(256), 0, stream, inputs, canon, n);
    sort_u64(mem, stream, canon, si, n);
    hipLaunchKernelGGL(canon_kernel, dim3(blocks(n)), dim3(256), 0, stream, table, canon, n);
    sort_u64(mem, stream, canon, st, n);
    // ---- 2. classify, order the events
    u32* kind_in = mem.alloc<u32>(n);
    u32* kind_tab = mem.alloc<u32>(n);
    u32* other_in = mem.alloc<u32>(n);
    u32* other_tab = mem.alloc<u32>(n);
    hipLaunchKernelGGL(classify_kernel, dim3(blocks(2 * (size_t)n)), dim3(256), 0, stream, si, st, n, kind_in, kind_tab, other_in, other_tab,
                       permuted_table);
    u32* flag = mem.alloc<u32>(n + 1);
    u32* pop_before = mem.alloc<u32>(n + 1);
    u32* push_before = mem.alloc<u32>(n + 1);
    u32* tail_before = mem.alloc<u32>(n + 1);
    hipLaunchKernelGGL(flag_kernel, dim3(blocks(n + 1)), dim3(256), 0, stream, kind_in, n, (u32)K_EVENT, flag);
    scan_exclusive(mem, stream, flag, pop_before, 0u, n + 1, rocprim::plus<u32>());
    hipLaunchKernelGGL(flag_kernel, dim3(blocks(n + 1)), dim3(256), 0, stream, kind_tab, n, (u32)K_EVENT, flag);
    scan_exclusive(mem, stream, flag, push_before, 0u, n + 1, rocprim::plus<u32>());
    hipLaunchKernelGGL(flag_kernel, dim3(blocks(n + 1)), dim3(256), 0, stream, kind_in, n, (u32)K_TAIL, flag);
    scan_exclusive(mem, stream, flag, tail_before, 0u, n + 1, rocprim::plus<u32>());
    u32 totals[2];
    HIP_CHECK(hipMemcpyAsync(&totals[0], pop_before + n, 4, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipMemcpyAsync(&totals[1], push_before + n, 4, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    const u32 count = totals[0] + totals[1];
    // ---- 3. the stack, as bracket matching
    int* delta = mem.alloc<int>(count + 1);
    u32* src = mem.alloc<u32>(count + 1);
    u32* empty_pop = mem.alloc<u32>(count + 1);
    u32* empty_before = mem.alloc<u32>(count + 1);
    u32* free_push = mem.alloc<u32>(count + 1);
    u32* free_before = mem.alloc<u32>(count + 1);
    if (count) {
        int* s = mem.alloc<int>(count);
        int* runmin = mem.alloc<int>(count);
        int* clamped = mem.alloc<int>(count);
        u32* key = mem.alloc<u32>(count);
        u32* ident = mem.alloc<u32>(count);
        u32* key_sorted = mem.alloc<u32>(count);
        u32* ev_sorted = mem.alloc<u32>(count);
        hipLaunchKernelGGL(place_events_kernel, dim3(blocks(2 * (size_t)n)), dim3(256), 0, stream, n, kind_in, kind_tab, other_in, other_tab,
                           pop_before, push_before, delta, src);
        scan_inclusive(mem, stream, delta, s, count, rocprim::plus<int>());
        hipLaunchKernelGGL(clamp_min_kernel, dim3(blocks(count)), dim3(256), 0, stream, s, count, clamped);
        scan_inclusive(mem, stream, clamped, runmin, count, rocprim::minimum<int>());
        hipLaunchKernelGGL(level_kernel, dim3(blocks(count + 1)), dim3(256), 0, stream, delta, s, runmin, count, key, ident, empty_pop);
        size_t bytes = 0;
        HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, key, key_sorted, ident, ev_sorted, count, 0, 32, stream));
        void* tmp = mem.alloc<unsigned char>(bytes);
        HIP_CHECK(rocprim::radix_sort_pairs(tmp, bytes, key, key_sorted, ident, ev_sorted, count, 0, 32, stream));
        hipLaunchKernelGGL(pair_kernel, dim3(blocks(count + 1)), dim3(256), 0, stream, key_sorted, ev_sorted, count, delta, src, st,
                           permuted_table, free_push);
    } else {
        HIP_CHECK(hipMemsetAsync(empty_pop, 0, 4, stream));
        HIP_CHECK(hipMemsetAsync(free_push, 0, 4, stream));
    }
    scan_exclusive(mem, stream, empty_pop, empty_before, 0u, count + 1, rocprim::plus<u32>());
    scan_exclusive(mem, stream, free_push, free_before, 0u, count + 1, rocprim::plus<u32>());
    // ---- 4. leftovers
    u32* slots = mem.alloc<u32>(n);
    u64* values = mem.alloc<u64>(n);
    hipLaunchKernelGGL(gather_lists_kernel, dim3(blocks(std::max<size_t>(n, count))), dim3(256), 0, stream, n, count, delta, src, empty_pop,
                       empty_before, free_push, free_before, kind_in, tail_before, st, slots, values);
    hipLaunchKernelGGL(fill_kernel, dim3(blocks(n)), dim3(256), 0, stream, slots, values, free_before + count, permuted_table);
    HIP_CHECK(hipGetLastError());
}

}  // namespace ola
