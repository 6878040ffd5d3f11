// Micro-benchmark: issue rate of the VALU instructions 64-bit modular arithmetic is built from on gfx950.
// Each kernel runs a long unrolled sequence of one instruction on independent registers; with 8 waves per SIMD the
// result is the throughput in cycles per wave-instruction per SIMD (4 = full rate for wave64 on a 16-lane SIMD).
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rates tools/ubench/valu_rates.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
#define ITER 256

#define BENCH_KERNEL(name, asm_body, ...)                                                         \
    __global__ __launch_bounds__(256) void name(unsigned* out, unsigned seed) {                       \
        unsigned a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 ^ 0x1234567, a3 = a1 + 77;          \
        unsigned long long w0 = a0, w1 = a1, w2 = a2, w3 = a3;                                         \
        double d0 = a0, d1 = a1, d2 = a2, d3 = a3;                                                     \
        for (int i = 0; i < ITER; i++) { REP16(asm volatile(asm_body : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : : __VA_ARGS__);) } \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + (unsigned)(w0 + w1 + w2 + w3) + (unsigned)(d0 + d1 + d2 + d3); \
    }

// every body has 4 independent instructions
BENCH_KERNEL(k_mov, "v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0", "memory")
BENCH_KERNEL(k_add_u32, "v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0", "memory")
BENCH_KERNEL(k_add3, "v_add3_u32 %0, %0, %1, %2\n v_add3_u32 %1, %1, %2, %3\n v_add3_u32 %2, %2, %3, %0\n v_add3_u32 %3, %3, %0, %1", "memory")
BENCH_KERNEL(k_add_co, "v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %1, vcc, %1, %2, vcc\n v_add_co_u32 %2, vcc, %2, %3\n v_addc_co_u32 %3, vcc, %3, %0, vcc", "vcc")
BENCH_KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %1, %1, %2\n v_mul_lo_u32 %2, %2, %3\n v_mul_lo_u32 %3, %3, %0", "memory")
BENCH_KERNEL(k_mul_hi, "v_mul_hi_u32 %0, %0, %1\n v_mul_hi_u32 %1, %1, %2\n v_mul_hi_u32 %2, %2, %3\n v_mul_hi_u32 %3, %3, %0", "memory")
BENCH_KERNEL(k_mul_u24, "v_mul_u32_u24 %0, %0, %1\n v_mul_u32_u24 %1, %1, %2\n v_mul_u32_u24 %2, %2, %3\n v_mul_u32_u24 %3, %3, %0", "memory")
BENCH_KERNEL(k_mad_u24, "v_mad_u32_u24 %0, %0, %1, %2\n v_mad_u32_u24 %1, %1, %2, %3\n v_mad_u32_u24 %2, %2, %3, %0\n v_mad_u32_u24 %3, %3, %0, %1", "memory")
BENCH_KERNEL(k_mul_hi_u24, "v_mul_hi_u32_u24 %0, %0, %1\n v_mul_hi_u32_u24 %1, %1, %2\n v_mul_hi_u32_u24 %2, %2, %3\n v_mul_hi_u32_u24 %3, %3, %0", "memory")
BENCH_KERNEL(k_mad_u64_u32, "v_mad_u64_u32 %4, vcc, %0, %1, %4\n v_mad_u64_u32 %5, vcc, %1, %2, %5\n v_mad_u64_u32 %6, vcc, %2, %3, %6\n v_mad_u64_u32 %7, vcc, %3, %0, %7", "vcc")
BENCH_KERNEL(k_lshl_add_u64, "v_lshl_add_u64 %4, %4, 0, %5\n v_lshl_add_u64 %5, %5, 0, %6\n v_lshl_add_u64 %6, %6, 0, %7\n v_lshl_add_u64 %7, %7, 0, %4", "memory")
BENCH_KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc", "memory")
BENCH_KERNEL(k_cmp_u64, "v_cmp_lt_u64 vcc, %4, %5\n v_cmp_lt_u64 vcc, %5, %6\n v_cmp_lt_u64 vcc, %6, %7\n v_cmp_lt_u64 vcc, %7, %4", "vcc")
BENCH_KERNEL(k_cmp_u32, "v_cmp_lt_u32 vcc, %0, %1\n v_cmp_lt_u32 vcc, %1, %2\n v_cmp_lt_u32 vcc, %2, %3\n v_cmp_lt_u32 vcc, %3, %0", "vcc")
BENCH_KERNEL(k_fma_f64, "v_fma_f64 %8, %8, %9, %10\n v_fma_f64 %9, %9, %10, %11\n v_fma_f64 %10, %10, %11, %8\n v_fma_f64 %11, %11, %8, %9", "memory")
BENCH_KERNEL(k_fma_f32, "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %0\n v_fma_f32 %3, %3, %0, %1", "memory")
BENCH_KERNEL(k_lshlrev_b64, "v_lshlrev_b64 %4, 3, %4\n v_lshlrev_b64 %5, 5, %5\n v_lshlrev_b64 %6, 7, %6\n v_lshlrev_b64 %7, 9, %7", "memory")
BENCH_KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %1, 7\n v_alignbit_b32 %1, %1, %2, 9\n v_alignbit_b32 %2, %2, %3, 11\n v_alignbit_b32 %3, %3, %0, 13", "memory")
BENCH_KERNEL(k_pk_add_u16, "v_pk_add_u16 %0, %0, %1\n v_pk_add_u16 %1, %1, %2\n v_pk_add_u16 %2, %2, %3\n v_pk_add_u16 %3, %3, %0", "memory")
BENCH_KERNEL(k_mad_i32_i24, "v_mad_i32_i24 %0, %0, %1, %2\n v_mad_i32_i24 %1, %1, %2, %3\n v_mad_i32_i24 %2, %2, %3, %0\n v_mad_i32_i24 %3, %3, %0, %1", "memory")
BENCH_KERNEL(k_dot4_u8, "v_dot4_u32_u8 %0, %0, %1, %2\n v_dot4_u32_u8 %1, %1, %2, %3\n v_dot4_u32_u8 %2, %2, %3, %0\n v_dot4_u32_u8 %3, %3, %0, %1", "memory")

// v_cmp writing vcc followed by the v_cndmask that consumes it (the compiler's select idiom); counts as 4 instructions
BENCH_KERNEL(k_cmp_cnd, "v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_u32 vcc, %2, %3\n v_cndmask_b32 %0, %0, %1, vcc", "vcc")
BENCH_KERNEL(k_cmp_cnd_sgpr, "v_cmp_lt_u32 s[10:11], %0, %1\n v_cndmask_b32 %2, %2, %3, s[10:11]\n v_cmp_lt_u32 s[12:13], %2, %3\n v_cndmask_b32 %0, %0, %1, s[12:13]", "s10", "s11", "s12", "s13")
BENCH_KERNEL(k_cnd_sgpr_only, "v_cndmask_b32 %0, %0, %1, s[10:11]\n v_cndmask_b32 %1, %1, %2, s[10:11]\n v_cndmask_b32 %2, %2, %3, s[10:11]\n v_cndmask_b32 %3, %3, %0, s[10:11]", "memory")
BENCH_KERNEL(k_cnd_indep, "v_cndmask_b32 %0, %1, %2, vcc\n v_cndmask_b32 %1, %2, %3, vcc\n v_cndmask_b32 %2, %3, %0, vcc\n v_cndmask_b32 %3, %0, %1, vcc", "memory")
// carry-mask idiom: add, carry -> all-ones mask via subb, masked correction; 4 instructions
BENCH_KERNEL(k_carry_mask, "v_add_co_u32 %0, vcc, %0, %1\n v_subb_co_u32 %2, vcc, 0, 0, vcc\n v_and_b32 %2, %2, %3\n v_add_u32 %0, %0, %2", "vcc")
BENCH_KERNEL(k_and, "v_and_b32 %0, %0, %1\n v_and_b32 %1, %1, %2\n v_and_b32 %2, %2, %3\n v_and_b32 %3, %3, %0", "memory")
BENCH_KERNEL(k_sub_co, "v_sub_co_u32 %0, vcc, %0, %1\n v_subb_co_u32 %1, vcc, %1, %2, vcc\n v_sub_co_u32 %2, vcc, %2, %3\n v_subb_co_u32 %3, vcc, %3, %0, vcc", "vcc")
BENCH_KERNEL(k_mad_u64_chain, "v_mad_u64_u32 %4, vcc, %0, %1, %4\n v_mad_u64_u32 %4, vcc, %1, %2, %4\n v_mad_u64_u32 %4, vcc, %2, %3, %4\n v_mad_u64_u32 %4, vcc, %3, %0, %4", "vcc")
BENCH_KERNEL(k_add_u32_chain, "v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %2\n v_add_u32 %0, %0, %3\n v_add_u32 %0, %0, %1", "memory")
BENCH_KERNEL(k_xor, "v_xor_b32 %0, %0, %1\n v_xor_b32 %1, %1, %2\n v_xor_b32 %2, %2, %3\n v_xor_b32 %3, %3, %0", "memory")
BENCH_KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, 3, %1\n v_lshl_or_b32 %1, %1, 5, %2\n v_lshl_or_b32 %2, %2, 7, %3\n v_lshl_or_b32 %3, %3, 9, %0", "memory")
BENCH_KERNEL(k_sub_u32, "v_sub_u32 %0, %0, %1\n v_sub_u32 %1, %1, %2\n v_sub_u32 %2, %2, %3\n v_sub_u32 %3, %3, %0", "memory")

BENCH_KERNEL(k_mad_i64_i32, "v_mad_i64_i32 %4, vcc, %0, %1, %4\n v_mad_i64_i32 %5, vcc, %1, %2, %5\n v_mad_i64_i32 %6, vcc, %2, %3, %6\n v_mad_i64_i32 %7, vcc, %3, %0, %7", "vcc")
BENCH_KERNEL(k_and_lit, "v_and_b32 %0, 0xffffff, %1\n v_and_b32 %1, 0xffffff, %2\n v_and_b32 %2, 0xffffff, %3\n v_and_b32 %3, 0xffffff, %0", "memory")
BENCH_KERNEL(k_ashr, "v_ashrrev_i32 %0, 16, %1\n v_ashrrev_i32 %1, 16, %2\n v_ashrrev_i32 %2, 16, %3\n v_ashrrev_i32 %3, 16, %0", "memory")
BENCH_KERNEL(k_lshl32, "v_lshlrev_b32 %0, 6, %1\n v_lshlrev_b32 %1, 6, %2\n v_lshlrev_b32 %2, 6, %3\n v_lshlrev_b32 %3, 6, %0", "memory")
BENCH_KERNEL(k_lshr32, "v_lshrrev_b32 %0, 6, %1\n v_lshrrev_b32 %1, 6, %2\n v_lshrrev_b32 %2, 6, %3\n v_lshrrev_b32 %3, 6, %0", "memory")
BENCH_KERNEL(k_lshl_add_u32, "v_lshl_add_u32 %0, %1, 6, %2\n v_lshl_add_u32 %1, %2, 6, %3\n v_lshl_add_u32 %2, %3, 6, %0\n v_lshl_add_u32 %3, %0, 6, %1", "memory")
BENCH_KERNEL(k_perm, "v_perm_b32 %0, %1, %2, %3\n v_perm_b32 %1, %2, %3, %0\n v_perm_b32 %2, %3, %0, %1\n v_perm_b32 %3, %0, %1, %2", "memory")
BENCH_KERNEL(k_add_sdwa, "v_add_u32_sdwa %0, sext(%1), %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %1, sext(%2), %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %2, sext(%3), %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %3, sext(%0), %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD", "memory")
BENCH_KERNEL(k_or, "v_or_b32 %0, %0, %1\n v_or_b32 %1, %1, %2\n v_or_b32 %2, %2, %3\n v_or_b32 %3, %3, %0", "memory")
BENCH_KERNEL(k_bfe, "v_bfe_u32 %0, %1, 8, 24\n v_bfe_u32 %1, %2, 8, 24\n v_bfe_u32 %2, %3, 8, 24\n v_bfe_u32 %3, %0, 8, 24", "memory")
BENCH_KERNEL(k_and_or, "v_and_or_b32 %0, %1, %2, %3\n v_and_or_b32 %1, %2, %3, %0\n v_and_or_b32 %2, %3, %0, %1\n v_and_or_b32 %3, %0, %1, %2", "memory")
// the butterfly stream: dependent adds and subs on 8 registers (as the T-form radix kernels issue them)
BENCH_KERNEL(k_bfly, "v_add_u32 %0, %0, %1\n v_sub_u32 %1, %0, %1\n v_add_u32 %2, %2, %3\n v_sub_u32 %3, %2, %3", "memory")
BENCH_KERNEL(k_mix, "v_mad_i64_i32 %4, vcc, %0, %1, %4\n v_and_b32 %0, 0xffffff, %2\n v_alignbit_b32 %1, %2, %3, 24\n v_add_u32 %2, %2, %3", "vcc")

// VGPR banks: does it matter which registers the two sources of a fast-class instruction come from?  Explicit registers,
// index mod 4 equal ("same bank": what limb i of two T-form elements in 4-aligned register quadruples gives) against
// all different.  (round 4)
#define BANK_CLOBBER "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55"
BENCH_KERNEL(k_add_same_bank, "v_add_u32 v40, v44, v48\n v_add_u32 v41, v45, v49\n v_add_u32 v42, v46, v50\n v_add_u32 v43, v47, v51", BANK_CLOBBER)
BENCH_KERNEL(k_add_diff_bank, "v_add_u32 v40, v45, v50\n v_add_u32 v41, v46, v51\n v_add_u32 v42, v47, v48\n v_add_u32 v43, v44, v49", BANK_CLOBBER)
BENCH_KERNEL(k_add_src_same_dst_diff, "v_add_u32 v41, v44, v48\n v_add_u32 v42, v45, v49\n v_add_u32 v43, v46, v50\n v_add_u32 v40, v47, v51", BANK_CLOBBER)
BENCH_KERNEL(k_bfly_same_bank, "v_add_u32 v40, v44, v48\n v_sub_u32 v52, v44, v48\n v_add_u32 v41, v45, v49\n v_sub_u32 v53, v45, v49", BANK_CLOBBER)
BENCH_KERNEL(k_bfly_diff_bank, "v_add_u32 v40, v44, v49\n v_sub_u32 v52, v44, v49\n v_add_u32 v41, v45, v50\n v_sub_u32 v53, v45, v50", BANK_CLOBBER)
BENCH_KERNEL(k_mad_same_bank, "v_mad_i64_i32 v[40:41], vcc, v44, v48, v[52:53]\n v_mad_i64_i32 v[42:43], vcc, v45, v49, v[54:55]\n v_mad_i64_i32 v[40:41], vcc, v46, v50, v[52:53]\n v_mad_i64_i32 v[42:43], vcc, v47, v51, v[54:55]", BANK_CLOBBER, "vcc")
BENCH_KERNEL(k_mad_diff_bank, "v_mad_i64_i32 v[40:41], vcc, v45, v50, v[52:53]\n v_mad_i64_i32 v[42:43], vcc, v46, v51, v[54:55]\n v_mad_i64_i32 v[40:41], vcc, v47, v48, v[52:53]\n v_mad_i64_i32 v[42:43], vcc, v44, v49, v[54:55]", BANK_CLOBBER, "vcc")

typedef void (*kern_t)(unsigned*, unsigned);
struct B { const char* name; kern_t k; };

int main(int argc, char** argv) {
    const int waves_per_simd = argc > 1 ? atoi(argv[1]) : 8;   // 8 = saturated; 2 = what a 200-VGPR kernel runs with
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double mhz = prop.clockRate / 1000.0;
    printf("device %s, %d CUs, %.0f MHz\n", prop.name, cus, mhz);
    const int blocks = cus * waves_per_simd;  // N blocks x 4 waves per CU = N waves per SIMD
    printf("%d waves per SIMD\n", waves_per_simd);
    unsigned* out;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    B list[] = {{"v_mov_b32", k_mov}, {"v_add_u32", k_add_u32}, {"v_add3_u32", k_add3}, {"v_add_co/addc_co_u32", k_add_co}, {"v_mul_lo_u32", k_mul_lo},
                {"v_mul_hi_u32", k_mul_hi}, {"v_mul_u32_u24", k_mul_u24}, {"v_mad_u32_u24", k_mad_u24}, {"v_mul_hi_u32_u24", k_mul_hi_u24},
                {"v_mad_u64_u32", k_mad_u64_u32}, {"v_lshl_add_u64", k_lshl_add_u64}, {"v_cndmask_b32", k_cndmask}, {"v_cmp_lt_u64", k_cmp_u64},
                {"v_cmp_lt_u32", k_cmp_u32}, {"v_fma_f64", k_fma_f64}, {"v_fma_f32", k_fma_f32}, {"v_lshlrev_b64", k_lshlrev_b64},
                {"v_alignbit_b32", k_alignbit}, {"v_pk_add_u16", k_pk_add_u16}, {"v_mad_i32_i24", k_mad_i32_i24}, {"v_dot4_u32_u8", k_dot4_u8},
                {"cmp(vcc)+cndmask", k_cmp_cnd}, {"cmp(sgpr)+cndmask", k_cmp_cnd_sgpr}, {"cndmask sgpr mask", k_cnd_sgpr_only}, {"cndmask vcc indep", k_cnd_indep},
                {"add_co,subb,and,add", k_carry_mask}, {"v_and_b32", k_and}, {"v_sub_co/subb_co", k_sub_co}, {"v_mad_u64_u32 chain", k_mad_u64_chain},
                {"v_add_u32 chain", k_add_u32_chain}, {"v_xor_b32", k_xor}, {"v_lshl_or_b32", k_lshl_or}, {"v_sub_u32", k_sub_u32},
                {"v_mad_i64_i32", k_mad_i64_i32}, {"v_and_b32 literal", k_and_lit}, {"v_ashrrev_i32", k_ashr}, {"v_lshlrev_b32", k_lshl32},
                {"v_lshrrev_b32", k_lshr32}, {"v_lshl_add_u32", k_lshl_add_u32}, {"v_perm_b32", k_perm}, {"v_add_u32_sdwa sext", k_add_sdwa},
                {"v_or_b32", k_or}, {"v_bfe_u32", k_bfe}, {"v_and_or_b32", k_and_or}, {"add/sub butterflies", k_bfly}, {"mad,and,alignbit,add", k_mix},
                {"v_add_u32 src same bank", k_add_same_bank}, {"v_add_u32 all banks differ", k_add_diff_bank}, {"v_add_u32 src same, dst other", k_add_src_same_dst_diff},
                {"butterfly src same bank", k_bfly_same_bank}, {"butterfly src diff bank", k_bfly_diff_bank},
                {"v_mad_i64_i32 src same bank", k_mad_same_bank}, {"v_mad_i64_i32 src diff bank", k_mad_diff_bank}};
    if (argc > 2) { const int skip = atoi(argv[2]); for (int i = 0; i + skip < (int)(sizeof(list) / sizeof(list[0])); i++) list[i] = list[i + skip]; }   // argv[2]: start at entry N
    const int n_run = (int)(sizeof(list) / sizeof(list[0])) - (argc > 2 ? atoi(argv[2]) : 0);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int bi = 0; bi < n_run; bi++) {
        B& b = list[bi];
        hipLaunchKernelGGL(b.k, dim3(blocks), dim3(256), 0, 0, out, 1u);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; r++) hipLaunchKernelGGL(b.k, dim3(blocks), dim3(256), 0, 0, out, 1u);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        ms /= 5;
        // per SIMD: 8 waves x ITER x 16 x 4 instructions
        const double instr_per_simd = (double)waves_per_simd * ITER * 16 * 4;
        const double cycles = ms * 1e-3 * mhz * 1e6;
        printf("%-22s %8.3f ms  %6.2f cycles / wave-instruction\n", b.name, ms, cycles / instr_per_simd);
    }
    return 0;
}
