// GPU self-test of the third-generation NTT pass kernels, pass by pass: every kernel variant (strided widths 5..9, contiguous and
// natural-order closing passes, forward / inverse, with and without the coset multipliers) is launched through ntt3_launch on
// random data and compared word for word with the host execution of the same per-thread phases (tests/host_ntt3_check.cpp).
// build (on the GPU box): hipcc --offload-arch=gfx950 -O2 -std=c++17 -c -o /tmp/st.o tests/gpu_ntt3_selftest.cpp && hipcc -o gpu_ntt3_selftest /tmp/st.o olavm_amd/lib/obj/ntt3.o
#define NTT3_NO_MAIN
#include "host_ntt3_check.cpp"

#include <hip/hip_runtime.h>

#include "../olavm_amd/csrc/device_ctx.h"
#include "../olavm_amd/csrc/ntt3.h"

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(2); } } while (0)

static u64* up(const std::vector<u64>& v) { u64* d; CK(hipMalloc(&d, v.size() * 8 + 8)); CK(hipMemcpy(d, v.data(), v.size() * 8, hipMemcpyHostToDevice)); return d; }

template <int MODE, bool INV>
static int one(int R, int L, int lo, bool coset, bool inplace) {
    const size_t n = (size_t)1 << L, cols = 3, cosets = coset ? 3 : 1;
    std::vector<u64> in(cols * n * (inplace ? cosets : 1));
    for (auto& v : in) v = rnd();
    for (size_t i = 0; i < n; i += 97) in[i] = 0xFFFFFFFFFFFFFFFFull;
    // tables
    u64 rootR = gl_root_of_unity(R), rootP = gl_root_of_unity(lo + R);
    if (INV) { rootR = gl_inv(rootR); rootP = gl_inv(rootP); }
    std::vector<u64> tw = powers(rootR, (size_t)1 << R), ptw, dig;
    std::vector<u64> shifts = {GL_GENERATOR, gl_mul(GL_GENERATOR, gl_root_of_unity(L + 1)), gl_mul(GL_GENERATOR, gl_root_of_unity(L + 2))};
    const size_t blk = (size_t)1 << (lo + R);
    if (MODE == N3_STRIDED) {
        ptw.resize(blk * cosets);
        for (size_t cs = 0; cs < cosets; cs++)
            for (size_t d = 0; d < ((size_t)1 << R); d++) {
                const u64 wq = gl_pow(rootP, brev((u32)d, R));
                u64 w = 12345, sm = 1;
                for (size_t M = 0; M < ((size_t)1 << lo); M++) { ptw[cs * blk + (d << lo) + M] = coset ? gl_mul(w, sm) : w; w = gl_mul(w, wq); sm = gl_mul(sm, shifts[cs]); }
            }
        if (coset) for (size_t cs = 0; cs < cosets; cs++) { std::vector<u64> dd = powers(gl_pow(shifts[cs], (u64)1 << lo), (size_t)1 << R); dig.insert(dig.end(), dd.begin(), dd.end()); }
    }
    N3Params p = {};
    p.log_n = L; p.lo = lo; p.ncols = cols;
    p.in_col_stride = inplace ? n * cosets : n; p.in_coset_stride = inplace ? n : 0;
    p.out_col_stride = n * cosets; p.out_coset_stride = n;
    p.ptw_coset_stride = coset ? blk : 0;
    // host
    std::vector<u64> hout(cols * n * cosets, 0);
    N3Params hp = p;
    hp.in = in.data(); hp.out = hout.data(); hp.tw = tw.data(); hp.ptw = ptw.empty() ? nullptr : ptw.data(); hp.sc_dig = dig.empty() ? nullptr : dig.data();
    dispatch_pass<MODE, INV, i32>(R, hp, cols, cosets);
    // device
    u64 *din = up(in), *dout, *dtw = up(tw), *dptw = ptw.empty() ? nullptr : up(ptw), *ddig = dig.empty() ? nullptr : up(dig);
    if (inplace) dout = din; else { CK(hipMalloc(&dout, hout.size() * 8)); CK(hipMemset(dout, 0, hout.size() * 8)); }
    N3Params dp = p;
    dp.in = din; dp.out = dout; dp.tw = dtw; dp.ptw = dptw; dp.sc_dig = ddig;
    int bad = 0;
    try { ntt3_launch(dp, R, MODE, INV, cols, cosets, 0); } catch (const OlaError& e) { printf("launch error: %s\n", e.what()); bad = 1; }
    CK(hipDeviceSynchronize());
    std::vector<u64> got(hout.size());
    CK(hipMemcpy(got.data(), dout, got.size() * 8, hipMemcpyDeviceToHost));
    size_t first = 0;
    for (size_t i = 0; i < got.size(); i++) if (got[i] != hout[i]) { if (!bad) first = i; bad++; }
    printf("mode %d R=%2d L=%d lo=%2d inverse=%d coset=%d inplace=%d : %s", MODE, R, L, lo, (int)INV, (int)coset, (int)inplace, bad ? "FAIL" : "ok");
    if (bad) printf(" (%d words differ, first at %zu: got %llx want %llx)", bad, first, got[first], hout[first]);
    printf("\n");
    CK(hipFree(din)); if (!inplace) CK(hipFree(dout)); CK(hipFree(dtw)); if (dptw) CK(hipFree(dptw)); if (ddig) CK(hipFree(ddig));
    return bad != 0;
}

int main() {
    int fails = 0;
    const int L = 18;
    for (int R = 5; R <= 9; R++) {
        fails += one<N3_STRIDED, false>(R, L, L - R, false, false);
        fails += one<N3_STRIDED, true>(R, L, L - R, false, false);
        fails += one<N3_STRIDED, false>(R, L, L - R, true, false);
        fails += one<N3_STRIDED, false>(R, L, 8, false, true);   // a middle pass (upper index bits present), in place
    }
    fails += one<N3_LAST_BITREV, false>(13, L, 0, false, true);
    fails += one<N3_LAST_BITREV, true>(13, L, 0, false, true);
    fails += one<N3_LAST_BITREV, false>(13, L, 0, true, true);    // several coset slices in place
    fails += one<N3_LAST_NATURAL, false>(9, L, 0, false, false);
    fails += one<N3_LAST_NATURAL, true>(9, L, 0, false, false);
    fails += one<N3_LAST_BITREV, false>(13, 20, 0, false, true);
    fails += one<N3_STRIDED, false>(7, 20, 13, true, false);
    printf(fails ? "FAILED\n" : "all ok\n");
    return fails ? 1 : 0;
}
