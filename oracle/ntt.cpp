// ORACLE -- test infrastructure only (see gl.hpp header).  CPU restatement of the reference's `cfft` NTT.
//
// Follows (relative to /root/reference/plonky2/field/src):
//   cfft/mod.rs:233-271   get_twiddles / get_inv_twiddles  (w^i, i < n/2, bit-reversed; inverse root = w^(n-1))
//   cfft/mod.rs:282-290   permute_index
//   cfft/serial.rs:9-16   evaluate_poly               coeffs (natural) -> values at w^i (natural)
//   cfft/serial.rs:20-50  evaluate_poly_with_offset   coeffs -> values at offset*g^m, m < n*blowup (natural)
//   cfft/serial.rs:52-62  interpolate_poly            values -> coeffs, scaled by n^-1
//   cfft/serial.rs:64-78  interpolate_poly_with_offset
//   cfft/serial.rs:89-127 fft_in_place  (in-place butterflies with bit-reversed twiddles, output bit-reversed)
//   fft.rs:218-252        naive O(n^2) evaluation used by the reference's own test as the cross-check
#include "oracle.hpp"

namespace ola_oracle {

void permute(u64* v, size_t n) {  // serial.rs:80-88
    int bits = log2_strict(n);
    for (size_t i = 0; i < n; i++) {
        size_t j = reverse_bits(i, bits);
        if (j > i) { u64 t = v[i]; v[i] = v[j]; v[j] = t; }
    }
}

std::vector<u64> get_twiddles(size_t n) {
    u64 root = gl_root_of_unity(log2_strict(n));
    std::vector<u64> tw(n / 2);
    u64 acc = 1;
    for (size_t i = 0; i < n / 2; i++) { tw[i] = acc; acc = gl_mul(acc, root); }
    if (tw.size() > 1) permute(tw.data(), tw.size());
    return tw;
}

std::vector<u64> get_inv_twiddles(size_t n) {
    u64 root = gl_root_of_unity(log2_strict(n));
    u64 inv_root = gl_pow(root, (u64)n - 1);
    std::vector<u64> tw(n / 2);
    u64 acc = 1;
    for (size_t i = 0; i < n / 2; i++) { tw[i] = acc; acc = gl_mul(acc, inv_root); }
    if (tw.size() > 1) permute(tw.data(), tw.size());
    return tw;
}

// serial.rs:89-127.  `count` interleaved sub-transforms of `size = len/stride` points each are processed
// together; the recursion first handles the two half-size problems at doubled stride, then applies the
// untwiddled butterflies of block 0 and the twiddled butterflies of blocks 1.. (twiddle index = block).
static void fft_in_place(u64* v, size_t len, const u64* tw, size_t count, size_t stride, size_t offset) {
    size_t size = len / stride;
    if (size > 2) {
        if (stride == count && count < 256) {
            fft_in_place(v, len, tw, 2 * count, 2 * stride, offset);
        } else {
            fft_in_place(v, len, tw, count, 2 * stride, offset);
            fft_in_place(v, len, tw, count, 2 * stride, offset + stride);
        }
    }
    for (size_t o = offset; o < offset + count; o++) {
        u64 a = v[o], b = v[o + stride];
        v[o] = gl_add(a, b);
        v[o + stride] = gl_sub(a, b);
    }
    size_t last = offset + size * stride;
    size_t blk = 1;
    for (size_t o = offset + 2 * stride; o < last; o += 2 * stride, blk++) {
        for (size_t j = o; j < o + count; j++) {
            u64 a = v[j], b = gl_mul(v[j + stride], tw[blk]);
            v[j] = gl_add(a, b);
            v[j + stride] = gl_sub(a, b);
        }
    }
}

void evaluate_poly(u64* p, size_t n) {
    if (n == 1) return;
    std::vector<u64> tw = get_twiddles(n);
    fft_in_place(p, n, tw.data(), 1, 1, 0);
    permute(p, n);
}

std::vector<u64> evaluate_poly_with_offset(const u64* p, size_t n, u64 domain_offset, size_t blowup) {
    size_t domain = n * blowup;
    u64 g = gl_root_of_unity(log2_strict(domain));
    std::vector<u64> tw = get_twiddles(n);
    std::vector<u64> out(domain);
    int bbits = log2_strict(blowup);
    for (size_t i = 0; i < blowup; i++) {
        u64 idx = blowup == 1 ? 0 : reverse_bits(i, bbits);
        u64 off = gl_mul(gl_pow(g, idx), domain_offset);
        u64 f = 1;
        u64* chunk = out.data() + i * n;
        for (size_t k = 0; k < n; k++) { chunk[k] = gl_mul(p[k], f); f = gl_mul(f, off); }
        if (n > 1) fft_in_place(chunk, n, tw.data(), 1, 1, 0);
    }
    if (domain > 1) permute(out.data(), domain);
    return out;
}

void interpolate_poly(u64* e, size_t n) {
    if (n == 1) return;
    std::vector<u64> tw = get_inv_twiddles(n);
    fft_in_place(e, n, tw.data(), 1, 1, 0);
    u64 inv_len = gl_inv((u64)n % GL_P);
    for (size_t i = 0; i < n; i++) e[i] = gl_mul(e[i], inv_len);
    permute(e, n);
}

void interpolate_poly_with_offset(u64* e, size_t n, u64 domain_offset) {
    if (n > 1) {
        std::vector<u64> tw = get_inv_twiddles(n);
        fft_in_place(e, n, tw.data(), 1, 1, 0);
        permute(e, n);
    }
    u64 doi = gl_inv(domain_offset);
    u64 off = gl_inv((u64)n % GL_P);
    for (size_t i = 0; i < n; i++) { e[i] = gl_mul(e[i], off); off = gl_mul(off, doi); }
}

// Naive O(n^2) evaluation at shift * w^i -- the reference's own cross-check recipe (fft.rs:233-252,
// polynomial/mod.rs:494-538).
std::vector<u64> naive_eval(const u64* coeffs, size_t ncoeffs, size_t domain, u64 shift) {
    u64 w = gl_root_of_unity(log2_strict(domain));
    std::vector<u64> out(domain);
    u64 x = shift;
    for (size_t i = 0; i < domain; i++) {
        u64 acc = 0;
        for (size_t k = ncoeffs; k-- > 0;) acc = gl_add(gl_mul(acc, x), coeffs[k]);
        out[i] = acc;
        x = gl_mul(x, w);
    }
    return out;
}

// Extension-field variants: twiddles are base-field (quadratic.rs:61-65 / goldilocks_extensions.rs:27), so an
// extension NTT is two independent base NTTs on the (a, b) planes.
std::vector<Ext2> ext_coset_fft(const std::vector<Ext2>& coeffs, u64 shift) {
    size_t n = coeffs.size();
    std::vector<u64> a(n), b(n);
    for (size_t i = 0; i < n; i++) { a[i] = coeffs[i].a; b[i] = coeffs[i].b; }
    std::vector<u64> ea = evaluate_poly_with_offset(a.data(), n, shift, 1);
    std::vector<u64> eb = evaluate_poly_with_offset(b.data(), n, shift, 1);
    std::vector<Ext2> out(n);
    for (size_t i = 0; i < n; i++) out[i] = Ext2{ea[i], eb[i]};
    return out;
}

}  // namespace ola_oracle
