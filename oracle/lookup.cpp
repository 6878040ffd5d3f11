// ORACLE -- test infrastructure only (see gl.hpp header).  Sequential restatement of the lookup argument's column
// generator, circuits/src/stark/lookup.rs:68-132 (`permuted_cols`): sort the canonical inputs and the canonical table,
// walk both with one merge loop that keeps the not-yet-used table values on a stack, hand a stacked value to every
// repeated input (or remember the slot when the stack is empty), and finally pair the remembered / left-over slots with
// what is left on the stack, bottom first.
#include <algorithm>
#include <vector>

#include "oracle.hpp"

using namespace ola_oracle;

extern "C" void oracle_permuted_cols(const u64* inputs, const u64* table, size_t n, u64* permuted_inputs, u64* permuted_table) {
    std::vector<u64> si(n), st(n);
    for (size_t k = 0; k < n; k++) { si[k] = gl_canon(inputs[k]); st[k] = gl_canon(table[k]); }   // lookup.rs:80-89
    std::sort(si.begin(), si.end());
    std::sort(st.begin(), st.end());
    std::vector<size_t> unused_inds;
    std::vector<u64> unused_vals;
    std::vector<u64> pt(n, 0);
    size_t i = 0, j = 0;
    while (j < n && i < n) {                                                                // :96-117
        const u64 a = si[i], b = st[j];
        if (a > b) {
            unused_vals.push_back(b);
            j++;
        } else if (a < b) {
            if (!unused_vals.empty()) { pt[i] = unused_vals.back(); unused_vals.pop_back(); }
            else unused_inds.push_back(i);
            i++;
        } else {
            pt[i] = b;
            i++;
            j++;
        }
    }
    for (; j < n; j++) unused_vals.push_back(st[j]);                                        // :120-122
    for (; i < n; i++) unused_inds.push_back(i);                                            // :123-125
    for (size_t k = 0; k < unused_inds.size() && k < unused_vals.size(); k++) pt[unused_inds[k]] = unused_vals[k];   // :126-128
    for (size_t k = 0; k < n; k++) { permuted_inputs[k] = si[k]; permuted_table[k] = pt[k]; }
}
