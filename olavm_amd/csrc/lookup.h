// Device-side generation of lookup-argument columns (lookup.hip, its own translation unit).
#pragma once
#include <cstddef>
#include <cstdint>

#include "device_ctx.h"
#include "gl.cuh"

namespace ola {

// permuted_cols (circuits/src/stark/lookup.rs:68-132): all four pointers are device memory of n words; inputs / table may
// hold non-canonical words.  permuted_inputs = the inputs sorted (canonical), permuted_table as the reference builds it.
void permuted_cols_dev(DeviceCtx* ctx, const u64* inputs, const u64* table, size_t n, u64* permuted_inputs, u64* permuted_table);

}  // namespace ola
