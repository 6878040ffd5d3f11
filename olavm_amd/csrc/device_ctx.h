// Device context shared by all kernels of the backend: one HIP device, one compute stream, a persistent pool for
// tables, and an error slot the C ABI reports through ola_gpu_last_error().  No exceptions cross the C ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <stdexcept>
#include <string>
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

struct DeviceCtx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    std::vector<void*> persistent;

    void* alloc_persistent(size_t bytes) {
        void* p = nullptr;
        HIP_CHECK(hipMalloc(&p, bytes ? bytes : 8));
        persistent.push_back(p);
        return p;
    }
    void* alloc(size_t bytes) {
        void* p = nullptr;
        HIP_CHECK(hipMalloc(&p, bytes ? bytes : 8));
        return p;
    }
    void free(void* p) {
        if (p) (void)hipFree(p);
    }
    ~DeviceCtx() {
        for (void* p : persistent) (void)hipFree(p);
        if (owns_stream && stream) (void)hipStreamDestroy(stream);
    }
};

}  // namespace ola
