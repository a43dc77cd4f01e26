// common.h -- private helpers shared by the host side of the MI355X HPR-LP library.
#pragma once

#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "structs.h"

namespace hprlp {

// Every HIP failure becomes an exception that is caught at the extern "C" boundary and turned into
// status "ERROR" / NULL (the reference lets it escape: include/cuda_kernels/cuda_check.h:56-63).
#define HIP_CHECK(expr)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            throw std::runtime_error(std::string("HIP error ") + hipGetErrorString(e_) + " at " + \
                                     __FILE__ + ":" + std::to_string(__LINE__) + " in " #expr);  \
        }                                                                                        \
    } while (0)

void set_last_error(const std::string &msg);
const char *last_error_cstr();

using clock_type = std::chrono::steady_clock;
inline clock_type::time_point time_now() { return clock_type::now(); }
inline double time_since(clock_type::time_point t0) {
    return std::chrono::duration<double>(clock_type::now() - t0).count();
}

// HPRLP_TIMING=1: where the allocator's time goes (hipMalloc / hipFree of multi-GB set-up temporaries are not free)
struct AllocStats {
    double malloc_s = 0, free_s = 0;
    long mallocs = 0, frees = 0;
    static AllocStats &get() {
        static AllocStats s;
        return s;
    }
};

// Process-wide cache of large freed device blocks (host_model.cpp).  hipMalloc of a multi-GB set-up temporary right after the
// previous solver's buffers were freed stalls for 0.5-0.9 s now and then on MI355X / ROCm 7.2 (measured with HPRLP_TIMING=1: one
// hipMalloc call, a third of config 5's time-to-tolerance); blocks of 1 MiB and more therefore go back to this cache instead of
// the driver and are handed out again to requests of the same size class (rounded up to 2 MiB).  Keyed by the device that OWNS
// the block (not the freeing thread's current device), mutex-protected, capped PER DEVICE at min(kDeviceCacheCapBytes, a third of
// the device's memory); a hipMalloc that runs out of memory trims the cache and retries once (device_malloc_or_trim);
// HPRLP_NO_ALLOC_CACHE=1 switches it off, hprlp_release_device_cache() returns everything.
constexpr size_t kDeviceCacheMinBytes = size_t(1) << 20;
constexpr size_t kDeviceCacheCapBytes = size_t(96) << 30;
void *device_cache_get(size_t bytes, size_t *capacity);  // nullptr: nothing suitable cached
bool device_cache_put(void *p, size_t capacity);        // false: not cached (caller frees)
void device_cache_trim();
void *device_malloc_or_trim(size_t bytes);              // hipMalloc; on out-of-memory: trim the cache, retry once, then throw

// Device buffer with RAII; sized in elements.
template <class T>
struct DBuf {
    T *p = nullptr;
    size_t n = 0;
    DBuf() = default;
    explicit DBuf(size_t count) { alloc(count); }
    DBuf(const DBuf &) = delete;
    DBuf &operator=(const DBuf &) = delete;
    DBuf(DBuf &&o) noexcept : p(o.p), n(o.n), cap_bytes(o.cap_bytes) { o.p = nullptr; o.n = 0; o.cap_bytes = 0; }
    DBuf &operator=(DBuf &&o) noexcept {
        if (this != &o) { release(); p = o.p; n = o.n; cap_bytes = o.cap_bytes; o.p = nullptr; o.n = 0; o.cap_bytes = 0; }
        return *this;
    }
    ~DBuf() { release(); }
    size_t cap_bytes = 0;  // size of the underlying block (>= n * sizeof(T); blocks from the cache are rounded up)
    void alloc(size_t count) {
        release();
        n = count;
        size_t bytes = (count ? count : 1) * sizeof(T);
        if (bytes >= kDeviceCacheMinBytes) {
            bytes = (bytes + (size_t(2) << 20) - 1) / (size_t(2) << 20) * (size_t(2) << 20);
            if (void *q = device_cache_get(bytes, &cap_bytes)) {
                p = static_cast<T *>(q);
                return;
            }
        }
        const auto t0 = time_now();
        p = static_cast<T *>(device_malloc_or_trim(bytes));
        cap_bytes = bytes;
        AllocStats::get().malloc_s += time_since(t0);
        ++AllocStats::get().mallocs;
    }
    void alloc_zero(size_t count) {
        alloc(count);
        HIP_CHECK(hipMemset(p, 0, (count ? count : 1) * sizeof(T)));
    }
    void release() {
        if (p) {
            if (!(cap_bytes >= kDeviceCacheMinBytes && device_cache_put(p, cap_bytes))) {
                const auto t0 = time_now();
                (void)hipFree(p);
                AllocStats::get().free_s += time_since(t0);
                ++AllocStats::get().frees;
            }
        }
        p = nullptr;
        n = 0;
        cap_bytes = 0;
    }
    void upload(const T *src, size_t count) {
        if (count) HIP_CHECK(hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice));
    }
    void download(T *dst, size_t count) const {
        if (count) HIP_CHECK(hipMemcpy(dst, p, count * sizeof(T), hipMemcpyDeviceToHost));
    }
    operator T *() const { return p; }
};

// Pinned host buffer.
template <class T>
struct HBuf {
    T *p = nullptr;
    size_t n = 0;
    HBuf() = default;
    HBuf(const HBuf &) = delete;
    HBuf &operator=(const HBuf &) = delete;
    ~HBuf() { if (p) (void)hipHostFree(p); }
    void alloc(size_t count) {
        n = count;
        HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&p), (count ? count : 1) * sizeof(T), hipHostMallocDefault));
        std::memset(p, 0, (count ? count : 1) * sizeof(T));
    }
    T &operator[](size_t i) { return p[i]; }
};

// Host-side CSR transpose, stable in row order (reference src/utils.cu:203-232).
void csr_transpose_host(int rows, int cols, long nnz, const int *rp, const int *ci, const double *v,
                        std::vector<int> &trp, std::vector<int> &tci, std::vector<double> &tv);
// rows [c0, c1) of the transpose only
void csr_transpose_range_host(int rows, int c0, int c1, const int *rp, const int *ci, const double *v,
                              std::vector<int> &trp, std::vector<int> &tci, std::vector<double> &tv);

// Deterministic start vector for the power iteration (spec in oracle/hpr_oracle.c; this is the
// product's own implementation of the same counter RNG).
void power_start_vector(int m, unsigned long long seed, long long offset, double *z);

// Print cadence of the iteration log (reference src/utils.cu:100-102).
int log_step(int iter);

void free_lp_info_cpu_members(LP_info_cpu *model);

// once per process, on the calling thread: HIP runtime, device 0's context, the code objects every solve needs (abi.cpp)
void warm_for_first_solve();

}  // namespace hprlp
