// gather_probe.hip -- developer micro-benchmark: how many bytes does the fabric move for one random 8-byte gather
// from an 80 MB vector, depending on the load flavour?  (The remainder path of the tiled kernel does 10 M such gathers
// per launch on config 5 and the FETCH_SIZE counter charges ~124 bytes each.)  Not part of the library.
// Build: hipcc -O3 --offload-arch=gfx950 tools/gather_probe.hip -o bin/gather_probe
// Run under rocprofv3 --pmc FETCH_SIZE to get bytes per gather (2 * FETCH_SIZE KB * 1024 / gathers).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                               \
    do {                                                                    \
        hipError_t e = (x);                                                 \
        if (e != hipSuccess) {                                              \
            printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); \
            exit(1);                                                        \
        }                                                                   \
    } while (0)

static inline uint64_t mix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0xBF58476D1CE4E5B9ULL;
    return z ^ (z >> 31);
}

template <int MODE>
__device__ __forceinline__ double load8(const double *p) {
    double v;
    if constexpr (MODE == 0) {
        v = *p;
    } else if constexpr (MODE == 1) {
        v = __builtin_nontemporal_load(p);
    } else if constexpr (MODE == 2) {
        asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    } else if constexpr (MODE == 3) {
        asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    } else if constexpr (MODE == 4) {
        asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    } else if constexpr (MODE == 5) {
        asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1 nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    } else if constexpr (MODE == 6) {
        asm volatile("global_load_dwordx2 %0, %1, off sc0 nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    } else {
        asm volatile("global_load_dwordx2 %0, %1, off sc1 nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    }
    return v;
}

// every thread gathers PER random entries (indices streamed, coalesced)
template <int MODE>
__global__ void __launch_bounds__(256) k_gather(const int *__restrict__ idx, const double *__restrict__ vec, double *__restrict__ out, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int j = __builtin_nontemporal_load(idx + i);
    out[i] = load8<MODE>(vec + j);
}

// scalar-cache path: one wave walks 64 indices with s_load
__global__ void __launch_bounds__(256) k_gather_scalar(const int *__restrict__ idx, const double *__restrict__ vec, double *__restrict__ out, long n) {
    const long w = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const long base = w * 64;
    if (base >= n) return;
    const int mine = (base + lane < n) ? idx[base + lane] : 0;
    double acc = 0.0;
    for (int k = 0; k < 64; ++k) {
        const int j = __builtin_amdgcn_readlane(mine, k);
        const double v = *reinterpret_cast<const double *>(__builtin_assume_aligned(vec + j, 8));  // uniform address -> s_load
        if (lane == k) acc = v;
    }
    if (base + lane < n) out[base + lane] = acc;
}

int main(int argc, char **argv) {
    const long n = argc > 1 ? atol(argv[1]) : 10000000;      // gathers
    const long len = argc > 2 ? atol(argv[2]) : 10000000;    // vector length
    std::vector<int> h(n);
    for (long i = 0; i < n; ++i) h[i] = (int)(mix64(i) % (uint64_t)len);
    int *idx;
    double *vec, *out;
    CK(hipMalloc(&idx, n * 4));
    CK(hipMalloc(&vec, len * 8));
    CK(hipMalloc(&out, n * 8));
    CK(hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemset(vec, 0, len * 8));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const unsigned grid = (unsigned)((n + 255) / 256);
    auto run = [&](const char *name, auto launch) {
        for (int w = 0; w < 2; ++w) launch();
        CK(hipEventRecord(a));
        for (int r = 0; r < 10; ++r) launch();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        printf("%-28s %8.3f us per launch, %6.1f M gathers/ms\n", name, ms * 100.0, n / (ms * 100.0) / 1e3);
    };
    run("plain", [&] { hipLaunchKernelGGL(k_gather<0>, dim3(grid), dim3(256), 0, 0, idx, vec, out, n); });
    run("nontemporal", [&] { hipLaunchKernelGGL(k_gather<1>, dim3(grid), dim3(256), 0, 0, idx, vec, out, n); });
    run("sc0", [&] { hipLaunchKernelGGL(k_gather<2>, dim3(grid), dim3(256), 0, 0, idx, vec, out, n); });
    run("sc1", [&] { hipLaunchKernelGGL(k_gather<3>, dim3(grid), dim3(256), 0, 0, idx, vec, out, n); });
    run("sc0 sc1", [&] { hipLaunchKernelGGL(k_gather<4>, dim3(grid), dim3(256), 0, 0, idx, vec, out, n); });
    run("sc0 sc1 nt", [&] { hipLaunchKernelGGL(k_gather<5>, dim3(grid), dim3(256), 0, 0, idx, vec, out, n); });
    run("sc0 nt", [&] { hipLaunchKernelGGL(k_gather<6>, dim3(grid), dim3(256), 0, 0, idx, vec, out, n); });
    run("sc1 nt", [&] { hipLaunchKernelGGL(k_gather<7>, dim3(grid), dim3(256), 0, 0, idx, vec, out, n); });
    run("scalar", [&] { hipLaunchKernelGGL(k_gather_scalar, dim3(grid), dim3(256), 0, 0, idx, vec, out, n); });
    CK(hipDeviceSynchronize());
    return 0;
}
