// mlp_probe.hip -- how much memory-level parallelism one CU can keep: streaming reads with K independent 16-byte loads in flight per
// lane, W waves per workgroup, G workgroups per CU; every wave streams its own contiguous range (the pattern of the tiled kernels).
// Reports GB/s chip-wide: where it saturates in K x W x G is the per-CU cap on outstanding requests.
// Build: hipcc -O3 --offload-arch=gfx950 tools/mlp_probe.hip -o bin/mlp_probe
#include <hip/hip_runtime.h>

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

typedef double d2_t __attribute__((ext_vector_type(2)));

template <int K, bool NT>
__global__ void k_stream(const d2_t *__restrict__ src, size_t per_wave, double *out) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const d2_t *p = src + static_cast<size_t>(wave) * per_wave + lane;  // per_wave: 16-byte elements, multiple of 64 * K
    double s = 0.0;
    for (size_t i = 0; i < per_wave; i += 64 * K) {
        d2_t v[K];
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = NT ? __builtin_nontemporal_load(p + i + 64 * k) : p[i + 64 * k];
#pragma unroll
        for (int k = 0; k < K; ++k) s += v[k].x + v[k].y;
    }
    if (s == 1.2345e-300) out[0] = s;
}

template <int K>
static void run(const d2_t *buf, size_t total16, int threads, int blocks_per_cu, double *out) {
    const int waves = 256 * blocks_per_cu * threads / 64;
    size_t per_wave = total16 / waves / (64 * K) * (64 * K);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_stream<K, true>), dim3(256 * blocks_per_cu), dim3(threads), 0, 0, buf, per_wave, out);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    const double bytes = static_cast<double>(per_wave) * waves * 16.0;
    printf("K=%2d loads/lane  %4d threads x %d WG/CU (%2d waves/CU, %5.1f KB issued in flight per CU)  %8.1f us  %7.0f GB/s\n", K, threads,
           blocks_per_cu, threads / 64 * blocks_per_cu, K * 1.0 * threads * blocks_per_cu * 16 / 1024.0, best * 1e3, bytes / best / 1e6);
    fflush(stdout);
}

int main() {
    const size_t bytes = size_t(2) << 30;  // 2 GiB: beyond the Infinity Cache
    d2_t *buf;
    double *out;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(&out, 8));
    CK(hipMemset(buf, 0, bytes));
    const size_t total16 = bytes / 16;
    for (int g : {1, 2})
        for (int threads : {64, 256, 512, 1024}) {
            if (threads * g > 2048) continue;
            run<1>(buf, total16, threads, g, out);
            run<2>(buf, total16, threads, g, out);
            run<4>(buf, total16, threads, g, out);
            run<8>(buf, total16, threads, g, out);
            run<16>(buf, total16, threads, g, out);
        }
    return 0;
}
