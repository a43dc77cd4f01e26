// stream_probe.hip -- what the memory system of the box delivers to plain streaming kernels with the read : write mixes
// of the half-step kernels (no gathers, no arithmetic to speak of): the practical ceiling the roofline fractions are read
// against.  Build: hipcc -O3 --offload-arch=gfx950 tools/stream_probe.hip -o build/stream_probe ; run: build/stream_probe [MB per array]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));          \
            std::exit(1);                                                         \
        }                                                                         \
    } while (0)

constexpr int kMax = 8;
typedef double dbl2 __attribute__((ext_vector_type(2)));
struct Ptrs {
    const dbl2 *r[kMax];
    dbl2 *w[kMax];
};

// every lane reads one double2 from each of R arrays and writes one to each of W arrays, grid-stride
template <int R, int W>
__global__ void __launch_bounds__(256) k_mix(Ptrs p, size_t n2, double *sink) {
    double acc = 0.0;
    for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < n2; i += static_cast<size_t>(gridDim.x) * 256) {
        dbl2 v[R > 0 ? R : 1];
#pragma unroll
        for (int a = 0; a < R; ++a) v[a] = __builtin_nontemporal_load(p.r[a] + i);
        dbl2 s = {0.0, 1.0};
#pragma unroll
        for (int a = 0; a < R; ++a) {
            s.x += v[a].x;
            s.y += v[a].y;
        }
        if (W == 0) acc += s.x + s.y;
#pragma unroll
        for (int a = 0; a < W; ++a) __builtin_nontemporal_store(s, p.w[a] + i);
    }
    if (W == 0 && acc == 12345.678) *sink = acc;
}

// the persistent-workgroup pattern of the tiled kernels: every workgroup streams its OWN contiguous range of each array
// (512 lanes x 16 B = 8 KB per array per trip), so the chip reads gridDim.x * (R + W) scattered streams at once
template <int R, int W>
__global__ void __launch_bounds__(512) k_private(Ptrs p, size_t n2, double *sink) {
    const size_t per = (n2 + gridDim.x - 1) / gridDim.x;
    const size_t b = blockIdx.x * per, e = b + per < n2 ? b + per : n2;
    double acc = 0.0;
    for (size_t i = b + threadIdx.x; i < e; i += 512) {
        dbl2 v[R > 0 ? R : 1];
#pragma unroll
        for (int a = 0; a < R; ++a) v[a] = __builtin_nontemporal_load(p.r[a] + i);
        dbl2 s = {0.0, 1.0};
#pragma unroll
        for (int a = 0; a < R; ++a) {
            s.x += v[a].x;
            s.y += v[a].y;
        }
        if (W == 0) acc += s.x + s.y;
#pragma unroll
        for (int a = 0; a < W; ++a) __builtin_nontemporal_store(s, p.w[a] + i);
    }
    if (W == 0 && acc == 12345.678) *sink = acc;
}

template <int R, int W>
void run_private(const char *name, Ptrs p, size_t n2, double *sink, int grid) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_private<R, W>), dim3(grid), dim3(512), 0, 0, p, n2, sink);
    const int reps = 20;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_private<R, W>), dim3(grid), dim3(512), 0, 0, p, n2, sink);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = static_cast<double>(n2) * 16.0 * (R + W);
    std::printf("%-10s private ranges, %4d workgroups of 512  %8.1f us  %7.0f GB/s\n", name, grid, ms / reps * 1e3,
                bytes / (ms / reps * 1e-3) / 1e9);
}

template <int R, int W>
void run(const char *name, Ptrs p, size_t n2, double *sink, int grid) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_mix<R, W>), dim3(grid), dim3(256), 0, 0, p, n2, sink);
    const int reps = 20;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_mix<R, W>), dim3(grid), dim3(256), 0, 0, p, n2, sink);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = static_cast<double>(n2) * 16.0 * (R + W);
    std::printf("%-10s grid %6d  %8.1f us  %7.0f GB/s\n", name, grid, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e9);
}

int main(int argc, char **argv) {
    const size_t mb = argc > 1 ? std::atol(argv[1]) : 80;  // per array (config 5: 80 MB per vector)
    const size_t n2 = mb * 1000 * 1000 / 16;
    Ptrs p;
    for (int a = 0; a < kMax; ++a) {
        dbl2 *x;
        CK(hipMalloc(&x, n2 * 16));
        CK(hipMemset(x, 0, n2 * 16));
        p.r[a] = x;
        CK(hipMalloc(&x, n2 * 16));
        CK(hipMemset(x, 0, n2 * 16));
        p.w[a] = x;
    }
    double *sink;
    CK(hipMalloc(&sink, 8));
    for (int grid : {2048, 8192, 65536}) {
        run<1, 0>("1r", p, n2, sink, grid);
        run<4, 0>("4r", p, n2, sink, grid);
        run<8, 0>("8r", p, n2, sink, grid);
        run<1, 1>("1r1w", p, n2, sink, grid);
        run<5, 2>("5r2w", p, n2, sink, grid);
        run<4, 1>("4r1w", p, n2, sink, grid);
        run<0, 2>("2w", p, n2, sink, grid);
    }
    for (int grid : {512, 1024}) {
        run_private<1, 0>("1r", p, n2, sink, grid);
        run_private<2, 0>("2r", p, n2, sink, grid);
        run_private<4, 0>("4r", p, n2, sink, grid);
        run_private<5, 2>("5r2w", p, n2, sink, grid);
    }
    return 0;
}
