// barrier_probe.hip -- developer micro-benchmark: cost of a grid-wide barrier (all workgroups of a cooperative launch)
// with device-scope release/acquire on MI355X, and of a barrier plus a cross-workgroup vector hand-off (each workgroup
// writes a slice, everybody reads the neighbour's slice after the barrier).  Decides whether a persistent multi-workgroup
// kernel can beat one launch per half-step (~4-8 us) for mid-size LPs.  Not part of the library.
// Build: hipcc -O3 --offload-arch=gfx950 tools/barrier_probe.hip -o bin/barrier_probe
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

struct Bar {
    unsigned count;
    unsigned pad0[31];
    unsigned gen;
    unsigned pad1[31];
};

// sense-reversing barrier; bounded spin so that a mistake cannot hang the GPU
__device__ __forceinline__ bool grid_barrier(Bar *b, unsigned nwg, unsigned *local_gen) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        const unsigned g = *local_gen;
        __atomic_thread_fence(__ATOMIC_RELEASE);  // agent scope by default for the HIP memory model
        const unsigned arrived = __hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        if (arrived == nwg) {
            __hip_atomic_store(&b->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&b->gen, g + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            long spins = 0;
            while (__hip_atomic_load(&b->gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == g) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 20000000L) {
                    ok = false;
                    break;
                }
            }
        }
        *local_gen = g + 1u;
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
    return ok;
}

__global__ void __launch_bounds__(256) k_barriers(Bar *b, int iters, int *fail) {
    __shared__ unsigned gen;
    if (threadIdx.x == 0) gen = __hip_atomic_load(&b->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    for (int i = 0; i < iters; ++i)
        if (!grid_barrier(b, gridDim.x, &gen)) {
            if (threadIdx.x == 0) *fail = 1;
            return;
        }
}

// each workgroup writes `slice` doubles, barrier, reads the slice of workgroup (id + shift) % n and checks it
__global__ void __launch_bounds__(256) k_handoff(Bar *b, double *vec, int slice, int iters, int shift, int *fail, int *wrong) {
    __shared__ unsigned gen;
    if (threadIdx.x == 0) gen = __hip_atomic_load(&b->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int n = gridDim.x;
    double acc = 0.0;
    for (int i = 0; i < iters; ++i) {
        for (int k = threadIdx.x; k < slice; k += 256) vec[(size_t)blockIdx.x * slice + k] = i * 1000.0 + blockIdx.x + acc * 0.0;
        if (!grid_barrier(b, n, &gen)) {
            if (threadIdx.x == 0) *fail = 1;
            return;
        }
        const int src = (blockIdx.x + shift) % n;
        for (int k = threadIdx.x; k < slice; k += 256) {
            const double v = vec[(size_t)src * slice + k];
            if (v != i * 1000.0 + src) atomicAdd(wrong, 1);
            acc += v;
        }
        if (!grid_barrier(b, n, &gen)) {  // nobody overwrites before everybody has read
            if (threadIdx.x == 0) *fail = 1;
            return;
        }
    }
    if (acc == -1.0) vec[0] = acc;
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    Bar *b;
    int *flags;
    double *vec;
    CK(hipMalloc(&b, sizeof(Bar)));
    CK(hipMemset(b, 0, sizeof(Bar)));
    CK(hipMalloc(&flags, 2 * sizeof(int)));
    CK(hipMemset(flags, 0, 2 * sizeof(int)));
    const int max_slice = 4096;
    CK(hipMalloc(&vec, (size_t)1024 * max_slice * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int nwg : {64, 128, 256, 512}) {
        int it = iters;
        int *fail = flags;
        void *args[] = {&b, &it, &fail};
        CK(hipEventRecord(e0));
        CK(hipLaunchCooperativeKernel((void *)k_barriers, dim3(nwg), dim3(256), args, 0, 0));
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        int h[2];
        CK(hipMemcpy(h, flags, sizeof(h), hipMemcpyDeviceToHost));
        printf("barrier only      %4d workgroups: %7.3f us per barrier%s\n", nwg, ms * 1e3 / iters, h[0] ? "  (TIMED OUT)" : "");
        if (h[0]) return 1;
    }
    for (int nwg : {128, 256}) {
        for (int slice : {128, 512, 4096}) {
            int it = iters, sl = slice, shift = nwg / 2 + 1;  // a workgroup on another XCD
            int *fail = flags, *wrong = flags + 1;
            void *args[] = {&b, &vec, &sl, &it, &shift, &fail, &wrong};
            CK(hipEventRecord(e0));
            CK(hipLaunchCooperativeKernel((void *)k_handoff, dim3(nwg), dim3(256), args, 0, 0));
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            int h[2];
            CK(hipMemcpy(h, flags, sizeof(h), hipMemcpyDeviceToHost));
            printf("write+bar+read+bar %4d workgroups, %5d doubles each: %7.3f us per round, %d stale reads%s\n", nwg, slice,
                   ms * 1e3 / iters, h[1], h[0] ? "  (TIMED OUT)" : "");
            if (h[0]) return 1;
            CK(hipMemset(flags, 0, 2 * sizeof(int)));
        }
    }
    return 0;
}
