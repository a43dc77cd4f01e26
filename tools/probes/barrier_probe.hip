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

// ------------------------------------------------------------------------------------------------
// Round 2: the XCD-hierarchical barrier of MI355X_MICROARCH.md (price list, row barrier-xcd): the workgroups of one XCD
// meet on a counter in THEIR L2's reach, the last arriver of an XCD (its leader for this episode) goes to the top counter,
// the last leader publishes the generation, every leader republishes it to its XCD.  Monotonic counters (episode e is
// complete at members * e), bounded spins, XCC id from the hardware register; the census of workgroups per XCD is taken
// once behind a flat barrier.
// ------------------------------------------------------------------------------------------------
struct XBar {
    unsigned members[8];   // workgroups per XCC (census)
    unsigned pad0[24];
    unsigned top_cnt;      // arrivals of XCC leaders, monotonic
    unsigned pad1[31];
    unsigned top_gen;      // completed episodes
    unsigned pad2[31];
    struct {
        unsigned cnt;      // arrivals on this XCC, monotonic
        unsigned pad[31];
        unsigned gen;      // completed episodes as seen by this XCC
        unsigned pad_[31];
    } x[8];
};

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u; }  // HW_REG_XCC_ID[3:0]

__device__ __forceinline__ bool spin_until_ge(unsigned *w, unsigned e) {
    long spins = 0;
    while (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < e) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 4000000L) return false;
    }
    return true;
}

// episode e = 1, 2, ...; nx = XCCs with members; every wave has drained its stores (s_waitcnt vmcnt(0)) before the call
__device__ __forceinline__ bool xcd_barrier(XBar *b, unsigned e, unsigned x, unsigned nx) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned a = __hip_atomic_fetch_add(&b->x[x].cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        if (a == b->members[x] * e) {  // last arriver of this XCC: its leader for the episode
            const unsigned t = __hip_atomic_fetch_add(&b->top_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
            if (t == nx * e) __hip_atomic_store(&b->top_gen, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else ok = spin_until_ge(&b->top_gen, e);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(&b->x[x].gen, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            ok = spin_until_ge(&b->x[x].gen, e);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    return ok;
}

// census behind one flat barrier (Bar), then `iters` hierarchical barriers; mode 0: barrier only; mode 1: the two half-steps
// of a config-3-sized HPR iteration as data movement: x-half = every workgroup gathers (hashed indices, ~2.3 per entry) from
// the y vector the others published and writes its slice of x, barrier, y-half likewise (~6.8 per row), barrier
__global__ void __launch_bounds__(256) k_xcd(Bar *flat, XBar *b, int iters, int mode, double *xv, double *yv, int N, int M, int *fail) {
    __shared__ unsigned gen0, sx, snx;
    if (threadIdx.x == 0) {
        gen0 = __hip_atomic_load(&flat->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sx = xcc_id();
        __hip_atomic_fetch_add(&b->members[sx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!grid_barrier(flat, gridDim.x, &gen0)) {
        if (threadIdx.x == 0) *fail = 1;
        return;
    }
    if (threadIdx.x == 0) {
        unsigned nx = 0;
        for (int k = 0; k < 8; ++k) nx += __hip_atomic_load(&b->members[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0;
        snx = nx;
    }
    __syncthreads();
    const unsigned x = sx, nx = snx;
    const int G = gridDim.x, w = blockIdx.x;
    const int n0 = (int)((long)N * w / G), n1 = (int)((long)N * (w + 1) / G), m0 = (int)((long)M * w / G), m1 = (int)((long)M * (w + 1) / G);
    unsigned e = 0;
    for (int i = 0; i < iters; ++i) {
        if (mode == 1) {
            for (int j = n0 + threadIdx.x; j < n1; j += 256) {
                unsigned h = (unsigned)j * 2654435761u;
                double s = __builtin_nontemporal_load(yv + (h % (unsigned)M));
                s += __builtin_nontemporal_load(yv + ((h >> 7) % (unsigned)M));
                if (j & 1) s += __builtin_nontemporal_load(yv + ((h >> 13) % (unsigned)M));
                xv[j] = s * 0.25 + 1.0;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (!xcd_barrier(b, ++e, x, nx)) {
            if (threadIdx.x == 0) *fail = 1;
            return;
        }
        if (mode == 1) {
            for (int r = m0 + threadIdx.x; r < m1; r += 256) {
                unsigned h = (unsigned)r * 2246822519u;
                double s = 0.0;
                for (int k = 0; k < 7; ++k) {
                    s += __builtin_nontemporal_load(xv + (h % (unsigned)N));
                    h = h * 1664525u + 1013904223u;
                }
                yv[r] = s * 0.1 + 1.0;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!xcd_barrier(b, ++e, x, nx)) {
                if (threadIdx.x == 0) *fail = 1;
                return;
            }
        }
    }
}

static int run_xcd(int nwg, int iters, int mode, Bar *flat, int *flags, double *xv, double *yv, hipEvent_t e0, hipEvent_t e1) {
    XBar *b;
    CK(hipMalloc(&b, sizeof(XBar)));
    CK(hipMemset(b, 0, sizeof(XBar)));
    CK(hipMemset(flags, 0, 2 * sizeof(int)));
    int it = iters, md = mode, N = 105728, M = 33874;
    int *fail = flags;
    void *args[] = {&flat, &b, &it, &md, &xv, &yv, &N, &M, &fail};
    CK(hipEventRecord(e0));
    CK(hipLaunchCooperativeKernel((void *)k_xcd, dim3(nwg), dim3(256), args, 0, 0));
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    int h[2];
    CK(hipMemcpy(h, flags, sizeof(h), hipMemcpyDeviceToHost));
    unsigned mem[8];
    CK(hipMemcpy(mem, b, sizeof(mem), hipMemcpyDeviceToHost));
    if (mode == 0)
        printf("XCD-hierarchical barrier %4d workgroups: %7.3f us per barrier%s   (members per XCC: %u %u %u %u %u %u %u %u)\n", nwg,
               ms * 1e3 / iters, h[0] ? "  (TIMED OUT)" : "", mem[0], mem[1], mem[2], mem[3], mem[4], mem[5], mem[6], mem[7]);
    else
        printf("config-3-sized iteration as data movement (gather + publish + XCD barrier, twice) %4d workgroups: %7.3f us per iteration%s\n",
               nwg, ms * 1e3 / iters, h[0] ? "  (TIMED OUT)" : "");
    CK(hipFree(b));
    return h[0];
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
    {
        double *xv, *yv;
        CK(hipMalloc(&xv, 105728 * 8));
        CK(hipMalloc(&yv, 33874 * 8));
        CK(hipMemset(xv, 0, 105728 * 8));
        CK(hipMemset(yv, 0, 33874 * 8));
        for (int nwg : {64, 128, 256, 512})
            if (run_xcd(nwg, iters, 0, b, flags, xv, yv, e0, e1)) return 1;
        for (int nwg : {128, 256, 512})
            if (run_xcd(nwg, iters, 1, b, flags, xv, yv, e0, e1)) return 1;
        CK(hipFree(xv));
        CK(hipFree(yv));
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
