// tiled_probe.hip -- developer prototype of the column-tiled SpMV (vector window staged in LDS).
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/tiled_probe.hip -o bin/tiled_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
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
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

static void gen(int m, int n, int per_row, int band, int r0, int r1, int *col, double *val) {
    std::vector<int> c(per_row);
    const int width = std::min(2 * band + 1, n);
    for (int r = r0; r < r1; ++r) {
        uint64_t st = mix64(0x1234 + r);
        long center = (long)r * n / m, base = std::max(0L, std::min<long>(center - band, n - width));
        for (int k = 0; k < per_row; ++k) {
            st = mix64(st);
            double u = (st >> 11) * (1.0 / 9007199254740992.0);
            st = mix64(st);
            double w = (st >> 11) * (1.0 / 9007199254740992.0);
            c[k] = (u < 0.05) ? (int)(w * n) : (int)(base + (long)(w * width));
        }
        std::sort(c.begin(), c.end());
        for (int k = 1; k < per_row; ++k)
            if (c[k] <= c[k - 1]) c[k] = c[k - 1] + 1;
        for (int k = 0; k < per_row; ++k) {
            col[(size_t)r * per_row + k] = std::min(c[k], n - 1);
            val[(size_t)r * per_row + k] = 1.0 + 1e-3 * ((r + k) % 7);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// tiled format
// ---------------------------------------------------------------------------------------------
struct Step {
    int col0;     // first column of the tile (tile steps)
    int e_begin;  // entry range in tval/tidx (tile step) or rval/rcol/rrow (remainder step)
    int e_end;
    int kind;     // bit0: 1 = remainder step (global gather); bit1: 1 = stage a new tile first
};

struct Tiled {
    std::vector<int> sb_ptr;  // steps of super-block sb: [sb_ptr[sb], sb_ptr[sb+1])
    std::vector<Step> steps;
    std::vector<double> tval;
    std::vector<uint32_t> tidx;  // lcol << 16 | lrow
    std::vector<double> rval;
    std::vector<int> rcol;
    std::vector<uint16_t> rrow;
};

static Tiled build_tiled(int m, int n, const int *rp, const int *col, const double *val, int R, int T, int E,
                         int dense_min) {
    Tiled t;
    const int nsb = (m + R - 1) / R, ntile = (n + T - 1) / T;
    t.sb_ptr.assign(nsb + 1, 0);
    std::vector<int> cnt(ntile), start(ntile);
    std::vector<int> touched;
    size_t rem_base = 0;
    for (int sb = 0; sb < nsb; ++sb) {
        const int r0 = sb * R, r1 = std::min(m, r0 + R);
        touched.clear();
        for (int k = rp[r0]; k < rp[r1]; ++k) {
            const int tl = col[k] / T;
            if (cnt[tl]++ == 0) touched.push_back(tl);
        }
        std::sort(touched.begin(), touched.end());
        // dense tiles: bucket entries (row-major scan keeps (row,col) order inside each tile)
        size_t base = t.tval.size();
        size_t total_dense = 0;
        for (int tl : touched)
            if (cnt[tl] >= dense_min) {
                start[tl] = (int)(base + total_dense);
                total_dense += cnt[tl];
            }
        t.tval.resize(base + total_dense);
        t.tidx.resize(base + total_dense);
        {
            // position lookup via start[] (advanced while filling)
            for (int r = r0; r < r1; ++r)
                for (int k = rp[r]; k < rp[r + 1]; ++k) {
                    const int tl = col[k] / T;
                    if (cnt[tl] >= dense_min) {
                        const int p = start[tl]++;
                        t.tval[p] = val[k];
                        t.tidx[p] = ((uint32_t)(col[k] - tl * T) << 16) | (uint32_t)(r - r0);
                    } else {
                        t.rval.push_back(val[k]);
                        t.rcol.push_back(col[k]);
                        t.rrow.push_back((uint16_t)(r - r0));
                    }
                }
        }
        // steps: dense tiles in ascending column order, split into chunks of E entries
        size_t p = base;
        for (int tl : touched)
            if (cnt[tl] >= dense_min) {
                int left = cnt[tl];
                bool first = true;
                while (left > 0) {
                    const int c = std::min(left, E);
                    t.steps.push_back(Step{tl * T, (int)p, (int)(p + c), first ? 2 : 0});
                    p += c;
                    left -= c;
                    first = false;
                }
            }
        // remainder steps
        {
            size_t rb = rem_base, re = t.rval.size();
            while (rb < re) {
                const size_t c = std::min<size_t>(re - rb, E);
                t.steps.push_back(Step{0, (int)rb, (int)(rb + c), 1});
                rb += c;
            }
            rem_base = re;
        }
        for (int tl : touched) cnt[tl] = 0;
        t.sb_ptr[sb + 1] = (int)t.steps.size();
    }
    return t;
}

// ---------------------------------------------------------------------------------------------
// kernel
// ---------------------------------------------------------------------------------------------
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt(0), which would
// wait for the global loads prefetched for the NEXT step and expose their full latency every step.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int NT, int R, int T, int K>
__global__ void __launch_bounds__(NT) k_tiled(const int *__restrict__ sb_ptr, const Step *__restrict__ steps,
                                              const double *__restrict__ tval, const uint32_t *__restrict__ tidx,
                                              const double *__restrict__ rval, const int *__restrict__ rcol,
                                              const uint16_t *__restrict__ rrow, const double *__restrict__ vec,
                                              double *__restrict__ out, int m, int n) {
    constexpr int E = NT * K;
    constexpr int TPT = T / NT;  // tile doubles per thread
    __shared__ double acc[R];
    __shared__ double ytile[T];
    __shared__ double prod[E];
    __shared__ uint16_t rows[E + 2];
    const int tid = threadIdx.x;
    const int sb = blockIdx.x;
    for (int i = tid; i < R; i += NT) acc[i] = 0.0;
    const int s0 = sb_ptr[sb], s1 = sb_ptr[sb + 1];
    if (s0 >= s1) {
        __syncthreads();
    }
    // registers holding the prefetched step
    Step nst = (s0 < s1) ? steps[s0] : Step{0, 0, 0, 0};
    double nv[K];
    uint32_t ni[K];
    uint16_t nr[K];
    double nt_[TPT];
    auto prefetch = [&](const Step &st) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int e = st.e_begin + tid + k * NT;
            const bool ok = e < st.e_end;
            const int ee = ok ? e : st.e_begin;
            if (st.kind & 1) {
                nv[k] = __builtin_nontemporal_load(rval + ee);
                ni[k] = (uint32_t)__builtin_nontemporal_load(rcol + ee);
                nr[k] = __builtin_nontemporal_load(rrow + ee);
            } else {
                nv[k] = __builtin_nontemporal_load(tval + ee);
                ni[k] = __builtin_nontemporal_load(tidx + ee);
                nr[k] = 0;
            }
        }
        if (st.kind & 2) {
#pragma unroll
            for (int j = 0; j < TPT; ++j) {
                const int c = st.col0 + tid + j * NT;
                nt_[j] = (c < n) ? vec[c] : 0.0;
            }
        }
    };
    if (s0 < s1) prefetch(nst);
    for (int s = s0; s < s1; ++s) {
        const Step st = nst;
        double cv[K];
        uint32_t ci[K];
        uint16_t cr[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            cv[k] = nv[k];
            ci[k] = ni[k];
            cr[k] = nr[k];
        }
        if (st.kind & 2) {
#pragma unroll
            for (int j = 0; j < TPT; ++j) ytile[tid + j * NT] = nt_[j];
        }
        if (s + 1 < s1) {
            nst = steps[s + 1];
            prefetch(nst);
        }
        lds_barrier();  // tile visible; previous step's head sums done
        const int cnt = st.e_end - st.e_begin;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int el = tid + k * NT;
            if (el < cnt) {
                double g;
                uint16_t rw;
                if (st.kind & 1) {
                    g = vec[ci[k]];
                    rw = cr[k];
                } else {
                    g = ytile[ci[k] >> 16];
                    rw = (uint16_t)(ci[k] & 0xffffu);
                }
                prod[el] = cv[k] * g;
                rows[el + 1] = rw;
            }
        }
        if (tid == 0) rows[0] = 0xffffu;  // sentinel: entry 0 is always a head
        lds_barrier();
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int el = tid + k * NT;
            if (el < cnt) {
                const uint16_t rw = rows[el + 1];
                if (rows[el] != rw) {  // head of a row segment
                    double sacc = acc[rw];
                    int j = el;
                    do {
                        sacc += prod[j];
                        ++j;
                    } while (j < cnt && rows[j + 1] == rw);
                    acc[rw] = sacc;
                }
            }
        }
    }
    __syncthreads();
    const int r0 = sb * R;
    for (int i = tid; i < R && r0 + i < m; i += NT) out[r0 + i] = acc[i];
}

int main(int argc, char **argv) {
    const int m = argc > 1 ? atoi(argv[1]) : 10000000, n = m, per_row = 20;
    const int band = argc > 2 ? atoi(argv[2]) : 100000;
    const size_t nnz = (size_t)m * per_row;
    std::vector<int> rp(m + 1), col(nnz);
    std::vector<double> val(nnz);
    for (int i = 0; i <= m; ++i) rp[i] = i * per_row;
    {
        int nt = std::max(1u, std::thread::hardware_concurrency());
        std::vector<std::thread> th;
        int chunk = (m + nt - 1) / nt;
        for (int t = 0; t < nt; ++t) {
            int a = t * chunk, b = std::min(m, a + chunk);
            if (a < b) th.emplace_back(gen, m, n, per_row, band, a, b, col.data(), val.data());
        }
        for (auto &t : th) t.join();
    }
    std::vector<double> vec(n);
    for (int i = 0; i < n; ++i) vec[i] = 1.0 + (i % 13) * 0.01;
    double *d_vec, *d_out;
    CK(hipMalloc(&d_vec, (size_t)n * 8));
    CK(hipMalloc(&d_out, (size_t)m * 8));
    CK(hipMemcpy(d_vec, vec.data(), (size_t)n * 8, hipMemcpyHostToDevice));
    std::vector<double> ref(m), got(m);
    for (int i = 0; i < m; i += 997) {
        double s = 0;
        for (int k = rp[i]; k < rp[i + 1]; ++k) s += val[k] * vec[col[k]];
        ref[i] = s;
    }
    const double bytes = 12.0 * nnz + 4.0 * (m + 1) + 8.0 * n + 8.0 * m;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));

    auto run_cfg = [&](const char *name, auto kern, int NT, int R, int T, int K, int dense_min) {
        Tiled t = build_tiled(m, n, rp.data(), col.data(), val.data(), R, T, NT * K, dense_min);
        const int nsb = (int)t.sb_ptr.size() - 1;
        int *d_sb, *d_rcol;
        Step *d_steps;
        double *d_tval, *d_rval;
        uint32_t *d_tidx;
        uint16_t *d_rrow;
        CK(hipMalloc(&d_sb, t.sb_ptr.size() * 4));
        CK(hipMalloc(&d_steps, std::max<size_t>(1, t.steps.size()) * sizeof(Step)));
        CK(hipMalloc(&d_tval, (t.tval.size() + 8) * 8));
        CK(hipMalloc(&d_tidx, (t.tidx.size() + 8) * 4));
        CK(hipMalloc(&d_rval, (t.rval.size() + 8) * 8));
        CK(hipMalloc(&d_rcol, (t.rcol.size() + 8) * 4));
        CK(hipMalloc(&d_rrow, (t.rrow.size() + 8) * 2));
        CK(hipMemcpy(d_sb, t.sb_ptr.data(), t.sb_ptr.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_steps, t.steps.data(), t.steps.size() * sizeof(Step), hipMemcpyHostToDevice));
        CK(hipMemcpy(d_tval, t.tval.data(), t.tval.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_tidx, t.tidx.data(), t.tidx.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_rval, t.rval.data(), t.rval.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_rcol, t.rcol.data(), t.rcol.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_rrow, t.rrow.data(), t.rrow.size() * 2, hipMemcpyHostToDevice));
        auto launch = [&]() {
            hipLaunchKernelGGL(kern, dim3(nsb), dim3(NT), 0, 0, d_sb, d_steps, d_tval, d_tidx, d_rval, d_rcol, d_rrow, d_vec,
                               d_out, m, n);
        };
        for (int i = 0; i < 3; ++i) launch();
        CK(hipDeviceSynchronize());
        const int reps = 20;
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        CK(hipMemcpy(got.data(), d_out, (size_t)m * 8, hipMemcpyDeviceToHost));
        double maxerr = 0;
        for (int i = 0; i < m; i += 997) maxerr = std::max(maxerr, std::fabs(got[i] - ref[i]));
        printf("%-34s %8.3f ms %8.1f GB/s  steps %zu (%.1f/sb)  dense %.1f%%  maxerr %.3g\n", name, ms, bytes / ms * 1e-6,
               t.steps.size(), (double)t.steps.size() / nsb, 100.0 * t.tval.size() / nnz, maxerr);
        hipFree(d_sb); hipFree(d_steps); hipFree(d_tval); hipFree(d_tidx); hipFree(d_rval); hipFree(d_rcol); hipFree(d_rrow);
    };
    printf("m=n=%d nnz=%zu band=%d bytes=%.3f GB\n", m, nnz, band, bytes * 1e-9);
    run_cfg("NT512 R2048 T4096 K4", k_tiled<512, 2048, 4096, 4>, 512, 2048, 4096, 4, 256);
    run_cfg("NT512 R4096 T4096 K4", k_tiled<512, 4096, 4096, 4>, 512, 4096, 4096, 4, 256);
    run_cfg("NT256 R2048 T2048 K4", k_tiled<256, 2048, 2048, 4>, 256, 2048, 2048, 4, 128);
    run_cfg("NT1024 R8192 T4096 K4", k_tiled<1024, 8192, 4096, 4>, 1024, 8192, 4096, 4, 256);
    run_cfg("NT512 R4096 T8192 K4", k_tiled<512, 4096, 8192, 4>, 512, 4096, 8192, 4, 512);
    return 0;
}
