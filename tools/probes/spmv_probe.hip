// spmv_probe.hip -- developer micro-benchmark: which part of the fused CSR kernel bounds it on MI355X?
// Runs variants of the stream-mode SpMV core on the banded benchmark matrix and prints time / GB/s.
// Not part of the library.  Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/spmv_probe.hip -o bin/spmv_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#define CK(x)                                                                   \
    do {                                                                        \
        hipError_t e = (x);                                                     \
        if (e != hipSuccess) {                                                  \
            printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__);     \
            exit(1);                                                            \
        }                                                                       \
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
        for (int k = 0; k < per_row; ++k) {
            col[(size_t)r * per_row + k] = std::min(c[k], n - 1);
            val[(size_t)r * per_row + k] = 1.0 + 1e-3 * (k % 7);
        }
    }
}

constexpr int W = 512;

__device__ __forceinline__ void wsync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// MODE 0: full (stream + gather + LDS + row sums + store)
// MODE 1: no gather (multiply by constant)
// MODE 2: gather but index = local sequential (k % n)
// MODE 3: no LDS phase: every lane keeps a private sum (wrong result; measures load side only)
// MODE 4: full with nontemporal matrix loads
template <int MODE>
__global__ void __launch_bounds__(256) k_stream(const int4 *blk, int nblk, const int *rowptr, const int *col,
                                                const double *val, const double *vec, double *out, int n) {
    __shared__ double lds[4][W];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + wave;
    if (b >= nblk) return;
    const int4 d = blk[b];
    const int r0 = d.x, nr = d.y, k0 = d.z, nz = d.w;
    const int *cp = col + k0;
    const double *vp = val + k0;
    int rs = 0, re = 0;
    if (lane < nr) {
        rs = rowptr[r0 + lane] - k0;
        re = rowptr[r0 + lane + 1] - k0;
    }
    double priv = 0.0;
    const int last = nz - 1;
    for (int base = 0; base < nz; base += 256) {
        const int j0 = base + lane, j1 = j0 + 64, j2 = j0 + 128, j3 = j0 + 192;
        const int q0 = min(j0, last), q1 = min(j1, last), q2 = min(j2, last), q3 = min(j3, last);
        double a0, a1, a2, a3;
        int c0, c1, c2, c3;
        if (MODE == 4) {
            a0 = __builtin_nontemporal_load(vp + q0); a1 = __builtin_nontemporal_load(vp + q1);
            a2 = __builtin_nontemporal_load(vp + q2); a3 = __builtin_nontemporal_load(vp + q3);
            c0 = __builtin_nontemporal_load(cp + q0); c1 = __builtin_nontemporal_load(cp + q1);
            c2 = __builtin_nontemporal_load(cp + q2); c3 = __builtin_nontemporal_load(cp + q3);
        } else {
            a0 = vp[q0]; a1 = vp[q1]; a2 = vp[q2]; a3 = vp[q3];
            c0 = cp[q0]; c1 = cp[q1]; c2 = cp[q2]; c3 = cp[q3];
        }
        double g0, g1, g2, g3;
        if (MODE == 1) {
            g0 = 1.0 + c0 * 1e-9; g1 = 1.0 + c1 * 1e-9; g2 = 1.0 + c2 * 1e-9; g3 = 1.0 + c3 * 1e-9;
        } else if (MODE == 2) {
            g0 = vec[(k0 + q0) % n]; g1 = vec[(k0 + q1) % n]; g2 = vec[(k0 + q2) % n]; g3 = vec[(k0 + q3) % n];
            g0 += c0 * 1e-9;
        } else {
            g0 = vec[c0]; g1 = vec[c1]; g2 = vec[c2]; g3 = vec[c3];
        }
        if (MODE == 3) {
            priv += a0 * g0 + a1 * g1 + a2 * g2 + a3 * g3;
        } else {
            lds[wave][j0] = a0 * g0; lds[wave][j1] = a1 * g1; lds[wave][j2] = a2 * g2; lds[wave][j3] = a3 * g3;
        }
    }
    if (MODE == 3) {
        if (lane < nr) out[r0 + lane] = priv;
        return;
    }
    wsync();
    if (lane < nr) {
        double s = 0.0;
        for (int j = rs; j < re; ++j) s += lds[wave][j];
        out[r0 + lane] = s;
    }
}

// 16-byte value loads: lane handles two consecutive nonzeros
__global__ void __launch_bounds__(256) k_wide(const int4 *blk, int nblk, const int *rowptr, const int *col,
                                              const double *val, const double *vec, double *out, int n) {
    __shared__ double lds[4][W + 2];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + wave;
    if (b >= nblk) return;
    const int4 d = blk[b];
    const int r0 = d.x, nr = d.y, k0 = d.z, nz = d.w;
    int rs = 0, re = 0;
    if (lane < nr) {
        rs = rowptr[r0 + lane] - k0;
        re = rowptr[r0 + lane + 1] - k0;
    }
    const int ka = k0 & ~1;          // 16-byte aligned start
    const int sh = k0 - ka;          // 0 or 1 leading foreign element
    const int tot = nz + sh;         // elements from ka
    const int lastpair = (tot - 1) >> 1;
    const double2 *vp = reinterpret_cast<const double2 *>(val + ka);
    const int2 *cp = reinterpret_cast<const int2 *>(col + ka);
    for (int base = 0; base * 2 < tot; base += 128) {
        const int p0 = min(base + lane, lastpair), p1 = min(base + lane + 64, lastpair);
        const double2 a0 = vp[p0], a1 = vp[p1];
        const int2 c0 = cp[p0], c1 = cp[p1];
        const double g00 = vec[c0.x], g01 = vec[c0.y], g10 = vec[c1.x], g11 = vec[c1.y];
        // element index relative to k0 is 2*p - sh (+1); slot -1 lands in lds[wave][0] padding via +1 offset
        double *L = &lds[wave][1 - sh];
        if (base + lane <= lastpair) {
            L[2 * (base + lane)] = a0.x * g00;
            L[2 * (base + lane) + 1] = a0.y * g01;
        }
        if (base + lane + 64 <= lastpair) {
            L[2 * (base + lane + 64)] = a1.x * g10;
            L[2 * (base + lane + 64) + 1] = a1.y * g11;
        }
    }
    wsync();
    if (lane < nr) {
        double s = 0.0;
        for (int j = rs; j < re; ++j) s += lds[wave][1 + j];
        out[r0 + lane] = s;
    }
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
    std::vector<int4> blk;
    for (int r = 0; r < m;) {
        int s = r, nz = 0;
        while (r < m && r - s < 64 && nz + per_row <= W) { nz += per_row; ++r; }
        blk.push_back(make_int4(s, r - s, rp[s], nz));
    }
    const int nblk = (int)blk.size();
    int *d_rp, *d_col;
    int4 *d_blk;
    double *d_val, *d_vec, *d_out;
    CK(hipMalloc(&d_rp, (m + 1) * 4)); CK(hipMalloc(&d_col, (nnz + 4) * 4)); CK(hipMalloc(&d_val, (nnz + 4) * 8));
    CK(hipMalloc(&d_blk, nblk * sizeof(int4))); CK(hipMalloc(&d_vec, (size_t)n * 8)); CK(hipMalloc(&d_out, (size_t)m * 8));
    CK(hipMemcpy(d_rp, rp.data(), (m + 1) * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_col, col.data(), nnz * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_val, val.data(), nnz * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_blk, blk.data(), nblk * sizeof(int4), hipMemcpyHostToDevice));
    std::vector<double> vec(n);
    for (int i = 0; i < n; ++i) vec[i] = 1.0 + (i % 13) * 0.01;
    CK(hipMemcpy(d_vec, vec.data(), (size_t)n * 8, hipMemcpyHostToDevice));
    const double bytes = 12.0 * nnz + 4.0 * (m + 1) + 8.0 * n + 8.0 * m;
    const int grid = (nblk + 3) / 4;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<double> ref(m), got(m);
    auto run = [&](const char *name, auto kern, bool check) {
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d_blk, nblk, d_rp, d_col, d_val, d_vec, d_out, n);
        CK(hipDeviceSynchronize());
        const int reps = 20;
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d_blk, nblk, d_rp, d_col, d_val, d_vec, d_out, n);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        double maxerr = -1;
        if (check) {
            CK(hipMemcpy(got.data(), d_out, (size_t)m * 8, hipMemcpyDeviceToHost));
            maxerr = 0;
            for (int i = 0; i < m; i += 997) maxerr = std::max(maxerr, std::fabs(got[i] - ref[i]));
        }
        printf("%-28s %8.3f ms  %8.1f GB/s (algorithmic)  maxerr %.3g\n", name, ms, bytes / ms * 1e-6, maxerr);
    };
    for (int i = 0; i < m; i += 997) {
        double s = 0;
        for (int k = rp[i]; k < rp[i + 1]; ++k) s += val[k] * vec[col[k]];
        ref[i] = s;
    }
    printf("m=n=%d nnz=%zu band=%d blocks=%d bytes=%.3f GB\n", m, nnz, band, nblk, bytes * 1e-9);
    run("full (current)", k_stream<0>, true);
    run("no gather", k_stream<1>, false);
    run("sequential gather", k_stream<2>, false);
    run("no LDS phase", k_stream<3>, false);
    run("full, nontemporal matrix", k_stream<4>, true);
    run("16B loads", k_wide, true);
    return 0;
}
