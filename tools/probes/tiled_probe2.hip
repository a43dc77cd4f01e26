// tiled_probe2.hip -- developer prototype v2 of the column-tiled SpMV: lane-owned row segments, one barrier per tile.
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/tiled_probe2.hip -o bin/tiled_probe2
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
// tiled format v2: entries of a dense (super-block, tile) are sorted by (row, col) and packed in
// chunks of K=4; a row segment never straddles a chunk (zero-valued padding entries continue the
// previous row), so one lane owns every segment it touches and no two lanes of a step write the
// same accumulator.  Segments longer than K and entries of sparse tiles go to the remainder list.
// ---------------------------------------------------------------------------------------------
constexpr int K = 4;
static int g_rotate = 0, g_band = 0, g_step_mult = 1;
struct Step {
    int col0;     // first column of the tile (tile steps)
    int e_begin;  // entry range (multiple of K for tile steps)
    int e_end;
    int kind;     // bit0: remainder step (global gather); bit1: stage a new tile first
};

struct Tiled {
    std::vector<int> sb_mid;  // first remainder step of each super-block
    std::vector<int> sb_ptr;
    std::vector<Step> steps;
    std::vector<double> tval;
    std::vector<uint32_t> tidx;  // lcol << 16 | lrow
    std::vector<double> rval;
    std::vector<int> rcol;
    std::vector<uint16_t> rrow;
    size_t pad = 0;
};

static Tiled build_tiled(int m, int n, const int *rp, const int *col, const double *val, int R, int T, int E,
                         int dense_min, int Er = 0) {
    Tiled t;
    if (Er == 0) Er = E;
    const int nsb = (m + R - 1) / R, ntile = (n + T - 1) / T;
    t.sb_ptr.assign(nsb + 1, 0);
    std::vector<int> cnt(ntile, 0), slot(ntile, -1);
    std::vector<int> touched;
    struct Ent { int row, col; double v; };
    std::vector<std::vector<Ent>> bucket;
    size_t rem_base = 0;
    for (int sb = 0; sb < nsb; ++sb) {
        const int r0 = sb * R, r1 = std::min(m, r0 + R);
        touched.clear();
        for (int k = rp[r0]; k < rp[r1]; ++k) {
            const int tl = col[k] / T;
            if (cnt[tl]++ == 0) touched.push_back(tl);
        }
        std::sort(touched.begin(), touched.end());
        int nd = 0;
        for (int tl : touched) slot[tl] = (cnt[tl] >= dense_min) ? nd++ : -1;
        if ((int)bucket.size() < nd) bucket.resize(nd);
        for (int i = 0; i < nd; ++i) bucket[i].clear();
        for (int r = r0; r < r1; ++r)
            for (int k = rp[r]; k < rp[r + 1]; ++k) {
                const int tl = col[k] / T;
                if (slot[tl] >= 0) bucket[slot[tl]].push_back(Ent{r - r0, col[k] - tl * T, val[k]});
                else { t.rval.push_back(val[k]); t.rcol.push_back(col[k]); t.rrow.push_back((uint16_t)(r - r0)); }
            }
        // rotated tile order: the 32 consecutive super-blocks that run together on one XCD all start at
        // the first tile of the LAST of them, so that they read the same tile at about the same time
        std::vector<int> order_t(touched.begin(), touched.end());
        if (g_rotate) {
            const int grp_last = std::min(nsb - 1, (sb / 32) * 32 + 31);
            const int start_tile = std::max(0, (int)(((long)grp_last * R * (long)n / m - g_band) / T));
            auto it = std::lower_bound(order_t.begin(), order_t.end(), start_tile);
            std::rotate(order_t.begin(), it, order_t.end());
        }
        for (int tl : order_t) {
            if (slot[tl] < 0) continue;
            std::vector<Ent> &b = bucket[slot[tl]];  // already sorted by (row, col)
            const size_t tile_begin = t.tval.size();
            size_t i = 0;
            int last_row = b.empty() ? 0 : b[0].row;
            while (i < b.size()) {
                size_t j = i;
                while (j < b.size() && b[j].row == b[i].row) ++j;
                const int len = (int)(j - i);
                if (len > K) {  // long segment: remainder path
                    for (size_t q = i; q < j; ++q) { t.rval.push_back(b[q].v); t.rcol.push_back(tl * T + b[q].col); t.rrow.push_back((uint16_t)b[q].row); }
                    i = j;
                    continue;
                }
                const int pos = (int)((t.tval.size() - tile_begin) % K);
                if (pos + len > K)
                    for (int q = pos; q < K; ++q) { t.tval.push_back(0.0); t.tidx.push_back((uint32_t)last_row); ++t.pad; }
                for (size_t q = i; q < j; ++q) { t.tval.push_back(b[q].v); t.tidx.push_back(((uint32_t)b[q].col << 16) | (uint32_t)b[q].row); }
                last_row = b[i].row;
                i = j;
            }
            while ((t.tval.size() - tile_begin) % K) { t.tval.push_back(0.0); t.tidx.push_back((uint32_t)last_row); ++t.pad; }
            size_t p = tile_begin, end = t.tval.size();
            bool first = true;
            while (p < end) {
                const size_t c = std::min<size_t>(end - p, E);
                t.steps.push_back(Step{tl * T, (int)p, (int)(p + c), first ? 2 : 0});
                p += c;
                first = false;
            }
        }
        // NOTE: remainder entries of this super-block were appended in a mixed order (sparse tiles in row
        // order, long segments per tile); sort them by (row, col) so that head sums see contiguous rows
        {
            const size_t rb = rem_base, re = t.rval.size();
            std::vector<size_t> ord(re - rb);
            for (size_t q = 0; q < ord.size(); ++q) ord[q] = rb + q;
            std::stable_sort(ord.begin(), ord.end(), [&](size_t a, size_t b2) {
                return t.rrow[a] != t.rrow[b2] ? t.rrow[a] < t.rrow[b2] : t.rcol[a] < t.rcol[b2];
            });
            std::vector<double> v2(ord.size()); std::vector<int> c2(ord.size()); std::vector<uint16_t> r2(ord.size());
            for (size_t q = 0; q < ord.size(); ++q) { v2[q] = t.rval[ord[q]]; c2[q] = t.rcol[ord[q]]; r2[q] = t.rrow[ord[q]]; }
            std::copy(v2.begin(), v2.end(), t.rval.begin() + rb);
            std::copy(c2.begin(), c2.end(), t.rcol.begin() + rb);
            std::copy(r2.begin(), r2.end(), t.rrow.begin() + rb);
            // pad the tile steps of this super-block to a multiple of g_step_mult with empty steps
            while (g_step_mult > 1 && (int)t.steps.size() > t.sb_ptr[sb] && ((int)t.steps.size() - t.sb_ptr[sb]) % g_step_mult) {
                const Step lastst = t.steps.back();
                t.steps.push_back(Step{lastst.col0, lastst.e_end, lastst.e_end, 2});
            }
            t.sb_mid.push_back((int)t.steps.size());
            size_t p = rb;
            while (p < re) {
                const size_t c = std::min<size_t>(re - p, Er);
                t.steps.push_back(Step{0, (int)p, (int)(p + c), 1});
                p += c;
            }
            rem_base = re;
        }
        for (int tl : touched) { cnt[tl] = 0; slot[tl] = -1; }
        t.sb_ptr[sb + 1] = (int)t.steps.size();
    }
    return t;
}

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int NT, int R, int T, int NBUF, int ABL = 0>
__global__ void __launch_bounds__(NT) k_tiled2(const int *__restrict__ sb_ptr, const Step *__restrict__ steps,
                                               const double *__restrict__ tval, const uint32_t *__restrict__ tidx,
                                               const double *__restrict__ rval, const int *__restrict__ rcol,
                                               const uint16_t *__restrict__ rrow, const double *__restrict__ vec,
                                               double *__restrict__ out, int m, int n) {
    constexpr int E = NT * K;
    constexpr int TPT = T / NT;
    static_assert(T * 8 >= E * 8 + (E + 2) * 2, "remainder scratch must fit in the tile buffer");
    __shared__ double acc[R];
    __shared__ double ytile_buf[NBUF][T];
    int cur_buf = 0;
    double *ytile = ytile_buf[0];
    const int tid = threadIdx.x;
    const int sb = blockIdx.x;
    for (int i = tid; i < R; i += NT) acc[i] = 0.0;
    const int s0 = sb_ptr[sb], s1 = sb_ptr[sb + 1];
    Step nst = (s0 < s1) ? steps[s0] : Step{0, 0, 0, 0};
    double nv[K];
    uint32_t ni[K];
    uint16_t nr[K];
    double nt_[TPT];
    auto prefetch = [&](const Step &st) {
        if (st.kind & 1) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int e = st.e_begin + tid + k * NT;
                const int ee = (e < st.e_end) ? e : st.e_begin;
                nv[k] = __builtin_nontemporal_load(rval + ee);
                ni[k] = (uint32_t)__builtin_nontemporal_load(rcol + ee);
                nr[k] = __builtin_nontemporal_load(rrow + ee);
            }
        } else if (ABL == 3) {
            nv[0] = nv[1] = nv[2] = nv[3] = 1.0;
            ni[0] = ni[1] = ni[2] = ni[3] = (uint32_t)((tid * 4) % R) | ((uint32_t)(tid % T) << 16);
        } else {
            const int e = st.e_begin + K * tid;
            const int ee = (e < st.e_end) ? e : st.e_begin;
            typedef double d2_t __attribute__((ext_vector_type(2)));
            typedef unsigned u4_t __attribute__((ext_vector_type(4)));
            const d2_t a = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(tval + ee));
            const d2_t b = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(tval + ee) + 1);
            const u4_t q = __builtin_nontemporal_load(reinterpret_cast<const u4_t *>(tidx + ee));
            nv[0] = a.x; nv[1] = a.y; nv[2] = b.x; nv[3] = b.y;
            ni[0] = q.x; ni[1] = q.y; ni[2] = q.z; ni[3] = q.w;
        }
        if ((st.kind & 2) && ABL != 2) {
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
        for (int k = 0; k < K; ++k) { cv[k] = nv[k]; ci[k] = ni[k]; cr[k] = nr[k]; }
        if (st.kind & 1) lds_barrier();  // remainder scratch aliases the tile: wait for its last readers
        if (st.kind & 2) {
            if (NBUF == 1) lds_barrier();       // single buffer: every reader of the old tile must be done
            else { cur_buf ^= 1; ytile = ytile_buf[cur_buf]; }
            if (ABL != 2) {
#pragma unroll
            for (int j = 0; j < TPT; ++j) ytile[tid + j * NT] = nt_[j];
            }
        }
        double *prod = ytile;                                              // remainder steps reuse the tile buffer
        uint16_t *rows = reinterpret_cast<uint16_t *>(ytile + E);
        if (s + 1 < s1) {
            nst = steps[s + 1];
            prefetch(nst);
        }
        const int cnt = st.e_end - st.e_begin;
        if (!(st.kind & 1)) {
            lds_barrier();  // tile visible; previous step's accumulator updates done
            if (ABL == 1) {
                if (K * tid < cnt) acc[ci[0] & 0xffffu] = cv[0] + cv[1] + cv[2] + cv[3] + (double)(ci[1] + ci[2] + ci[3]);
            } else if (K * tid < cnt) {
                uint32_t cur = ci[0] & 0xffffu;
                double sacc = acc[cur];
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const uint32_t rw = ci[k] & 0xffffu;
                    if (rw != cur) {
                        acc[cur] = sacc;
                        cur = rw;
                        sacc = acc[cur];
                    }
                    sacc += cv[k] * ytile[ci[k] >> 16];
                }
                acc[cur] = sacc;
            }
        } else {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int el = tid + k * NT;
                if (el < cnt) {
                    prod[el] = cv[k] * vec[ci[k]];
                    rows[el + 1] = cr[k];
                }
            }
            if (tid == 0) rows[0] = 0xffffu;
            lds_barrier();
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int el = tid + k * NT;
                if (el < cnt) {
                    const uint16_t rw = rows[el + 1];
                    if (rows[el] != rw) {
                        double sacc = acc[rw];
                        int j = el;
                        do { sacc += prod[j]; ++j; } while (j < cnt && rows[j + 1] == rw);
                        acc[rw] = sacc;
                    }
                }
            }
        }
    }
    __syncthreads();
    const int r0 = sb * R;
    for (int i = tid; i < R && r0 + i < m; i += NT) out[r0 + i] = acc[i];
}

// v3: prefetch depth D (register sets rotate by compile-time unrolling), single tile buffer
template <int NT, int R, int T, int D>
__global__ void __launch_bounds__(NT) k_tiled3(const int *__restrict__ sb_ptr, const Step *__restrict__ steps,
                                               const double *__restrict__ tval, const uint32_t *__restrict__ tidx,
                                               const double *__restrict__ rval, const int *__restrict__ rcol,
                                               const uint16_t *__restrict__ rrow, const double *__restrict__ vec,
                                               double *__restrict__ out, int m, int n) {
    constexpr int E = NT * K;
    constexpr int TPT = T / NT;
    static_assert(T * 8 >= E * 8 + (E + 2) * 2, "remainder scratch must fit in the tile buffer");
    __shared__ double acc[R];
    __shared__ double ytile[T];
    double *prod = ytile;
    uint16_t *rows = reinterpret_cast<uint16_t *>(ytile + E);
    const int tid = threadIdx.x;
    const int sb = blockIdx.x;
    for (int i = tid; i < R; i += NT) acc[i] = 0.0;
    const int s0 = sb_ptr[sb], s1 = sb_ptr[sb + 1];
    struct Regs {
        Step st;
        double v[K];
        uint32_t i[K];
        uint16_t r[K];
        double t[TPT];
    };
    Regs q[D];
    auto prefetch = [&](Regs &g, int s) {
        g.st = steps[s];
        const Step &st = g.st;
        if (st.kind & 1) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int e = st.e_begin + tid + k * NT;
                const int ee = (e < st.e_end) ? e : st.e_begin;
                g.v[k] = __builtin_nontemporal_load(rval + ee);
                g.i[k] = (uint32_t)__builtin_nontemporal_load(rcol + ee);
                g.r[k] = __builtin_nontemporal_load(rrow + ee);
            }
        } else {
            const int e = st.e_begin + K * tid;
            const int ee = (e < st.e_end) ? e : st.e_begin;
            typedef double d2_t __attribute__((ext_vector_type(2)));
            typedef unsigned u4_t __attribute__((ext_vector_type(4)));
            const d2_t a = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(tval + ee));
            const d2_t b = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(tval + ee) + 1);
            const u4_t w = __builtin_nontemporal_load(reinterpret_cast<const u4_t *>(tidx + ee));
            g.v[0] = a.x; g.v[1] = a.y; g.v[2] = b.x; g.v[3] = b.y;
            g.i[0] = w.x; g.i[1] = w.y; g.i[2] = w.z; g.i[3] = w.w;
        }
        if (st.kind & 2) {
#pragma unroll
            for (int j = 0; j < TPT; ++j) {
                const int c = st.col0 + tid + j * NT;
                g.t[j] = (c < n) ? vec[c] : 0.0;
            }
        }
    };
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (s0 + d < s1) prefetch(q[d], s0 + d);
    for (int sbase = s0; sbase < s1; sbase += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int s = sbase + d;
            if (s < s1) {
                const Step st = q[d].st;
                double cv[K];
                uint32_t ci[K];
                uint16_t cr[K];
#pragma unroll
                for (int k = 0; k < K; ++k) { cv[k] = q[d].v[k]; ci[k] = q[d].i[k]; cr[k] = q[d].r[k]; }
                if (st.kind & 3) lds_barrier();  // readers of the old tile / scratch must be done
                if (st.kind & 2) {
#pragma unroll
                    for (int j = 0; j < TPT; ++j) ytile[tid + j * NT] = q[d].t[j];
                }
                if (s + D < s1) prefetch(q[d], s + D);
                const int cnt = st.e_end - st.e_begin;
                if (!(st.kind & 1)) {
                    lds_barrier();
                    if (K * tid < cnt) {
                        uint32_t cur = ci[0] & 0xffffu;
                        double sacc = acc[cur];
#pragma unroll
                        for (int k = 0; k < K; ++k) {
                            const uint32_t rw = ci[k] & 0xffffu;
                            if (rw != cur) {
                                acc[cur] = sacc;
                                cur = rw;
                                sacc = acc[cur];
                            }
                            sacc += cv[k] * ytile[ci[k] >> 16];
                        }
                        acc[cur] = sacc;
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const int el = tid + k * NT;
                        if (el < cnt) {
                            prod[el] = cv[k] * vec[ci[k]];
                            rows[el + 1] = cr[k];
                        }
                    }
                    if (tid == 0) rows[0] = 0xffffu;
                    lds_barrier();
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const int el = tid + k * NT;
                        if (el < cnt) {
                            const uint16_t rw = rows[el + 1];
                            if (rows[el] != rw) {
                                double sacc = acc[rw];
                                int j = el;
                                do { sacc += prod[j]; ++j; } while (j < cnt && rows[j + 1] == rw);
                                acc[rw] = sacc;
                            }
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    const int r0 = sb * R;
    for (int i = tid; i < R && r0 + i < m; i += NT) out[r0 + i] = acc[i];
}


// v4: branch-free prefetch (every tile step stages its tile, clamped addresses), step descriptors
// two ahead, KC chunks of 4 entries per lane; remainder steps run in a second loop.
template <int NT, int R, int T, int KC, int ABL = 0>
__global__ void __launch_bounds__(NT) k_tiled4(const int *__restrict__ sb_ptr, const int *__restrict__ sb_mid,
                                               const Step *__restrict__ steps,
                                               const double *__restrict__ tval, const uint32_t *__restrict__ tidx,
                                               const double *__restrict__ rval, const int *__restrict__ rcol,
                                               const uint16_t *__restrict__ rrow, const double *__restrict__ vec,
                                               double *__restrict__ out, int m, int n, int nsb_total) {
    constexpr int KR = ((T * 8 - 8) / (10 * NT)) < K ? ((T * 8 - 8) / (10 * NT)) : K;
    constexpr int E = NT * KR;  // remainder step capacity
    constexpr int TPT = T / NT;
    static_assert(KR >= 1 && T * 8 >= E * 8 + (E + 2) * 2, "remainder scratch must fit in the tile buffer");
    typedef double d2_t __attribute__((ext_vector_type(2)));
    typedef unsigned u4_t __attribute__((ext_vector_type(4)));
    __shared__ double acc[R];
    __shared__ double ytile[T];
    const int tid = threadIdx.x;
    // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs, so give XCD x the contiguous
    // range of super-blocks [x*per, (x+1)*per): concurrently running workgroups of one XCD then share
    // most of their column window in that XCD's L2
    const int per = gridDim.x / 8;   // the grid is padded to a multiple of 8 workgroups
    const int sb = (blockIdx.x % 8) * per + blockIdx.x / 8;
    if (sb >= nsb_total) return;
    for (int i = tid; i < R; i += NT) acc[i] = 0.0;
    const int s0 = sb_ptr[sb], smid = sb_mid[sb], s1 = sb_ptr[sb + 1];
    // ---- tile steps
    if (s0 < smid) {
        d2_t va[KC], vb[KC];
        u4_t ix[KC];
        double tl[TPT];
        Step st = steps[s0];
        Step st_next = steps[min(s0 + 1, smid - 1)];
        auto issue = [&](const Step &q) {
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                const int e = q.e_begin + K * (tid + c * NT);
                const int ee = (e < q.e_end) ? e : q.e_begin;
                if (ABL == 1) {
                    va[c] = d2_t{1.0, 1.0}; vb[c] = d2_t{1.0, 1.0};
                    const unsigned w = ((unsigned)(tid % T) << 16) | (unsigned)((tid * 4) % R);
                    ix[c] = u4_t{w, w + 1, w + 2, w + 3};
                } else {
                va[c] = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(tval + ee));
                vb[c] = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(tval + ee) + 1);
                ix[c] = __builtin_nontemporal_load(reinterpret_cast<const u4_t *>(tidx + ee));
                }
            }
            if (ABL == 2 || ABL == 3 || ABL == 4) {
#pragma unroll
                for (int j = 0; j < TPT; ++j) tl[j] = 1.0;
            } else {
#pragma unroll
            for (int j = 0; j < TPT; ++j) tl[j] = vec[min(q.col0 + tid + j * NT, n - 1)];
            }
        };
        issue(st);
        for (int s = s0; s < smid; ++s) {
            const Step cur = st;
            d2_t ca[KC], cb[KC];
            u4_t cx[KC];
#pragma unroll
            for (int c = 0; c < KC; ++c) { ca[c] = va[c]; cb[c] = vb[c]; cx[c] = ix[c]; }
            if (ABL != 4) lds_barrier();  // every lane is done reading the previous tile
            if (ABL != 3 && ABL != 4) {
#pragma unroll
            for (int j = 0; j < TPT; ++j) ytile[tid + j * NT] = tl[j];
            }
            st = st_next;
            st_next = steps[min(s + 2, smid - 1)];
            if (s + 1 < smid) issue(st);
            if (ABL != 4) lds_barrier();  // tile visible
            const int cnt = cur.e_end - cur.e_begin;
            if (ABL == 3 || ABL == 4) {
                double z = 0;
#pragma unroll
                for (int c = 0; c < KC; ++c) z += ca[c].x + ca[c].y + cb[c].x + cb[c].y + (double)(cx[c].x + cx[c].y + cx[c].z + cx[c].w);
                if (z == 123.456) acc[tid % R] = z;
                continue;
            }
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                if (K * (tid + c * NT) < cnt) {
                    const double v[K] = {ca[c].x, ca[c].y, cb[c].x, cb[c].y};
                    const uint32_t id[K] = {cx[c].x, cx[c].y, cx[c].z, cx[c].w};
                    // all LDS reads first (the rows of this chunk are touched by no other lane in this
                    // step), then the running segment sums in registers, then the writes: two LDS round
                    // trips per chunk instead of one per entry
                    uint32_t rw[K];
                    double a[K], y[K];
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        rw[k] = id[k] & 0xffffu;
                        a[k] = acc[rw[k]];
                        y[k] = ytile[id[k] >> 16];
                    }
                    double sk[K];
                    sk[0] = a[0] + v[0] * y[0];
#pragma unroll
                    for (int k = 1; k < K; ++k) sk[k] = ((rw[k] == rw[k - 1]) ? sk[k - 1] : a[k]) + v[k] * y[k];
#pragma unroll
                    for (int k = 0; k < K; ++k)
                        if (k == K - 1 || rw[k] != rw[k + 1]) acc[rw[k]] = sk[k];
                }
            }
        }
    }
    // ---- remainder steps (global gathers, head sums through the tile buffer as scratch)
    double *prod = ytile;
    uint16_t *rows = reinterpret_cast<uint16_t *>(ytile + E);
    for (int s = smid; s < s1; ++s) {
        const Step st = steps[s];
        const int cnt = st.e_end - st.e_begin;
        lds_barrier();
#pragma unroll
        for (int k = 0; k < KR; ++k) {
            const int el = tid + k * NT;
            if (el < cnt) {
                const int e = st.e_begin + el;
                prod[el] = rval[e] * vec[rcol[e]];
                rows[el + 1] = rrow[e];
            }
        }
        if (tid == 0) rows[0] = 0xffffu;
        lds_barrier();
#pragma unroll
        for (int k = 0; k < KR; ++k) {
            const int el = tid + k * NT;
            if (el < cnt) {
                const uint16_t rw = rows[el + 1];
                if (rows[el] != rw) {
                    double sacc = acc[rw];
                    int j = el;
                    do { sacc += prod[j]; ++j; } while (j < cnt && rows[j + 1] == rw);
                    acc[rw] = sacc;
                }
            }
        }
    }
    __syncthreads();
    const int r0 = sb * R;
    for (int i = tid; i < R && r0 + i < m; i += NT) out[r0 + i] = acc[i];
}


// v5: v4 with prefetch depth D (register sets rotated by compile-time unrolling, clamped step index
// so that every issue is unconditional)
template <int NT, int R, int T, int KC, int D, int ABL = 0>
__global__ void __launch_bounds__(NT) k_tiled5(const int *__restrict__ sb_ptr, const int *__restrict__ sb_mid,
                                               const Step *__restrict__ steps,
                                               const double *__restrict__ tval, const uint32_t *__restrict__ tidx,
                                               const double *__restrict__ rval, const int *__restrict__ rcol,
                                               const uint16_t *__restrict__ rrow, const double *__restrict__ vec,
                                               double *__restrict__ out, int m, int n, int nsb_total) {
    constexpr int E = NT * K;
    constexpr int TPT = T / NT;
    static_assert(T * 8 >= E * 8 + (E + 2) * 2, "remainder scratch must fit in the tile buffer");
    typedef double d2_t __attribute__((ext_vector_type(2)));
    typedef unsigned u4_t __attribute__((ext_vector_type(4)));
    __shared__ double acc[R];
    __shared__ double ytile[T];
    const int tid = threadIdx.x;
    const int per = gridDim.x / 8;
    const int sb = (blockIdx.x % 8) * per + blockIdx.x / 8;
    if (sb >= nsb_total) return;
    for (int i = tid; i < R; i += NT) acc[i] = 0.0;
    const int s0 = sb_ptr[sb], smid = sb_mid[sb], s1 = sb_ptr[sb + 1];
    if (s0 < smid) {
        d2_t va[D][KC], vb[D][KC];
        u4_t ix[D][KC];
        double tl[D][TPT];
        Step stq[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            stq[d] = steps[min(s0 + d, smid - 1)];
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                const int e = stq[d].e_begin + K * (tid + c * NT);
                const int ee = (e < stq[d].e_end) ? e : stq[d].e_begin;
                va[d][c] = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(tval + ee));
                vb[d][c] = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(tval + ee) + 1);
                ix[d][c] = __builtin_nontemporal_load(reinterpret_cast<const u4_t *>(tidx + ee));
            }
#pragma unroll
            for (int j = 0; j < TPT; ++j) tl[d][j] = vec[min(stq[d].col0 + tid + j * NT, n - 1)];
        }
        for (int sbase = s0; sbase < smid; sbase += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int s = sbase + d;
                if (s < smid) {
                    const Step cur = stq[d];
                    d2_t ca[KC], cb[KC];
                    u4_t cx[KC];
#pragma unroll
                    for (int c = 0; c < KC; ++c) { ca[c] = va[d][c]; cb[c] = vb[d][c]; cx[c] = ix[d][c]; }
                    if (ABL != 4) lds_barrier();
                    if (ABL != 4) {
#pragma unroll
                        for (int j = 0; j < TPT; ++j) ytile[tid + j * NT] = tl[d][j];
                    }
                    {   // refill this register set with step s + D
                        stq[d] = steps[min(s + D, smid - 1)];
#pragma unroll
                        for (int c = 0; c < KC; ++c) {
                            const int e = stq[d].e_begin + K * (tid + c * NT);
                            const int ee = (e < stq[d].e_end) ? e : stq[d].e_begin;
                            va[d][c] = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(tval + ee));
                            vb[d][c] = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(tval + ee) + 1);
                            ix[d][c] = __builtin_nontemporal_load(reinterpret_cast<const u4_t *>(tidx + ee));
                        }
#pragma unroll
                        for (int j = 0; j < TPT; ++j) tl[d][j] = vec[min(stq[d].col0 + tid + j * NT, n - 1)];
                    }
                    if (ABL != 4) lds_barrier();
                    const int cnt = cur.e_end - cur.e_begin;
                    if (ABL == 4) {
                        double z = 0;
#pragma unroll
                        for (int c = 0; c < KC; ++c) z += ca[c].x + ca[c].y + cb[c].x + cb[c].y + (double)(cx[c].x + cx[c].y + cx[c].z + cx[c].w);
                        if (z == 123.456) acc[tid % R] = z;
                    } else {
#pragma unroll
                    for (int c = 0; c < KC; ++c) {
                        if (K * (tid + c * NT) < cnt) {
                            const double v[K] = {ca[c].x, ca[c].y, cb[c].x, cb[c].y};
                            const uint32_t id[K] = {cx[c].x, cx[c].y, cx[c].z, cx[c].w};
                            uint32_t rw[K];
                            double a[K], y[K];
#pragma unroll
                            for (int k = 0; k < K; ++k) {
                                rw[k] = id[k] & 0xffffu;
                                a[k] = acc[rw[k]];
                                y[k] = ytile[id[k] >> 16];
                            }
                            double sk[K];
                            sk[0] = a[0] + v[0] * y[0];
#pragma unroll
                            for (int k = 1; k < K; ++k) sk[k] = ((rw[k] == rw[k - 1]) ? sk[k - 1] : a[k]) + v[k] * y[k];
#pragma unroll
                            for (int k = 0; k < K; ++k)
                                if (k == K - 1 || rw[k] != rw[k + 1]) acc[rw[k]] = sk[k];
                        }
                    }
                    }
                }
            }
        }
    }
    double *prod = ytile;
    uint16_t *rows = reinterpret_cast<uint16_t *>(ytile + E);
    for (int s = smid; s < s1; ++s) {
        const Step st = steps[s];
        const int cnt = st.e_end - st.e_begin;
        lds_barrier();
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int el = tid + k * NT;
            if (el < cnt) {
                const int e = st.e_begin + el;
                prod[el] = rval[e] * vec[rcol[e]];
                rows[el + 1] = rrow[e];
            }
        }
        if (tid == 0) rows[0] = 0xffffu;
        lds_barrier();
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int el = tid + k * NT;
            if (el < cnt) {
                const uint16_t rw = rows[el + 1];
                if (rows[el] != rw) {
                    double sacc = acc[rw];
                    int j = el;
                    do { sacc += prod[j]; ++j; } while (j < cnt && rows[j + 1] == rw);
                    acc[rw] = sacc;
                }
            }
        }
    }
    __syncthreads();
    const int r0 = sb * R;
    for (int i = tid; i < R && r0 + i < m; i += NT) out[r0 + i] = acc[i];
}


// v6: two register sets processed back to back in a straight-line loop body (the step list of a
// super-block is padded to an even count), so that the compiler can keep one step of loads in flight
// across the barriers of the other
template <int NT, int R, int T>
__global__ void __launch_bounds__(NT) k_tiled6(const int *__restrict__ sb_ptr, const int *__restrict__ sb_mid,
                                               const Step *__restrict__ steps,
                                               const double *__restrict__ tval, const uint32_t *__restrict__ tidx,
                                               const double *__restrict__ rval, const int *__restrict__ rcol,
                                               const uint16_t *__restrict__ rrow, const double *__restrict__ vec,
                                               double *__restrict__ out, int m, int n, int nsb_total) {
    constexpr int KR = ((T * 8 - 8) / (10 * NT)) < K ? ((T * 8 - 8) / (10 * NT)) : K;
    constexpr int E = NT * KR;
    constexpr int TPT = T / NT;
    typedef double d2_t __attribute__((ext_vector_type(2)));
    typedef unsigned u4_t __attribute__((ext_vector_type(4)));
    __shared__ double acc[R];
    __shared__ double ytile[T];
    const int tid = threadIdx.x;
    const int per = gridDim.x / 8;
    const int sb = (blockIdx.x % 8) * per + blockIdx.x / 8;
    if (sb >= nsb_total) return;
    for (int i = tid; i < R; i += NT) acc[i] = 0.0;
    const int s0 = sb_ptr[sb], smid = sb_mid[sb], s1 = sb_ptr[sb + 1];
    if (s0 < smid) {
        d2_t vaA, vbA, vaB, vbB;
        u4_t ixA, ixB;
        double tlA[TPT], tlB[TPT];
        Step stA, stB;
        auto issue = [&](const Step &q, d2_t &va, d2_t &vb, u4_t &ix, double (&tl)[TPT]) {
            const int e = q.e_begin + K * tid;
            const int ee = (e < q.e_end) ? e : q.e_begin;
            va = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(tval + ee));
            vb = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(tval + ee) + 1);
            ix = __builtin_nontemporal_load(reinterpret_cast<const u4_t *>(tidx + ee));
#pragma unroll
            for (int j = 0; j < TPT; ++j) tl[j] = vec[min(q.col0 + tid + j * NT, n - 1)];
        };
        auto process = [&](Step &st, d2_t &va, d2_t &vb, u4_t &ix, double (&tl)[TPT], int next_s) {
            const Step cur = st;
            const d2_t ca = va, cb = vb;
            const u4_t cx = ix;
            lds_barrier();
#pragma unroll
            for (int j = 0; j < TPT; ++j) ytile[tid + j * NT] = tl[j];
            st = steps[min(next_s, smid - 1)];
            issue(st, va, vb, ix, tl);
            lds_barrier();
            if (K * tid < cur.e_end - cur.e_begin) {
                const double v[K] = {ca.x, ca.y, cb.x, cb.y};
                const uint32_t id[K] = {cx.x, cx.y, cx.z, cx.w};
                uint32_t rw[K];
                double a[K], y[K];
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    rw[k] = id[k] & 0xffffu;
                    a[k] = acc[rw[k]];
                    y[k] = ytile[id[k] >> 16];
                }
                double sk[K];
                sk[0] = a[0] + v[0] * y[0];
#pragma unroll
                for (int k = 1; k < K; ++k) sk[k] = ((rw[k] == rw[k - 1]) ? sk[k - 1] : a[k]) + v[k] * y[k];
#pragma unroll
                for (int k = 0; k < K; ++k)
                    if (k == K - 1 || rw[k] != rw[k + 1]) acc[rw[k]] = sk[k];
            }
        };
        stA = steps[s0];
        issue(stA, vaA, vbA, ixA, tlA);
        stB = steps[s0 + 1];
        issue(stB, vaB, vbB, ixB, tlB);
        for (int s = s0; s < smid; s += 2) {
            process(stA, vaA, vbA, ixA, tlA, s + 2);
            process(stB, vaB, vbB, ixB, tlB, s + 3);
        }
    }
    double *prod = ytile;
    uint16_t *rows = reinterpret_cast<uint16_t *>(ytile + E);
    for (int s = smid; s < s1; ++s) {
        const Step st = steps[s];
        const int cnt = st.e_end - st.e_begin;
        lds_barrier();
#pragma unroll
        for (int k = 0; k < KR; ++k) {
            const int el = tid + k * NT;
            if (el < cnt) {
                const int e = st.e_begin + el;
                prod[el] = rval[e] * vec[rcol[e]];
                rows[el + 1] = rrow[e];
            }
        }
        if (tid == 0) rows[0] = 0xffffu;
        lds_barrier();
#pragma unroll
        for (int k = 0; k < KR; ++k) {
            const int el = tid + k * NT;
            if (el < cnt) {
                const uint16_t rw = rows[el + 1];
                if (rows[el] != rw) {
                    double sacc = acc[rw];
                    int j = el;
                    do { sacc += prod[j]; ++j; } while (j < cnt && rows[j + 1] == rw);
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
               t.steps.size(), (double)t.steps.size() / nsb, 100.0 * (t.tval.size() - t.pad) / nnz, maxerr);
        printf("      padding %.2f%%\n", 100.0 * t.pad / nnz);
        hipFree(d_sb); hipFree(d_steps); hipFree(d_tval); hipFree(d_tidx); hipFree(d_rval); hipFree(d_rcol); hipFree(d_rrow);
    };
    auto run_cfg4 = [&](const char *name, auto kern, int NT, int R, int T, int KC, int dense_min) {
        const int KR = std::min(4, (T * 8 - 8) / (10 * NT));
        Tiled t = build_tiled(m, n, rp.data(), col.data(), val.data(), R, T, NT * 4 * KC, dense_min, NT * KR);
        const int nsb = (int)t.sb_ptr.size() - 1;
        int *d_sb, *d_mid, *d_rcol;
        Step *d_steps;
        double *d_tval, *d_rval;
        uint32_t *d_tidx;
        uint16_t *d_rrow;
        CK(hipMalloc(&d_sb, t.sb_ptr.size() * 4));
        CK(hipMalloc(&d_mid, t.sb_mid.size() * 4));
        CK(hipMalloc(&d_steps, std::max<size_t>(1, t.steps.size()) * sizeof(Step)));
        CK(hipMalloc(&d_tval, (t.tval.size() + 8) * 8));
        CK(hipMalloc(&d_tidx, (t.tidx.size() + 8) * 4));
        CK(hipMalloc(&d_rval, (t.rval.size() + 8) * 8));
        CK(hipMalloc(&d_rcol, (t.rcol.size() + 8) * 4));
        CK(hipMalloc(&d_rrow, (t.rrow.size() + 8) * 2));
        CK(hipMemcpy(d_sb, t.sb_ptr.data(), t.sb_ptr.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_mid, t.sb_mid.data(), t.sb_mid.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_steps, t.steps.data(), t.steps.size() * sizeof(Step), hipMemcpyHostToDevice));
        CK(hipMemcpy(d_tval, t.tval.data(), t.tval.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_tidx, t.tidx.data(), t.tidx.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_rval, t.rval.data(), t.rval.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_rcol, t.rcol.data(), t.rcol.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_rrow, t.rrow.data(), t.rrow.size() * 2, hipMemcpyHostToDevice));
        auto launch = [&]() {
            hipLaunchKernelGGL(kern, dim3((nsb + 7) / 8 * 8), dim3(NT), 0, 0, d_sb, d_mid, d_steps, d_tval, d_tidx, d_rval, d_rcol, d_rrow,
                               d_vec, d_out, m, n, nsb);
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
        printf("%-34s %8.3f ms %8.1f GB/s  steps %zu (%.1f/sb)  dense %.1f%% pad %.1f%% maxerr %.3g\n", name, ms,
               bytes / ms * 1e-6, t.steps.size(), (double)t.steps.size() / nsb, 100.0 * (t.tval.size() - t.pad) / nnz,
               100.0 * t.pad / nnz, maxerr);
        hipFree(d_sb); hipFree(d_mid); hipFree(d_steps); hipFree(d_tval); hipFree(d_tidx); hipFree(d_rval); hipFree(d_rcol); hipFree(d_rrow);
    };
    printf("m=n=%d nnz=%zu band=%d bytes=%.3f GB\n", m, nnz, band, bytes * 1e-9);
    g_band = band;
    run_cfg4("v4 NT512 R8192 T2048 KC1", k_tiled4<512, 8192, 2048, 1>, 512, 8192, 2048, 1, 256);
    g_step_mult = 2;
    run_cfg4("v6 NT512 R8192 T2048 D2", k_tiled6<512, 8192, 2048>, 512, 8192, 2048, 1, 256);
    run_cfg4("v6 NT1024 R8192 T4096 D2 (1 WG/CU)", k_tiled6<1024, 8192, 4096>, 1024, 8192, 4096, 1, 512);
    run_cfg4("v6 NT1024 R4096 T4096 D2 (2 WG/CU)", k_tiled6<1024, 4096, 4096>, 1024, 4096, 4096, 1, 512);
    run_cfg4("v6 NT512 R4096 T2048 D2 (3 WG/CU)", k_tiled6<512, 4096, 2048>, 512, 4096, 2048, 1, 256);
    run_cfg4("v6 NT512 R2048 T2048 D2 (5 WG/CU)", k_tiled6<512, 2048, 2048>, 512, 2048, 2048, 1, 256);
    g_step_mult = 1;
    for (g_rotate = 0; g_rotate < 0; ++g_rotate) {
    printf("rotate=%d\n", g_rotate);
    run_cfg4("v4 NT1024 R8192 T8192 KC2", k_tiled4<1024, 8192, 8192, 2>, 1024, 8192, 8192, 2, 512);
    run_cfg4("v4 NT1024 R8192 T8192 KC1", k_tiled4<1024, 8192, 8192, 1>, 1024, 8192, 8192, 1, 512);
    run_cfg4("v4 NT512 R4096 T4096 KC1", k_tiled4<512, 4096, 4096, 1>, 512, 4096, 4096, 1, 512);
    run_cfg4("v4 NT512 R4096 T4096 KC2", k_tiled4<512, 4096, 4096, 2>, 512, 4096, 4096, 2, 512);
    run_cfg4("v4 NT512 R8192 T4096 KC1", k_tiled4<512, 8192, 4096, 1>, 512, 8192, 4096, 1, 512);
    run_cfg4("v4 NT512 R8192 T4096 KC2", k_tiled4<512, 8192, 4096, 2>, 512, 8192, 4096, 2, 512);
    run_cfg4("v4 NT256 R4096 T4096 KC2", k_tiled4<256, 4096, 4096, 2>, 256, 4096, 4096, 2, 512);
    }
        run_cfg("v3 NT1024 R8192 T8192 D2", k_tiled3<1024, 8192, 8192, 2>, 1024, 8192, 8192, 4, 512);
    return 0;
}
