// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the HPR-LP iteration.
//
// The hot kernel is k_spmv_fused<Epi>: a CSR row-block SpMV whose epilogue is the whole half-step
// (projection, reflection, Halpern averaging) or a residual/dot reduction, so one HPR half-step is
// one launch.  Each wave owns one row block:
//   stream mode  (<= 64 rows, <= 512 nonzeros): the wave streams val/col coalesced (lane = nonzero),
//                gathers the vector, stages the products in LDS in CSR order, then lane t sums row t
//                sequentially -- the same summation order as a sequential CPU row dot, which makes
//                the result bit-identical to the oracle for these rows;
//   vector mode  (one row, > 256 nonzeros): lanes stride the row, __shfl_xor tree reduction.
// Compiled with -ffp-contract=off: products and sums round separately (see oracle/hpr_oracle.c).
//
// Reference formulas: src/cuda_kernels/HPR_cuda_kernels.cu:203-295 (updates), :160-189 (residuals),
// :91-157 (scaling), src/scaling.cu:5-38, src/power_iteration.cu:60-100.
#include "kernels.h"

#include <algorithm>
#include <vector>
#include <cmath>
#include <type_traits>

namespace hprlp {

// ------------------------------------------------------------------------------------------------
// wave / block reductions (fixed order => deterministic)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

// Sum NACC per-thread accumulators over the block; thread 0 stores partials[i*stride + blockIdx.x].
// `red`: NW * NACC doubles of LDS that no lane still uses.
template <int NACC, int NW>
__device__ __forceinline__ void block_store_partials_in(double (&acc)[NACC], double *partials, int stride, double (*red)[NACC > 0 ? NACC : 1]) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
        double v = wave_sum(acc[i]);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            double v = red[0][i];
#pragma unroll
            for (int w = 1; w < NW; ++w) v += red[w][i];
            partials[(size_t)i * stride + blockIdx.x] = v;
        }
    }
}

template <int NACC, int NW = kWavesPerBlock>
__device__ __forceinline__ void block_store_partials(double (&acc)[NACC], double *partials, int stride) {
    __shared__ double red[NW][NACC > 0 ? NACC : 1];
    block_store_partials_in<NACC, NW>(acc, partials, stride, red);
}

// Order LDS traffic between lanes of one wave (no s_barrier needed: one wave's DS ops are in order).
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ------------------------------------------------------------------------------------------------
// the fused CSR kernel
// ------------------------------------------------------------------------------------------------
// What a matrix entry a and the gathered vector element g contribute to the row sum: a * g unless the
// epilogue defines its own term (the Curtis-Reid pass sums -log|a| - g over the same pattern).
template <class Epi, class = void>
struct TermOf {
    static __device__ __forceinline__ double f(double a, double g) { return a * g; }
};
template <class Epi>
struct TermOf<Epi, std::void_t<decltype(&Epi::term)>> {
    static __device__ __forceinline__ double f(double a, double g) { return Epi::term(a, g); }
};
template <class Epi>
__device__ __forceinline__ double term(double a, double g) {
    return TermOf<Epi>::f(a, g);
}
// Tiled kernels: an epilogue with kLogTerm sums (stored value - g) over the entries, the stored values being -log|a| and NaN
// in the padding slots (launch_tiled_refresh_log); everybody else sums stored value * g (padding: 0).
template <class Epi, class = void>
struct LogTerm : std::false_type {};
template <class Epi>
struct LogTerm<Epi, std::void_t<decltype(Epi::kLogTerm)>> : std::true_type {};

template <class Epi>
__global__ void __launch_bounds__(kThreads) k_spmv_fused(CsrDev A, Epi epi) {
    constexpr int NV = Epi::NV;
    constexpr int NACC = Epi::NACC;
    __shared__ double lds[kWavesPerBlock][NV][kStreamW];

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // Workgroups are dealt round-robin over the 8 XCDs (blockIdx.x % 8 names the XCD's share): give every XCD a CONTIGUOUS
    // range of row blocks, so that neighbouring rows -- whose column windows overlap in a banded or block-structured LP --
    // gather through ONE 4 MiB L2 instead of pulling the same lines of the vector into up to eight of them (block-angular
    // ladder point, y-half: 1.65 x the algorithmic bytes left the L2s with the plain blockIdx order).  A bijection of
    // [0, gridDim.x): share x owns start_x = x * (grid / 8) + min(x, grid % 8) and the next grid / 8 (+ 1) workgroups.
    const int xcd = blockIdx.x & 7, per_lo = gridDim.x >> 3, rem = gridDim.x & 7;
    const int wg = xcd * per_lo + min(xcd, rem) + (blockIdx.x >> 3);
    const int b = wg * kWavesPerBlock + wave;
    double acc[NACC > 0 ? NACC : 1];
#pragma unroll
    for (int i = 0; i < (NACC > 0 ? NACC : 1); ++i) acc[i] = 0.0;

    epi.begin();

    if (b < A.nblk) {
        const int4 d = A.blk[b];
        const int r0 = d.x, nr = d.y, k0 = d.z, nz = d.w;
        const int *__restrict__ col = A.col + k0;
        const double *__restrict__ val = A.val + k0;

        if (nr == 0 || (nr == 1 && nz > kLongRow)) {
            // ---- vector mode: one long row per wave (nr == 0: one chunk of a split row, see below -- of any length: the last
            // chunk of a row of 4097 entries holds one)
            double s[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) s[v] = 0.0;
            int j = lane;
            // eight independent load chains first (a 4096-entry row: 2 trips of 8 instead of 4 of 4 per lane pair), same
            // accumulation sequence per lane as the four-wide loop below
            for (; j + 7 * kWave < nz; j += 8 * kWave) {
                double a[8];
                int c[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    a[u] = __builtin_nontemporal_load(val + j + u * kWave);
                    c[u] = __builtin_nontemporal_load(col + j + u * kWave);
                }
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const double *__restrict__ g = epi.gv[v];
                    double gg[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) gg[u] = g[c[u]];
#pragma unroll
                    for (int u = 0; u < 8; ++u) s[v] += term<Epi>(a[u], gg[u]);
                }
            }
            for (; j + 3 * kWave < nz; j += 4 * kWave) {
                double a0 = __builtin_nontemporal_load(val + j), a1 = __builtin_nontemporal_load(val + j + kWave),
                       a2 = __builtin_nontemporal_load(val + j + 2 * kWave), a3 = __builtin_nontemporal_load(val + j + 3 * kWave);
                int c0 = __builtin_nontemporal_load(col + j), c1 = __builtin_nontemporal_load(col + j + kWave),
                    c2 = __builtin_nontemporal_load(col + j + 2 * kWave), c3 = __builtin_nontemporal_load(col + j + 3 * kWave);
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const double *__restrict__ g = epi.gv[v];
                    double g0 = g[c0], g1 = g[c1], g2 = g[c2], g3 = g[c3];
                    s[v] += term<Epi>(a0, g0);
                    s[v] += term<Epi>(a1, g1);
                    s[v] += term<Epi>(a2, g2);
                    s[v] += term<Epi>(a3, g3);
                }
            }
            if (j < nz) {
                // the tail: at most three more entries per lane (j + 3 * kWave >= nz here), in ONE clamped, masked step with all loads
                // in flight.  It was a loop of one load + one dependent gather per trip: a row of 65..255 entries never enters the
                // loops above and paid three serial round trips to memory (192-entry rows of a dense-block matrix: the whole launch
                // 0.166 -> 0.129 ms; rows of 1000-3000: 3-4 %).  Same additions in the same order.  (Measured and not kept, same box:
                // wave-uniform trips with a three- or four-wide tail: 0.139-0.148 ms on the 192-entry rows, 3-5 % ahead on rows of
                // 1000-3000.)
                const int last = nz - 1;
                const int q1 = min(j + kWave, last), q2 = min(j + 2 * kWave, last);
                const double a0 = val[j], a1 = val[q1], a2 = val[q2];
                const int c0 = col[j], c1 = col[q1], c2 = col[q2];
                const bool m1 = j + kWave < nz, m2 = j + 2 * kWave < nz;
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const double *__restrict__ g = epi.gv[v];
                    const double g0 = g[c0], g1 = g[c1], g2 = g[c2];
                    s[v] += term<Epi>(a0, g0);
                    if (m1) s[v] += term<Epi>(a1, g1);
                    if (m2) s[v] += term<Epi>(a2, g2);
                }
            }
#pragma unroll
            for (int v = 0; v < NV; ++v) s[v] = wave_sum(s[v]);
            if (lane == 0) {
                if (nr == 0) {
                    // chunk of a row longer than kSplitRow: r0 is the chunk slot; k_long_finish adds the
                    // chunk sums in order and runs the epilogue for the row
#pragma unroll
                    for (int v = 0; v < NV; ++v) A.long_partial[static_cast<size_t>(r0) * 2 + v] = s[v];
                } else {
                    typename Epi::Row rw = epi.load_row(r0);
                    epi.apply(r0, rw, s, acc);
                }
            }
        } else {
            // ---- stream mode
            typename Epi::Row rw;
            int rs = 0, re = 0;
            if (lane < nr) {
                rw = epi.load_row(r0 + lane);
                rs = A.rowptr[r0 + lane] - k0;
                re = A.rowptr[r0 + lane + 1] - k0;
            }
            if (nz > 0) {
                const int last = nz - 1;
                for (int base = 0; base < nz; base += 4 * kWave) {
                    // unconditional (clamped) loads keep four independent load chains in flight
                    const int j0 = base + lane, j1 = j0 + kWave, j2 = j0 + 2 * kWave, j3 = j0 + 3 * kWave;
                    const int q0 = min(j0, last), q1 = min(j1, last), q2 = min(j2, last), q3 = min(j3, last);
                    // the matrix is read once per launch: nontemporal loads keep it from evicting the
                    // gathered vector from L2 (measured +8 % on the 200M-nnz banded matrix)
                    // a matrix that fits in the L2s is read with the default policy so that it stays there
                    double a0, a1, a2, a3;
                    int c0, c1, c2, c3;
                    if (A.nt) {
                        a0 = __builtin_nontemporal_load(val + q0), a1 = __builtin_nontemporal_load(val + q1);
                        a2 = __builtin_nontemporal_load(val + q2), a3 = __builtin_nontemporal_load(val + q3);
                        c0 = __builtin_nontemporal_load(col + q0), c1 = __builtin_nontemporal_load(col + q1);
                        c2 = __builtin_nontemporal_load(col + q2), c3 = __builtin_nontemporal_load(col + q3);
                    } else {
                        a0 = val[q0], a1 = val[q1], a2 = val[q2], a3 = val[q3];
                        c0 = col[q0], c1 = col[q1], c2 = col[q2], c3 = col[q3];
                    }
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
                        const double *__restrict__ g = epi.gv[v];
                        double g0 = g[c0], g1 = g[c1], g2 = g[c2], g3 = g[c3];
                        // j3 < base + 256 <= kStreamW whenever base < nz <= kStreamW: in-bounds stores
                        lds[wave][v][j0] = term<Epi>(a0, g0);
                        lds[wave][v][j1] = term<Epi>(a1, g1);
                        lds[wave][v][j2] = term<Epi>(a2, g2);
                        lds[wave][v][j3] = term<Epi>(a3, g3);
                    }
                }
            }
            wave_lds_sync();
            if (lane < nr) {
                double s[NV];
#pragma unroll
                for (int v = 0; v < NV; ++v) s[v] = 0.0;
                // CSR order, eight staged products at a time: the reads of a batch are independent (one LDS latency per
                // batch), only the additions form the chain.  One read and one wait per element made a 200-entry row of the
                // config-3 matrix an 8 us chain -- the whole x-half launch (11.1 -> 5 us).
                int j = rs;
                for (; j + 8 <= re; j += 8) {
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
                        double p[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) p[u] = lds[wave][v][j + u];
#pragma unroll
                        for (int u = 0; u < 8; ++u) s[v] += p[u];
                    }
                }
                if (j < re) {
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
                        double p[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) p[u] = lds[wave][v][min(j + u, re - 1)];
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            if (j + u < re) s[v] += p[u];
                    }
                }
                epi.apply(r0 + lane, rw, s, acc);
            }
        }
    }
    if constexpr (NACC > 0) block_store_partials<NACC>(acc, epi.partials, epi.stride);
}


// ------------------------------------------------------------------------------------------------
// the column-tiled fused kernel (format and rationale: tiled.h).  One workgroup = one super-block of
// kTileRows rows; per tile step: [barrier] stage the prefetched tile of the gathered vector into LDS,
// issue the loads of the next step, [barrier], every lane folds its chunk of 4 entries into the LDS
// accumulators (all LDS reads first, running segment sums in registers, then the writes).  The
// barriers order LDS only (s_waitcnt lgkmcnt(0); s_barrier): __syncthreads() would also drain vmcnt
// and expose the latency of the prefetched global loads in every step.  Super-blocks are mapped
// XCD-aware (contiguous range per XCD) and the workgroups are persistent (the resident number per
// XCD, each taking every slots-th super-block of the range), so that the workgroups of one XCD work
// on consecutive super-blocks in step; every sweep starts at its rotation point (finish_schedule in
// tiled_build.hip) so that they stage the same vector tile at the same time.  Summation order per
// row: tiles ascending from the rotation point, then the tiles below it, remainder entries last --
// fixed by the matrix, so results are reproducible run to run.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// field of sweep step r (uniform, 0..255) from the four table registers of a lane (lane l holds steps l, 64 + l, ...):
// selects on a uniform condition, then one v_readlane -- no control flow
__device__ __forceinline__ int step_field(int a0, int a1, int a2, int a3, int r) {
    const int q = r >> 6;
    const int v = q == 0 ? a0 : (q == 1 ? a1 : (q == 2 ? a2 : a3));
    return __builtin_amdgcn_readlane(v, r & 63);
}

// Sweep of `count` consecutive tile steps of a super-block, starting at position `first` of its cyclic step list
// [s0, s0 + nst): every step stages its tile of the gathered vector in `ytile` and folds its entries into `acc`.
// Called by all kTileThreads threads of the workgroup; acc must be initialised and visible (a barrier comes first).
// ED / TD: how many steps ahead the entry loads (HBM) and the tile loads (L2) are issued.  Full chip (two workgroups on
// every CU): <2, 1> -- deeper gains nothing there, the fabric is busy (profiles/r02_pmc_summary.md).  Few workgroups
// (the split form): a step cannot be shorter than the HBM latency / ED, so <3, 2>.
#ifndef HPRLP_PB_PHASE_STAMPS
#define HPRLP_PB_PHASE_STAMPS 0  // developer builds: wall-clock time of k_pb_fused's phases per workgroup (TiledDev::wgtimes, HPRLP_WG_TIMES=1 HPRLP_PB_STAMPS=1)
#endif
#ifndef HPRLP_SWEEP_ED
#define HPRLP_SWEEP_ED 2  // fused kernel: entry loads issued this many steps ahead ...
#define HPRLP_SWEEP_TD 1  // ... and tile loads this many
#endif
// Cache policy of the half-step epilogues (XEpi / YEpi).  The streams a row's update reads and writes once per launch (x, c, l, u,
// last_x / y, AL, AU, last_y) must not push the gathered vector's tiles out of the 4 MiB L2s: with default-policy accesses the
// tile misses of a config-5 launch were three times those of a launch without epilogue (TCC counters,
// profiles/r03_tiled_decomposition.md).  1: those loads and the store of x nontemporal; 2 (default): also the store of the
// published vector (x_hat / y); 3: also the hand-off's scattered stores into P (slower: measured).  Same-box A/B/A/B on config 5:
// 723 / 731 it/s (0) -> 735-741 (1) -> 752-761 (2) -> 722-735 (3).
#ifndef HPRLP_EPI_NT
#define HPRLP_EPI_NT 2
#endif

template <int ED, int TD, bool REP, bool STAMP = false, bool LOGTERM = false, int TC = kTileCols>
__device__ __forceinline__ void tiled_sweep(const TiledDev &t, int s0, int nst, int first, int count, const double *__restrict__ vec,
                                            int ncols, double *acc, double *ytile, int tid, unsigned long long *stamp = nullptr) {
    constexpr int NT = kTileThreads, R = kTileRows, K = kTileChunk;  // (R: the codes' row field, whatever the copy's height)
    // TC: columns per tile of the copy (tiled.h: kTileCols, or kTileColsNarrow -- its own instantiation: a run-time trip count of
    // the staging loops costs the sweep its counted waits, 0.69 -> 0.87 ms on config 5)
    constexpr int TPT = TC / NT;
    static_assert(TC == kTileCols || TC == kTileColsNarrow, "tile width");
    typedef double d2_t __attribute__((ext_vector_type(2)));
    const int lane = tid & 63;
    auto sidx = [&](int i) { i = min(i, count - 1) + first; return s0 + (i < nst ? i : i - nst); };
    // The step table of the sweep lives in registers: lane l of every wave holds steps seg + l, seg + 64 + l, ... of
    // the current segment of 256 sweep positions; a step's fields come out by v_readlane with the (uniform) sweep
    // position.  The table loads leave the per-step path and the in-order vector-memory queue; a sweep longer than
    // 256 steps (rare) restarts the pipeline per segment.
    int seg = 0, lim = 0;
    int tc0 = 0, tc1 = 0, tc2 = 0, tc3 = 0, tb0 = 0, tb1 = 0, tb2 = 0, tb3 = 0, te0 = 0, te1 = 0, te2 = 0, te3 = 0;  // col0 / e_begin / e_end
    auto getstep = [&](int k, int &col0, int &eb, int &ee) {
        const int r = min(k, lim - 1) - seg;
        col0 = step_field(tc0, tc1, tc2, tc3, r);
        eb = step_field(tb0, tb1, tb2, tb3, r);
        ee = step_field(te0, te1, te2, te3, r);
    };
    struct Ent {
        d2_t va, vb;
        uint32_t i0, i1, i2;  // four 24-bit entry codes in three words (tiled.h)
    };
    typedef uint32_t u3_t __attribute__((ext_vector_type(3), aligned(4)));  // one 12-byte load per lane
    auto issue_entries = [&](Ent &E, int k) {
        int col0, eb, ee_;
        getstep(k, col0, eb, ee_);
        const int e = eb + K * tid;
        const int ee = (e < ee_) ? e : eb;
        E.va = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(t.tval + ee));
        E.vb = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(t.tval + ee) + 1);
        const u3_t w = __builtin_nontemporal_load(reinterpret_cast<const u3_t *>(t.tidx3 + (ee / K) * 3));
        E.i0 = w.x;
        E.i1 = w.y;
        E.i2 = w.z;
    };
    // tile loads and LDS stores in pairs (one 16-byte access per lane instead of two 8-byte ones: the vector-memory issue
    // of a step is what stalls when two workgroups share a CU, profiles/r02_pmc_summary.md).  Lane pair index p covers
    // columns col0 + 2 p, col0 + 2 p + 1; behind the end of the vector the last valid pair is read (never referenced), and
    // the lane that holds the last column of an odd-length vector takes it from the pair's second half.
    typedef double d2u_t __attribute__((ext_vector_type(2), aligned(8)));
    auto issue_tile = [&](double (&tl)[TPT], int k) {
        int col0, eb, ee_;
        getstep(k, col0, eb, ee_);
        static_assert(TPT % 2 == 0, "tile loads come in pairs");
#pragma unroll
        for (int j = 0; j < TPT / 2; ++j) {
            const int p = col0 + 2 * (tid + j * NT);
            const int q = min(p, ncols - 2);
            d2u_t v = *reinterpret_cast<const d2u_t *>(vec + q);
            if (q != p) v.x = v.y;
            tl[2 * j] = v.x;
            tl[2 * j + 1] = v.y;
        }
    };
    auto store_tile = [&](const double (&tl)[TPT]) {
#pragma unroll
        for (int j = 0; j < TPT / 2; ++j) {
            d2_t v;
            v.x = tl[2 * j];
            v.y = tl[2 * j + 1];
            reinterpret_cast<d2_t *>(ytile)[tid + j * NT] = v;
        }
    };
    auto process = [&](const Ent &E, int k) {
        int col0, eb, ee_;
        getstep(k, col0, eb, ee_);
        if (K * tid < ee_ - eb) {
            const double v[K] = {E.va.x, E.va.y, E.vb.x, E.vb.y};
            const uint32_t c0 = E.i0, c1 = E.i1, c2 = E.i2;
            const uint32_t id[K] = {c0 & 0xffffffu, (c0 >> 24) | ((c1 & 0xffffu) << 8), (c1 >> 16) | ((c2 & 0xffu) << 16), c2 >> 8};
            uint32_t rw[K];
            double y[K];
#pragma unroll
            for (int k2 = 0; k2 < K; ++k2) {
                rw[k2] = id[k2] & (R - 1);
                y[k2] = ytile[id[k2] >> kTileRowBits];
            }
            double a[K];
#pragma unroll
            for (int k2 = 0; k2 < K; ++k2) a[k2] = acc[rw[k2]];
            double sk[K], pr[K];
#pragma unroll
            for (int k2 = 0; k2 < K; ++k2) pr[k2] = LOGTERM ? (v[k2] != v[k2] ? 0.0 : v[k2] - y[k2]) : v[k2] * y[k2];
            sk[0] = a[0] + pr[0];
#pragma unroll
            for (int k2 = 1; k2 < K; ++k2) sk[k2] = ((rw[k2] == rw[k2 - 1]) ? sk[k2 - 1] : a[k2]) + pr[k2];
#pragma unroll
            for (int k2 = 0; k2 < K; ++k2)
                if (k2 == K - 1 || rw[k2] != rw[k2 + 1]) acc[rw[k2]] = sk[k2];
        }
    };
    // REP (matrices with tiles of more than kTileStepCap entries, i.e. several steps per tile): a step whose tile is the
    // previous step's neither loads nor stages it again -- one barrier (the accumulators change owners between steps).
    // A separate instantiation: the extra control flow costs the common case its counted waits.
    auto same_tile = [&](int k) {
        if (k <= seg || k >= lim) return false;  // first step of a segment always stages; clamped look-ahead never loads
        int c0, c1, eb, ee_;
        getstep(k, c0, eb, ee_);
        getstep(k - 1, c1, eb, ee_);
        return c0 == c1;
    };
    // STAMP (diagnostic instantiation only, HPRLP_TILE_STAMPS=1): shader-clock time of wave 0 in the five phases of a step
    unsigned long long tq[6] = {0, 0, 0, 0, 0, 0}, ph[7] = {0, 0, 0, 0, 0, 0, 0};  // sums stay in registers until the sweep ends
    auto step = [&](Ent &E, double (&tl)[TPT], int k) {
        if (STAMP) tq[0] = __builtin_amdgcn_s_memtime();
        lds_barrier();  // every lane is done with the previous step (tile and accumulators)
        if (STAMP) tq[1] = __builtin_amdgcn_s_memtime();
        if (STAMP) {
            store_tile(tl);
            tq[2] = __builtin_amdgcn_s_memtime();  // tile data arrived and written
            issue_tile(tl, k + TD);
            tq[3] = __builtin_amdgcn_s_memtime();
            lds_barrier();
            tq[4] = __builtin_amdgcn_s_memtime();
            process(E, k);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            tq[5] = __builtin_amdgcn_s_memtime();
            issue_entries(E, k + ED);
            const unsigned long long t6 = __builtin_amdgcn_s_memtime();
            ph[0] += tq[1] - tq[0];  // barrier A (waiting for the slowest wave of the previous step)
            ph[1] += tq[2] - tq[1];  // wait for the tile loads + LDS stores
            ph[2] += tq[3] - tq[2];  // issue of the next tile's loads
            ph[3] += tq[4] - tq[3];  // barrier B
            ph[4] += tq[5] - tq[4];  // entries arrived + LDS gather / accumulate
            ph[5] += t6 - tq[5];     // issue of the entry loads
            ph[6] += 1;
            return;
        }
        if (REP) {
            const bool keep = same_tile(k);  // uniform
            if (!keep) {
                store_tile(tl);
            }
            if (!same_tile(k + TD)) issue_tile(tl, k + TD);
            if (!keep) lds_barrier();  // tile visible
        } else {
            store_tile(tl);
            issue_tile(tl, k + TD);
            lds_barrier();  // tile visible
        }
        process(E, k);
        issue_entries(E, k + ED);
    };
    for (seg = 0; seg < count; seg += 256) {
        lim = min(seg + 256, count);
        {
            const TileStep q0 = t.steps[sidx(seg + lane)], q1 = t.steps[sidx(seg + 64 + lane)];
            const TileStep q2 = t.steps[sidx(seg + 128 + lane)], q3 = t.steps[sidx(seg + 192 + lane)];
            tc0 = q0.col0; tb0 = q0.e_begin; te0 = q0.e_end;
            tc1 = q1.col0; tb1 = q1.e_begin; te1 = q1.e_end;
            tc2 = q2.col0; tb2 = q2.e_begin; te2 = q2.e_end;
            tc3 = q3.col0; tb3 = q3.e_begin; te3 = q3.e_end;
        }
        // prologue in the steady state's issue order (step j issues tile j + TD, then entries j + ED); the scheduling
        // barriers keep the compiler from interleaving the groups, which would force the loop's waits down to vmcnt(0)
        Ent E[ED];
        double tl[TD][TPT];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = -(ED > TD ? ED : TD); j < 0; ++j) {
            if (j + TD >= 0) {
                issue_tile(tl[(j + TD) % TD], seg + j + TD);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (j + ED >= 0) {
                issue_entries(E[(j + ED) % ED], seg + j + ED);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // unrolled over the register sets (and twice that for <2, 1>): the compiler's counted waits are only conservative
        // in the first step behind the loop header, where the prologue and the back edge merge
        constexpr int U = (ED * TD) % 2 == 0 && ED * TD > 2 ? ED * TD : 2 * ED * TD;
        static_assert(U % ED == 0 && U % TD == 0, "the unrolled body must return every register set to its role");
        bool done = false;
        for (int k = seg; !done;) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                step(E[u % ED], tl[u % TD], k);
                if (++k >= lim) {
                    done = true;
                    break;
                }
            }
        }
    }
    if (STAMP && tid == 0) {
#pragma unroll
        for (int i = 0; i < 7; ++i) stamp[i] += ph[i];
    }
}

// Remainder steps [smid, s1) of a super-block (propagation blocking, tiled.h): the products were written into this
// super-block's slice of P -- by k_far_products, or by the other half-step's epilogue (hand-off) -- in (source group) order; the
// step's codes (slot in the step << 16 | local row) are in (row, CSR) order.  A step: the products go to LDS (`prod`, K per
// lane, coalesced); lane l takes entries K l .. K l + K - 1 of the (row, CSR) order -- its codes come straight from memory into
// registers -- gathers their products from LDS and adds them up per row: rows that begin and end inside the chunk on the
// spot, the run at the chunk's start and the run at its end through a segmented scan over the lanes of the wave (a row's
// pieces are the tail of one lane, whole lanes, the head of a last one), a wave's last run through LDS after the barrier
// (`bnd`: its first lane takes no carry from the wave before).  All lanes work, three LDS barriers per step, a row of any
// length costs the same per entry.  (Until round 4 ONE head lane per row read code and product alternately -- two dependent LDS
// reads per entry: a matrix without column locality, 20 entries per row all in here, spent 0.28 ms on 4e7 entries.)
// Per-row order of the additions: fixed by the matrix (chunks in order, the scan's tree inside a wave).
// bnd: kTileThreads / 64 doubles of LDS for the values, then as many 32-bit words for the rows.
template <int K>
__device__ __forceinline__ void remainder_steps(const TiledDev &t, int smid, int s1, double *acc, double *prod, double *bnd, int tid) {
    constexpr int NT = kTileThreads, NW = kTileThreads / 64;
    static_assert(K == 4 || K == 6 || K == 8, "code loads: one or two 16-byte loads, or two 12-byte loads, per lane");
    if (smid >= s1) return;
    const int wave = tid >> 6, lane = tid & 63;
    double *bnd_val = bnd;
    uint32_t *bnd_row = reinterpret_cast<uint32_t *>(bnd + NW);
    constexpr uint32_t NOROW = 0x10000u;  // (codes keep the row in 16 bits)
    // the loads of step s + 1 are in flight while step s is folded
    double pv[K];
    uint32_t cv[K];
    TileStep st = t.steps[smid];
    auto issue = [&](const TileStep &z) {
        const int last = max(z.e_end - z.e_begin - 1, 0);
#pragma unroll
        for (int k = 0; k < K; ++k) pv[k] = __builtin_nontemporal_load(t.P + z.e_begin + min(tid + k * NT, last));  // clamped: branch-free
        // the lane's K consecutive codes in wide loads (4-byte aligned); behind the step's end the last whole chunk is read
        const int c0 = min(K * tid, max(last + 1 - K, 0));
        if constexpr (K == 6) {
            typedef uint32_t u3_t __attribute__((ext_vector_type(3), aligned(4)));
            const u3_t w0 = __builtin_nontemporal_load(reinterpret_cast<const u3_t *>(t.rq + z.e_begin + c0));
            const u3_t w1 = __builtin_nontemporal_load(reinterpret_cast<const u3_t *>(t.rq + z.e_begin + c0 + 3));
            cv[0] = w0.x; cv[1] = w0.y; cv[2] = w0.z; cv[3] = w1.x; cv[4] = w1.y; cv[5] = w1.z;
        } else {
            typedef uint32_t u4_t __attribute__((ext_vector_type(4), aligned(4)));
            const u4_t w0 = __builtin_nontemporal_load(reinterpret_cast<const u4_t *>(t.rq + z.e_begin + c0));
            cv[0] = w0.x; cv[1] = w0.y; cv[2] = w0.z; cv[3] = w0.w;
            if constexpr (K == 8) {
                const u4_t w1 = __builtin_nontemporal_load(reinterpret_cast<const u4_t *>(t.rq + z.e_begin + c0 + 4));
                cv[4] = w1.x; cv[5] = w1.y; cv[6] = w1.z; cv[7] = w1.w;
            }
        }
    };
    issue(st);
    for (int s = smid; s < s1; ++s) {
        const int cnt = st.e_end - st.e_begin;
        const TileStep nxt = t.steps[min(s + 1, s1 - 1)];
        lds_barrier();  // everybody is done with the previous step's products and row sums
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int el = tid + k * NT;
            if (el < cnt) prod[el] = pv[k];
        }
        // (a lane whose chunk crosses the step's end loaded the LAST whole chunk instead: move its codes down)
        uint32_t code[K];
        {
            const int shift = K * tid - min(K * tid, max(cnt - K, 0));  // 0 for every lane but at most one
#pragma unroll
            for (int u = 0; u < K; ++u) code[u] = cv[u];
            if (shift > 0 && shift < K) {
#pragma unroll
                for (int u = 0; u < K; ++u) {
                    uint32_t c = 0;
#pragma unroll
                    for (int q = 0; q < K; ++q) c = (q == u + shift) ? cv[q] : c;
                    code[u] = c;
                }
            }
        }
        if (s + 1 < s1) issue(nxt);
        lds_barrier();
        // this lane's chunk, sorted by row.  Entries behind the step's end continue the last valid row with a zero product; a
        // lane without any entry holds NOROW.
        const int nv = min(max(cnt - K * tid, 0), K);
        uint32_t row[K];
        double pr[K];
#pragma unroll
        for (int u = 0; u < K; ++u) {
            const bool ok = u < nv;
            row[u] = ok ? (code[u] & 0xffffu) : (u == 0 ? NOROW : row[u - 1]);
            pr[u] = ok ? prod[code[u] >> 16] : 0.0;
        }
        // in-lane: the run that starts at entry 0 (head), rows that begin and end inside the chunk (added on the spot: nobody
        // else holds entries of theirs in this step), the run that ends at entry K - 1 (tail)
        double run = pr[0], head = 0.0;
        bool single = true;
#pragma unroll
        for (int u = 1; u < K; ++u) {
            if (row[u] != row[u - 1]) {
                if (single) {
                    head = run;
                    single = false;
                } else {
                    acc[row[u - 1]] += run;
                }
                run = pr[u];
            } else {
                run += pr[u];
            }
        }
        const uint32_t rf = row[0], rl = row[K - 1];
        // segmented inclusive scan of the lanes' carry-outs (the tail, or everything for a one-row lane); a lane passes the
        // carry on iff it holds one row and continues its predecessor's
        const uint32_t rl_prev = __shfl_up(rl, 1, 64);
        const bool open_in = lane > 0 && rf == rl_prev && rf != NOROW;
        double v = run;
        int f = (single && open_in) ? 0 : 1;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const double vu = __shfl_up(v, off, 64);
            const int fu = __shfl_up(f, off, 64);
            if (lane >= off && !f) {
                v += vu;
                f |= fu;
            }
        }
        const double v_prev = __shfl_up(v, 1, 64);
        const double carry = open_in ? v_prev : 0.0;
        const int next_open = __shfl_down(static_cast<int>(open_in), 1, 64);
        if (!single) acc[rf] += carry + head;  // the head's row ends in this lane
        const double tail = single ? carry + run : run;
        if (lane == 63) {
            // the wave's last run may go on in the next wave (whose first lane takes no carry): added after the barrier
            bnd_row[wave] = rl;
            bnd_val[wave] = tail;
        } else if (!next_open && rl != NOROW) {
            acc[rl] += tail;
        }
        lds_barrier();
        if (tid < NW) {
            // the waves' last runs, one lane each; a row that fills whole waves appears more than once: the lane of its last
            // appearance adds them all, in wave order
            const uint32_t rw = bnd_row[tid];
            bool last_one = rw != NOROW;
            double tot = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const bool same = bnd_row[w] == rw;
                if (w <= tid && same) tot += bnd_val[w];
                if (w > tid && same) last_one = false;
            }
            if (last_one) acc[rw] += tot;
        }
        st = nxt;
    }
}

// Epilogue of a super-block (k_tiled_fused, k_pb_fused): row i of the block has its sum in acc[i]; the half-step / residual /
// dot epilogue runs on it, and with PUSH the value the half-step publishes replaces the sum (the hand-off reads it from there).
// Four rows per lane at a time, all their loads before the first update: one trip to memory per four rows instead of one
// per row (the update's stores may alias the next row's loads as far as the compiler knows, so it kept them in order:
// a 1984-row super-block of a mid-size matrix spent 8 us of its 96 in four such trips).
template <class Epi, bool PUSH>
__device__ __forceinline__ void epilogue_rows(const Epi &epi, double *acc, double (&racc)[Epi::NACC > 0 ? Epi::NACC : 1], int r0, int nr, int tid) {
    constexpr int NT = kTileThreads, B = 4;
    for (int i0 = tid; i0 < nr; i0 += B * NT) {
        typename Epi::Row rw[B];
        double sv[B][1];
#pragma unroll
        for (int u = 0; u < B; ++u) {
            const int i = min(i0 + u * NT, nr - 1);  // clamped: branch-free loads, surplus lanes skip the update
            rw[u] = epi.load_row(r0 + i);
            sv[u][0] = acc[i];
        }
#pragma unroll
        for (int u = 0; u < B; ++u) {
            const int i = i0 + u * NT;
            if (i < nr) {
                if constexpr (PUSH) acc[i] = epi.apply(r0 + i, rw[u], sv[u], racc);  // the published value replaces the row sum
                else epi.apply(r0 + i, rw[u], sv[u], racc);
            }
        }
    }
}

// Hand-off of one source group's remainder products (kernels.h: FarPush): entries [pb, pe) of the consumer's source-side lists,
// the group's fresh vector values in `vals` (LDS).  Four entries per lane and batch; the next batch's loads are in flight while
// this one's products are stored (one batch at a time left every batch a full trip to memory).
__device__ __forceinline__ void push_products(const FarPush &f, const double *vals, int pb, int pe, int tid) {
    constexpr int NT = kTileThreads, U = 4;
    if (pb >= pe) return;
    double a4[U];
    int p4[U];
    uint16_t c4[U];
    auto load_batch = [&](int k0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int z = min(k0 + u * NT, pe - 1);  // clamped: branch-free, surplus lanes are masked at the store
            a4[u] = __builtin_nontemporal_load(f.val + z);
            p4[u] = __builtin_nontemporal_load(f.pos + z);
            c4[u] = __builtin_nontemporal_load(f.lcol + z);
        }
    };
    int k = pb + tid;
    load_batch(k);
    for (; k < pe; k += U * NT) {
        double prd[U];
        int ps[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            prd[u] = a4[u] * vals[c4[u]];
            ps[u] = p4[u];
        }
        const int kn = k + U * NT;
        if (kn < pe) load_batch(kn);
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (k + u * NT < pe) f.P[ps[u]] = prd[u];
    }
}

// P position of entry k of a source group's list from the group's run table in LDS (tiled.h: f_rk / f_rp): the last run that
// starts at or before k.  Neighbouring lanes hold neighbouring entries -- mostly the same run: the reads are broadcasts.
__device__ __forceinline__ int run_position(const int *tab_k, const int *tab_p, int nr, int k) {
    int lo = 0, hi = nr - 1;  // tab_k[0] <= k
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab_k[mid] <= k) lo = mid;
        else hi = mid - 1;
    }
    return tab_p[lo] + (k - tab_k[lo]);
}

// push_products without reading f.pos: the consumer's run tables give the positions (all-remainder form: 18 instead of 22 bytes
// of traffic per entry on this side).  tab: 2 * kPbRunTabCap ints of LDS nobody else uses; g: the source group.
__device__ __forceinline__ void push_products_runs(const FarPush &f, const double *vals, int *tab, int g, int pb, int pe, int tid) {
    constexpr int NT = kTileThreads, U = 4;
    if (pb >= pe) return;  // (uniform)
    const int r0 = f.rptr[g], nr = f.rptr[g + 1] - r0;
    int *tab_k = tab, *tab_p = tab + kPbRunTabCap;
    double a4[U];
    uint16_t c4[U];
    auto load_batch = [&](int k0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int z = min(k0 + u * NT, pe - 1);  // clamped: branch-free, surplus lanes are masked at the store
            a4[u] = __builtin_nontemporal_load(f.val + z);
            c4[u] = __builtin_nontemporal_load(f.lcol + z);
        }
    };
    int k = pb + tid;
    load_batch(k);
    for (int i = tid; i < nr; i += NT) {
        tab_k[i] = f.rk[r0 + i];
        tab_p[i] = f.rp[r0 + i];
    }
    lds_barrier();
    // the positions of a batch are looked up while its loads are in flight (they depend on the entry index only)
    int ps[U];
#pragma unroll
    for (int u = 0; u < U; ++u) ps[u] = run_position(tab_k, tab_p, nr, min(k + u * NT, pe - 1));
    for (; k < pe; k += U * NT) {
        double prd[U];
        int pc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            prd[u] = a4[u] * vals[c4[u]];
            pc[u] = ps[u];
        }
        const int kn = k + U * NT;
        if (kn < pe) load_batch(kn);
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (k + u * NT < pe) f.P[pc[u]] = prd[u];
        if (kn < pe) {
#pragma unroll
            for (int u = 0; u < U; ++u) ps[u] = run_position(tab_k, tab_p, nr, min(kn + u * NT, pe - 1));
        }
    }
}

template <class Epi, bool REP, bool PUSH = false, bool NARROW = false>
__global__ void __launch_bounds__(kTileThreads, 4) k_tiled_fused(CsrDev A, Epi epi) {  // 4 waves per SIMD = two workgroups per CU
    static_assert(Epi::NV == 1, "the tiled kernel stages one gathered vector");
    constexpr int NT = kTileThreads, RMAX = kTileRows, T = kTileCols;
    constexpr int NACC = Epi::NACC;
    __shared__ double acc[RMAX];
    __shared__ __attribute__((aligned(16))) double ytile[T];
    const TiledDev &t = A.tiled;
    const int tid = threadIdx.x;
    const int R = t.R;  // rows per super-block of this copy (<= RMAX, tiled.h)
    const int per = t.per, slots = gridDim.x / 8;  // persistent: slot, slot + slots, ... of this XCD's range
    const int slot = blockIdx.x / 8;
    double racc[NACC > 0 ? NACC : 1];
#pragma unroll
    for (int i = 0; i < (NACC > 0 ? NACC : 1); ++i) racc[i] = 0.0;
    epi.begin();
    int wg_round = 0;  // HPRLP_WG_TIMES diagnostic
    const bool wg_stamp = t.wgtimes && tid == 0 && (t.wg_filter == 0 || (PUSH && NACC == 0));
    if (wg_stamp) t.wgtimes[blockIdx.x * 8] = wall_clock64();
    for (int q = slot; q < per; q += slots) {
        const int sb = (blockIdx.x % 8) * per + q;
        if (sb >= t.nsb) break;
        const double *__restrict__ vec = epi.gv[0];
        lds_barrier();  // the previous super-block's epilogue is done with acc
        const int ncols = A.cols;
        for (int i = tid; i < R; i += NT) acc[i] = 0.0;
        const int s0 = t.sb_ptr[sb], smid = t.sb_mid[sb], s1 = t.sb_ptr[sb + 1];
        if (s0 < smid) tiled_sweep<HPRLP_SWEEP_ED, HPRLP_SWEEP_TD, REP, false, LogTerm<Epi>::value, NARROW ? kTileColsNarrow : kTileCols>(t, s0, smid - s0, t.steps[s0].rot, smid - s0, vec, ncols, acc, ytile, tid);  // rotated sweep (tiled_build.hip, finish_schedule)
        remainder_steps<kTileRemK>(t, smid, s1, acc, ytile, ytile + kTileRemCap, tid);
        lds_barrier();
        const int r0 = sb * R;
        const int nr = min(R, A.rows - r0);
        // the hand-off's list bounds travel while the epilogue runs (a uniform load waited for on the spot is one more
        // exposed trip to memory per super-block)
        int pb = 0, pe = 0;
        if constexpr (PUSH) {
            pb = epi.push.gptr[sb];
            pe = epi.push.gptr[sb + 1];
        }
        epilogue_rows<Epi, PUSH>(epi, acc, racc, r0, nr, tid);
        if constexpr (PUSH) {
            // Hand-off (kernels.h: FarPush): this super-block's fresh values are the source group `sb` of the OTHER matrix'
            // remainder; write its products straight into that matrix' P -- what k_far_products would do in a launch of
            // its own after re-reading the vector from memory.  Same products bit for bit, same slots.
            // (a source group of the consumer's remainder = the rows of one super-block here: push_into() checks G == R)
            lds_barrier();
            push_products(epi.push, acc, pb, pe, tid);
        }
        if (wg_stamp && ++wg_round <= 5) t.wgtimes[blockIdx.x * 8 + wg_round] = wall_clock64();
    }
    if (wg_stamp) {
        t.wgtimes[blockIdx.x * 8 + 7] = wall_clock64();
        // where the workgroup ran: HW_ID (id 4; CU_ID bits 11:8, SH_ID 12, SE_ID 15:13) and XCC_ID (id 20) of wave 0
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        t.wgtimes[blockIdx.x * 8 + 6] = (static_cast<unsigned long long>(xcc & 0xfu) << 32) | hw;
    }
    if constexpr (NACC > 0) {
        // the tile buffer is the scratch of the block reduction: a separate array would push the workgroup past
        // 80 KiB of LDS and leave ONE workgroup per CU (measured: 0.92 instead of 0.73 ms per launch)
        __syncthreads();
        block_store_partials_in<NACC, kTileThreads / kWave>(racc, epi.partials, epi.stride, reinterpret_cast<double(*)[NACC]>(ytile));
    }
}

// ------------------------------------------------------------------------------------------------
// All-remainder form (round 4; tiled.h: kPbRemCap): the copy of a matrix without column locality stages no tile at all -- every
// entry's product arrives through P (written by the other half-step's epilogue, or by k_far_products) and a super-block's work is
// to add its slice of P row by row.  Its own kernel: LDS holds the accumulators of at most kPbRowsMax rows and steps of
// kPbRemCap = 4096 products with their codes (k_tiled_fused's remainder steps have the 16 KiB tile buffer: 1024), and a step's
// work is spread evenly over the lanes (remainder_steps<K>): lane t holds entries [t K, (t+1) K) of the step in registers, adds
// them per row, and a segmented scan over the wave (then over the waves' edge runs, through bnd) joins the rows that cross lanes.
// One head lane per row -- the first form -- left most lanes idle behind a few chains of dependent LDS reads (every row of such a
// matrix holds all its ~20 entries here: 0.28 ms for the 4e7 entries of a 2M x 2M matrix, three times the time its bytes take).
// Per-row order of the additions: as stored ((source group, CSR) order), grouped by lane and wave boundaries -- fixed by the
// matrix, so results are reproducible run to run.  Same persistent schedule, epilogue and hand-off as k_tiled_fused.
// ------------------------------------------------------------------------------------------------
template <class Epi, bool PUSH = false>
__global__ void __launch_bounds__(kTileThreads, 4) k_pb_fused(CsrDev A, Epi epi) {
    static_assert(Epi::NV == 1, "one gathered vector");
    constexpr int NT = kTileThreads, K = kPbRemK, CAP = kPbRemCap, NW = kTileThreads / 64;
    constexpr int NACC = Epi::NACC;
    __shared__ double acc[kPbRowsMax];
    __shared__ __attribute__((aligned(16))) double prod[CAP];
    __shared__ double bnd[3 * NW];
    const TiledDev &t = A.tiled;
    const int tid = threadIdx.x;
    const int R = t.R;
    const int per = t.per, slots = gridDim.x / 8;
    const int slot = blockIdx.x / 8;
    double racc[NACC > 0 ? NACC : 1];
#pragma unroll
    for (int i = 0; i < (NACC > 0 ? NACC : 1); ++i) racc[i] = 0.0;
    epi.begin();
#if HPRLP_PB_PHASE_STAMPS
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tq = wall_clock64(), tk0 = tq;
#define PB_STAMP(i) do { const unsigned long long now_ = wall_clock64(); ph[i] += now_ - tq; tq = now_; } while (0)
#else
#define PB_STAMP(i) do { } while (0)
#endif
    for (int q = slot; q < per; q += slots) {
        const int sb = (blockIdx.x % 8) * per + q;
        if (sb >= t.nsb) break;
        lds_barrier();  // the previous super-block's epilogue is done with acc
        for (int i = tid; i < R; i += NT) acc[i] = 0.0;
        const int smid = t.sb_mid[sb], s1 = t.sb_ptr[sb + 1];
        remainder_steps<K>(t, smid, s1, acc, prod, bnd, tid);
        lds_barrier();
        PB_STAMP(4);
        const int r0 = sb * R;
        const int nr = min(R, A.rows - r0);
        int pb = 0, pe = 0;
        if constexpr (PUSH) {
            pb = epi.push.gptr[sb];
            pe = epi.push.gptr[sb + 1];
        }
        epilogue_rows<Epi, PUSH>(epi, acc, racc, r0, nr, tid);
        PB_STAMP(5);
        if constexpr (PUSH) {
            // hand-off (kernels.h: FarPush), as in k_tiled_fused: this super-block's fresh values are source group `sb` of the
            // other matrix' lists
            lds_barrier();
            if (epi.push.rk) push_products_runs(epi.push, acc, reinterpret_cast<int *>(prod), sb, pb, pe, tid);  // (prod is free by now)
            else push_products(epi.push, acc, pb, pe, tid);
        }
        PB_STAMP(6);
    }
#if HPRLP_PB_PHASE_STAMPS
    if (t.wgtimes && tid == 0 && PUSH && NACC == 0) {
        ph[7] = wall_clock64() - tk0;
        for (int i = 0; i < 8; ++i) t.wgtimes[blockIdx.x * 8 + i] = ph[i];
    }
#endif
#undef PB_STAMP
    if constexpr (NACC > 0) {
        __syncthreads();
        block_store_partials_in<NACC, kTileThreads / kWave>(racc, epi.partials, epi.stride, reinterpret_cast<double(*)[NACC]>(prod));
    }
}

// ------------------------------------------------------------------------------------------------
// Piece form, for matrices with fewer super-blocks than the chip has workgroup slots (a 1/8 row shard of config 5 has 153,
// a 1/4 shard 306; a workgroup's sweep is a chain of dependent steps, so the chip only streams when every slot has one).
// The tile steps of all super-blocks, laid end to end, are cut into equal pieces (tiled_build.hip, finish_schedule); a
// workgroup sweeps the segments of its piece -- each a range of one super-block's steps -- into fresh accumulators and
// stores them (the super-block's last segment adds the remainder); k_tiled_finish adds a row's segments in order and
// runs the epilogue.  Per-row summation order: tiles ascending, remainder last -- fixed by the matrix and the piece count.
// ------------------------------------------------------------------------------------------------
template <bool REP, bool STAMP = false, bool NARROW = false>
__global__ void __launch_bounds__(kTileThreads, 4) k_tiled_part(CsrDev A, const double *__restrict__ vec) {
    constexpr int NT = kTileThreads, RMAX = kTileRows, T = kTileCols;
    __shared__ double acc[RMAX];
    __shared__ __attribute__((aligned(16))) double ytile[T];
    const TiledDev &t = A.tiled;
    const int tid = threadIdx.x;
    const int R = t.R;
    // XCD-aware: a contiguous range of pieces per XCD (neighbouring pieces read the same vector tiles)
    const int per = (t.n_pieces + 7) / 8;
    const int pc = (blockIdx.x % 8) * per + blockIdx.x / 8;
    if (pc >= t.n_pieces) return;
    for (int sg = t.piece_ptr[pc]; sg < t.piece_ptr[pc + 1]; ++sg) {
        const int4 d = t.segs[sg];
        const int sb = d.x;
        lds_barrier();  // the previous segment's store is done with acc
        for (int i = tid; i < R; i += NT) acc[i] = 0.0;
        const int s0 = t.sb_ptr[sb], smid = t.sb_mid[sb], s1 = t.sb_ptr[sb + 1];
        unsigned long long *st = STAMP ? t.stamps + static_cast<size_t>(pc) * 16 : nullptr;
        unsigned long long c0 = 0, c1 = 0, c2 = 0;
        if (STAMP) c0 = __builtin_amdgcn_s_memtime();
        if (d.z > 0) tiled_sweep<3, 2, REP, STAMP, false, NARROW ? kTileColsNarrow : kTileCols>(t, s0, smid - s0, d.y, d.z, vec, A.cols, acc, ytile, tid, st);
        if (STAMP) c1 = __builtin_amdgcn_s_memtime();
        if (d.w) remainder_steps<kTileRemK>(t, smid, s1, acc, ytile, ytile + kTileRemCap, tid);
        lds_barrier();
        if (STAMP) c2 = __builtin_amdgcn_s_memtime();
        double *out = t.parts + static_cast<size_t>(sg) * R;
        for (int i = tid; i < R; i += NT) out[i] = acc[i];
        if (STAMP && tid == 0) {
            st[8] += c1 - c0;   // whole sweep (table loads and pipeline fill included)
            st[9] += c2 - c1;   // remainder
            st[10] += __builtin_amdgcn_s_memtime() - c2;  // issue of the partial-sum stores
        }
    }
}

template <class Epi>
__global__ void __launch_bounds__(kThreads) k_tiled_finish(CsrDev A, Epi epi) {
    constexpr int NACC = Epi::NACC;
    const int R = A.tiled.R;
    double racc[NACC > 0 ? NACC : 1];
#pragma unroll
    for (int i = 0; i < (NACC > 0 ? NACC : 1); ++i) racc[i] = 0.0;
    epi.begin();
    const int r = blockIdx.x * kThreads + threadIdx.x;
    if (r < A.rows) {
        const int sb = r / R;
        const int a = A.tiled.slot_ptr[sb], b = A.tiled.slot_ptr[sb + 1];
        const double *p = A.tiled.parts + (r - sb * R);
        double sum = 0.0;
        for (int k = a; k < b; ++k) sum = (k == a) ? p[static_cast<size_t>(k) * R] : sum + p[static_cast<size_t>(k) * R];
        const double sv[1] = {sum};
        typename Epi::Row rw = epi.load_row(r);
        epi.apply(r, rw, sv, racc);
    }
    if constexpr (NACC > 0) block_store_partials<NACC>(racc, epi.partials, epi.stride);
}

__global__ void __launch_bounds__(kThreads) k_tiled_refresh(long n, const int *perm, const double *csr_val, double *out) {
    const long i = static_cast<long>(blockIdx.x) * kThreads + threadIdx.x;
    if (i < n) {
        const int p = perm[i];
        out[i] = p >= 0 ? csr_val[p] : 0.0;
    }
}

void launch_tiled_refresh(const DeviceTiled &t, const double *csr_val, hipStream_t s) {
    if (t.n_tile > 0)
        hipLaunchKernelGGL(k_tiled_refresh, dim3(static_cast<unsigned>((t.n_tile + kThreads - 1) / kThreads)), dim3(kThreads), 0, s,
                           t.n_tile, t.tperm.p, csr_val, t.tval.p);
    if (t.n_rem > 0 && t.f_val.p)
        hipLaunchKernelGGL(k_tiled_refresh, dim3(static_cast<unsigned>((t.n_rem + kThreads - 1) / kThreads)), dim3(kThreads), 0, s,
                           t.n_rem, t.f_perm.p, csr_val, t.f_val.p);
}

// the Curtis-Reid passes' values of a tiled copy: -log|a| (as CrEpi::term forms it), NaN in the padding slots
__global__ void __launch_bounds__(kThreads) k_tiled_refresh_log(long n, const int *perm, const double *csr_val, double *out) {
    const long i = static_cast<long>(blockIdx.x) * kThreads + threadIdx.x;
    if (i < n) {
        const int p = perm[i];
        out[i] = p >= 0 ? -log(fmax(fabs(csr_val[p]), 1e-300)) : __builtin_nan("");
    }
}

void launch_tiled_refresh_log(const DeviceTiled &t, const double *csr_val, double *tval_log, double *fval_log, hipStream_t s) {
    if (t.n_tile > 0)
        hipLaunchKernelGGL(k_tiled_refresh_log, dim3(static_cast<unsigned>((t.n_tile + kThreads - 1) / kThreads)), dim3(kThreads), 0, s,
                           t.n_tile, t.tperm.p, csr_val, tval_log);
    if (t.n_rem > 0 && t.f_val.p)
        hipLaunchKernelGGL(k_tiled_refresh_log, dim3(static_cast<unsigned>((t.n_rem + kThreads - 1) / kThreads)), dim3(kThreads), 0, s,
                           t.n_rem, t.f_perm.p, csr_val, fval_log);
}

// What a workgroup of the pre-pass works on: source group g, entries [b, e) of its list -- the whole group, or one chunk of a
// heavy group (tiled.h: f_work).  Workgroups are dealt round-robin over the 8 XCDs: every XCD gets a contiguous range of groups
// (work items), so that the runs its workgroups write side by side in P (layout [super-block][group]) meet in ONE L2 and leave
// it as whole lines.
__device__ __forceinline__ bool far_work_item(const TiledDev &t, int &g, int &b, int &e) {
    const int items = t.f_work ? t.n_work : t.n_groups;
    const int per = (items + 7) / 8;
    const int w = (blockIdx.x % 8) * per + blockIdx.x / 8;
    if (w >= items) return false;
    if (t.f_work) {
        const int4 d = t.f_work[w];
        g = d.x; b = d.y; e = d.z;
    } else {
        g = w; b = t.f_gptr[w]; e = t.f_gptr[w + 1];
    }
    return b < e;
}

// Pre-pass of a tiled launch (tiled.h): one workgroup per group of kFarGroup columns of the gathered vector.  The
// group's slice is staged in LDS with coalesced loads; the workgroup streams its remainder entries (value, local
// column, position in P -- ascending, so the stores of one (group, super-block) run are contiguous) and writes the
// products.  Every gathered element is read from memory once, no 128-byte line is fetched for 8 bytes.
template <bool LOGTERM = false>
__global__ void __launch_bounds__(kFarThreads) k_far_products(TiledDev t, const double *__restrict__ vec, int ncols) {
    __shared__ double v[kFarGroup];
    const int G = t.G;  // columns per source group (at most kFarGroup)
    int g, b, e;
    if (!far_work_item(t, g, b, e)) return;
    const int c0 = g * G, tid = threadIdx.x;
    const int w = min(G, ncols - c0);
    // the first batch of entries does not depend on the staged slice: its loads go out before the slice's
    constexpr int U = 4;
    double a[U];
    int pos[U];
    uint16_t lc[U];
    int k = b + tid;
    auto load_batch = [&](int k0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = min(k0 + u * kFarThreads, e - 1);  // clamped: branch-free, surplus lanes are masked at the store
            a[u] = __builtin_nontemporal_load(t.f_val + q);
            pos[u] = __builtin_nontemporal_load(t.f_pos + q);
            lc[u] = __builtin_nontemporal_load(t.f_lcol + q);
        }
    };
    load_batch(k);
    {   // slice -> LDS: all loads in flight before the first store
        constexpr int S = kFarGroup / kFarThreads;
        double sv[S];
#pragma unroll
        for (int i = 0; i < S; ++i)
            if (i * kFarThreads < G) sv[i] = vec[c0 + min(tid + i * kFarThreads, w - 1)];  // (uniform condition)
#pragma unroll
        for (int i = 0; i < S; ++i)
            if (i * kFarThreads < G) v[tid + i * kFarThreads] = sv[i];
    }
    __syncthreads();
    for (; k < e; k += U * kFarThreads) {
        double pr[U];
        int ps[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            pr[u] = LOGTERM ? a[u] - v[lc[u]] : a[u] * v[lc[u]];
            ps[u] = pos[u];
        }
        const int kn = k + U * kFarThreads;
        if (kn < e) load_batch(kn);  // next batch in flight while this one is stored
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (k + u * kFarThreads < e) t.P[ps[u]] = pr[u];
    }
}

// The same pre-pass for a consumer with run tables (all-remainder form, source groups of at most kPbRowsMax columns): positions
// from the group's run table in LDS instead of 4 bytes of f_pos per entry.
template <bool LOGTERM = false>
__global__ void __launch_bounds__(kFarThreads) k_far_products_runs(TiledDev t, const double *__restrict__ vec, int ncols) {
    __shared__ double v[kPbRowsMax];
    __shared__ int tab[2 * kPbRunTabCap];
    const int G = t.G;
    int g, b, e;
    if (!far_work_item(t, g, b, e)) return;
    const int c0 = g * G, tid = threadIdx.x;
    const int w = min(G, ncols - c0);
    const int r0 = t.f_rptr[g], nr = t.f_rptr[g + 1] - r0;
    int *tab_k = tab, *tab_p = tab + kPbRunTabCap;
    constexpr int U = 4;
    double a[U];
    uint16_t lc[U];
    int k = b + tid;
    auto load_batch = [&](int k0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = min(k0 + u * kFarThreads, e - 1);
            a[u] = __builtin_nontemporal_load(t.f_val + q);
            lc[u] = __builtin_nontemporal_load(t.f_lcol + q);
        }
    };
    load_batch(k);
    for (int i = tid; i < G; i += kFarThreads) v[i] = vec[c0 + min(i, w - 1)];
    for (int i = tid; i < nr; i += kFarThreads) {
        tab_k[i] = t.f_rk[r0 + i];
        tab_p[i] = t.f_rp[r0 + i];
    }
    __syncthreads();
    int ps[U];
#pragma unroll
    for (int u = 0; u < U; ++u) ps[u] = run_position(tab_k, tab_p, nr, min(k + u * kFarThreads, e - 1));
    for (; k < e; k += U * kFarThreads) {
        double pr[U];
        int pc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            pr[u] = LOGTERM ? a[u] - v[lc[u]] : a[u] * v[lc[u]];
            pc[u] = ps[u];
        }
        const int kn = k + U * kFarThreads;
        if (kn < e) load_batch(kn);
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (k + u * kFarThreads < e) t.P[pc[u]] = pr[u];
        if (kn < e) {
#pragma unroll
            for (int u = 0; u < U; ++u) ps[u] = run_position(tab_k, tab_p, nr, min(kn + u * kFarThreads, e - 1));
        }
    }
}

template <bool LOGTERM>
static void launch_far_products(const TiledDev &t, const double *vec, int ncols, hipStream_t s) {
    const dim3 grid(((t.f_work ? t.n_work : t.n_groups) + 7) / 8 * 8);
    if (t.f_rk && t.G <= kPbRowsMax) hipLaunchKernelGGL(k_far_products_runs<LOGTERM>, grid, dim3(kFarThreads), 0, s, t, vec, ncols);
    else hipLaunchKernelGGL(k_far_products<LOGTERM>, grid, dim3(kFarThreads), 0, s, t, vec, ncols);
}

// ------------------------------------------------------------------------------------------------
// epilogues
// ------------------------------------------------------------------------------------------------

// x-half (reference update_zx_{normal,check}_kernel, HPR_cuda_kernels.cu:203-247)
// STREAMED (set for matrices that run the tiled kernels, whose gathered vector lives in L2 as tiles): cache policy of the
// row's own streams, see HPRLP_EPI_NT above; small and mid-size LPs keep default-policy accesses (their vectors live in L2).
template <bool CHECK, bool STREAMED = false>
struct XEpi {
    static constexpr int NV = 1;
    static constexpr int NACC = CHECK ? 3 : 0;
    const double *gv[1];
    double *x, *x_hat;
    const double *l, *u, *c, *last_x;
    double *x_bar, *z_bar, *x_temp;
    Ctrl *ctrl;
    double *partials;
    int stride;
    // filled by begin()
    double sigma, f1, f2;
    FarPush push;  // hand-off of x_hat's remainder products to the y-half (kernels.h)
    const unsigned char *lu_code;  // which bounds to read (kernels.h: XHalfArgs::lu_code), or nullptr
    int x_mode;                    // kXRebuild / kXNoStore (kernels.h: XHalfArgs::x_mode); normal streamed variant only
    double f1p, f2p;               // the previous iteration's Halpern factors (filled by begin(), kXRebuild)
    static constexpr bool kPublishes = true;
    static constexpr bool kSkipsX = !CHECK && STREAMED;
    struct Row {
        double xi, ci, li, ui, lx;
    };
    __device__ __forceinline__ void begin() {
        const int k = ctrl->kx;
        sigma = ctrl->sigma;
        f1 = 1.0 / (static_cast<double>(k) + 2.0);
        f2 = 1.0 - f1;
        f1p = 1.0 / (static_cast<double>(k - 1) + 2.0);  // (as the previous x-half formed its f1, f2)
        f2p = 1.0 - f1p;
        if (blockIdx.x == 0 && threadIdx.x == 0) ctrl->ky = k;
    }
    __device__ __forceinline__ Row load_row(int r) const {
        constexpr bool NT = STREAMED && HPRLP_EPI_NT >= 1;
        auto ld = [](const double *p) { return NT ? __builtin_nontemporal_load(p) : *p; };
        double li, ui;
        if (lu_code) {
            // the same values the arrays hold, without reading the constant ones (codes are derived from the arrays: launch_bound_codes)
            const unsigned cd = NT ? __builtin_nontemporal_load(lu_code + r) : lu_code[r];
            li = (cd & kLoadL) ? ld(l + r) : ((cd & kZeroL) ? 0.0 : -__builtin_huge_val());
            ui = (cd & kLoadU) ? ld(u + r) : __builtin_huge_val();
        } else {
            li = ld(l + r);
            ui = ld(u + r);
        }
        const double lx = ld(last_x + r);
        if constexpr (kSkipsX) {
            if (x_mode & kXRebuild) return Row{f2p * ld(x_hat + r) + f1p * lx, ld(c + r), li, ui, lx};  // = the x the previous launch did not store
        }
        return Row{ld(x + r), ld(c + r), li, ui, lx};
    }
    // returns the value the half-step publishes for the other half's gather (x_hat)
    __device__ __forceinline__ double apply(int r, const Row &w, const double (&s)[1], double (&acc)[CHECK ? 3 : 1]) const {
        const double gc = s[0] - w.ci;
        const double zt = w.xi + sigma * gc;
        const double xb = fmin(w.ui, fmax(w.li, zt));
        const double xh = 2.0 * xb - w.xi;
        const double xn = f2 * xh + f1 * w.lx;
        {
            if constexpr (STREAMED && HPRLP_EPI_NT >= 2) __builtin_nontemporal_store(xh, x_hat + r);
            else x_hat[r] = xh;
            if (!kSkipsX || !(x_mode & kXNoStore)) {
                if constexpr (STREAMED && HPRLP_EPI_NT >= 1) __builtin_nontemporal_store(xn, x + r);
                else x[r] = xn;
            }
        }
        if constexpr (CHECK) {
            const double zb = (xb - zt) / sigma;
            const double dx = xb - xh;
            z_bar[r] = zb;
            x_bar[r] = xb;
            x_temp[r] = dx;
            acc[0] += w.ci * xb;
            acc[1] += xb * zb;
            acc[2] += dx * dx;
        }
        return xh;
    }
};

// y-half (reference update_y_{normal,check}_kernel, HPR_cuda_kernels.cu:249-295) + the Halpern
// counter advance (advance_halpern_factors_kernel, :192-200) folded in as the kx hand-off.
template <bool CHECK, bool STREAMED = false>
struct YEpi {
    static constexpr int NV = 1;
    static constexpr int NACC = CHECK ? 2 : 0;
    const double *gv[1];
    double *y;
    const double *AL, *AU, *last_y;
    double *y_bar, *y_obj, *y_temp;
    Ctrl *ctrl;
    double *partials;
    int stride;
    double fact1, fact2, hf1, hf2;
    FarPush push;  // hand-off of y's remainder products to the next x-half (kernels.h)
    const unsigned char *row_code;  // which of AL, AU to read (kernels.h: YHalfArgs::row_code), or nullptr
    static constexpr bool kPublishes = true;
    struct Row {
        double yi, lo, hi, ly;
    };
    __device__ __forceinline__ void begin() {
        const int k = ctrl->ky;
        fact1 = ctrl->lam_sigma;
        fact2 = ctrl->inv_lam_sigma;
        hf1 = 1.0 / (static_cast<double>(k) + 2.0);
        hf2 = 1.0 - hf1;
        if (blockIdx.x == 0 && threadIdx.x == 0) ctrl->kx = k + 1;
    }
    __device__ __forceinline__ Row load_row(int r) const {
        constexpr bool NT = STREAMED && HPRLP_EPI_NT >= 1;
        auto ld = [](const double *p) { return NT ? __builtin_nontemporal_load(p) : *p; };
        double lo, hi;
        if (row_code) {
            // the same values the arrays hold, without reading the constant ones (codes are derived from the arrays: launch_row_codes)
            const unsigned cd = NT ? __builtin_nontemporal_load(row_code + r) : row_code[r];
            hi = (cd & kLoadHi) ? ld(AU + r) : __builtin_huge_val();
            lo = (cd & kLoadLo) ? ld(AL + r) : ((cd & kRowEq) ? hi : -__builtin_huge_val());
        } else {
            lo = ld(AL + r);
            hi = ld(AU + r);
        }
        return Row{ld(y + r), lo, hi, ld(last_y + r)};
    }
    // returns the value the half-step publishes for the other half's gather (y)
    __device__ __forceinline__ double apply(int r, const Row &w, const double (&s)[1], double (&acc)[CHECK ? 2 : 1]) const {
        const double v = s[0] - fact1 * w.yi;
        const double d = fmax(w.lo - v, fmin(w.hi - v, 0.0));
        const double yb = fact2 * d;
        const double yh = 2.0 * yb - w.yi;
        const double yn = hf2 * yh + hf1 * w.ly;
        if constexpr (STREAMED && HPRLP_EPI_NT >= 2) __builtin_nontemporal_store(yn, y + r);
        else y[r] = yn;
        if constexpr (CHECK) {
            const double dy = yb - yh;
            const double yo = v + d;
            y_temp[r] = dy;
            y_bar[r] = yb;
            y_obj[r] = yo;
            acc[0] += yo * yb;
            acc[1] += dy * dy;
        }
        return yn;
    }
};

// dual residual (reference residual_compute_Rd_kernel, HPR_cuda_kernels.cu:183-189)
struct RdEpi {
    static constexpr int NV = 1;
    static constexpr int NACC = 1;
    const double *gv[1];
    const double *c, *z_bar, *col_norm;
    double *partials;
    int stride;
    struct Row {
        double ci, zi, cn;
    };
    __device__ __forceinline__ void begin() {}
    __device__ __forceinline__ Row load_row(int r) const { return Row{c[r], z_bar[r], col_norm[r]}; }
    __device__ __forceinline__ void apply(int, const Row &w, const double (&s)[1], double (&acc)[1]) const {
        const double rd = (w.ci - s[0] - w.zi) * w.cn;
        acc[0] += rd * rd;
    }
};

// primal residual (reference residual_compute_Rp_kernel, :160-172) and, when GAP, <A x_temp, y_temp>
// from the same pass over A (the reference runs a second SpMV, main_iterate.cu:245-253)
template <bool GAP>
struct RpEpi {
    static constexpr int NV = GAP ? 2 : 1;
    static constexpr int NACC = GAP ? 2 : 1;
    const double *gv[GAP ? 2 : 1];
    const double *AL, *AU, *row_norm, *y_temp;
    double *partials;
    int stride;
    struct Row {
        double lo, hi, rn, dy;
    };
    __device__ __forceinline__ void begin() {}
    __device__ __forceinline__ Row load_row(int r) const {
        return Row{AL[r], AU[r], row_norm[r], GAP ? y_temp[r] : 0.0};
    }
    __device__ __forceinline__ void apply(int, const Row &w, const double (&s)[GAP ? 2 : 1],
                                          double (&acc)[GAP ? 2 : 1]) const {
        const double v = s[0];
        const double rp = fmax(fmin(w.hi - v, 0.0), w.lo - v) * w.rn;
        acc[0] += rp * rp;
        if constexpr (GAP) acc[1] += s[1] * w.dy;
    }
};

struct GapEpi {
    static constexpr int NV = 1;
    static constexpr int NACC = 1;
    const double *gv[1];
    const double *y_temp;
    double *partials;
    int stride;
    struct Row {
        double dy;
    };
    __device__ __forceinline__ void begin() {}
    __device__ __forceinline__ Row load_row(int r) const { return Row{y_temp[r]}; }
    __device__ __forceinline__ void apply(int, const Row &w, const double (&s)[1], double (&acc)[1]) const {
        acc[0] += s[0] * w.dy;
    }
};

// plain product, optionally with out.out and out.q (power iteration, power_iteration.cu:73-93)
template <bool DOTS>
struct PlainEpi {
    static constexpr int NV = 1;
    static constexpr int NACC = DOTS ? 2 : 0;
    const double *gv[1];
    double *out;
    const double *q;
    double *partials;
    int stride;
    struct Row {
        double qi;
    };
    __device__ __forceinline__ void begin() {}
    __device__ __forceinline__ Row load_row(int r) const { return Row{DOTS ? q[r] : 0.0}; }
    __device__ __forceinline__ void apply(int r, const Row &w, const double (&s)[1], double (&acc)[DOTS ? 2 : 1]) const {
        out[r] = s[0];
        if constexpr (DOTS) {
            acc[0] += s[0] * s[0];
            acc[1] += s[0] * w.qi;
        }
    }
};

// plain product whose result is the OTHER matrix' gathered vector next (power iteration: A^T q feeds A): the epilogue hands the
// remainder products over like the half-steps do (kernels.h FarPush)
struct PlainPushEpi {
    static constexpr int NV = 1;
    static constexpr int NACC = 0;
    const double *gv[1];
    double *out;
    FarPush push;
    static constexpr bool kPublishes = true;
    struct Row {};
    __device__ __forceinline__ void begin() {}
    __device__ __forceinline__ Row load_row(int) const { return Row{}; }
    __device__ __forceinline__ double apply(int r, const Row &, const double (&s)[1], double (&)[1]) const {
        out[r] = s[0];
        return s[0];
    }
};

// ------------------------------------------------------------------------------------------------
// launch wrappers of the fused kernel
// ------------------------------------------------------------------------------------------------
// Rows longer than kSplitRow are cut into chunks that separate waves of k_spmv_fused reduce; this
// kernel adds the chunk sums of each such row in chunk order (deterministic) and runs the epilogue.
// Its reduction partials go behind those of the main kernel (index csr_grid() + blockIdx.x).
template <class Epi>
__global__ void __launch_bounds__(kThreads) k_long_finish(CsrDev A, Epi epi, int main_grid) {
    constexpr int NV = Epi::NV;
    constexpr int NACC = Epi::NACC;
    double acc[NACC > 0 ? NACC : 1];
#pragma unroll
    for (int i = 0; i < (NACC > 0 ? NACC : 1); ++i) acc[i] = 0.0;
    epi.begin();
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i < A.nlong) {
        const int4 d = A.longrows[i];  // {row, first chunk slot, one past the last slot, -}
        double s[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) s[v] = 0.0;
        for (int c = d.y; c < d.z; ++c) {
#pragma unroll
            for (int v = 0; v < NV; ++v) s[v] += A.long_partial[static_cast<size_t>(c) * 2 + v];
        }
        typename Epi::Row rw = epi.load_row(d.x);
        epi.apply(d.x, rw, s, acc);
    }
    if constexpr (NACC > 0) {
        // block_store_partials indexes by blockIdx.x: shift the base so that the slots follow the main kernel's
        block_store_partials<NACC>(acc, epi.partials + main_grid, epi.stride);
    }
}

template <class Epi, class = void>
struct Publishes : std::false_type {};
template <class Epi>
struct Publishes<Epi, std::void_t<decltype(Epi::kPublishes)>> : std::true_type {};

// far_ready: M's remainder buffer already holds the products of e.gv[0] (pushed by the producing half-step): no pre-pass.
// Returns true if the launch pushed (fused tiled kernel with a hand-off requested in e.push).
// Second kernel of a half-step whose matrix was split by columns (multi-GPU overlap): `base[r]` is the row sum over
// the local-column part, computed while the exchange was in flight; this launch adds the remote-column part and runs
// the epilogue.  Summation order per row: local entries (CSR order), then remote entries (CSR order).
template <class Epi>
struct WithBase : Epi {
    const double *base;
    struct Row {
        typename Epi::Row w;
        double b;
    };
    __device__ __forceinline__ Row load_row(int r) const { return Row{Epi::load_row(r), base[r]}; }
    __device__ __forceinline__ auto apply(int r, const Row &w, const double (&s)[1], double (&acc)[Epi::NACC > 0 ? Epi::NACC : 1]) const {
        const double t[1] = {w.b + s[0]};
        return Epi::apply(r, w.w, t, acc);
    }
};

// the long rows of a tiled matrix (tiled.h: TiledDev::side_*): their row sums, formed by the stream kernel over the CSR
// arrays, land in base[row]
struct SideEpi {
    static constexpr int NV = 1;
    static constexpr int NACC = 0;
    const double *gv[1];
    const int *rows;
    double *base;
    struct Row {};
    __device__ __forceinline__ void begin() {}
    __device__ __forceinline__ Row load_row(int) const { return Row{}; }
    __device__ __forceinline__ void apply(int r, const Row &, const double (&s)[1], double (&)[1]) const { base[rows[r]] = s[0]; }
};

template <class Epi>
static bool launch_tiled(const CsrDev &M, const Epi &e, hipStream_t s, bool far_ready);

template <class Epi>
static bool launch_fused(const CsrDev &M, const Epi &e, hipStream_t s, bool far_ready = false) {
    if (M.nblk <= 0) return false;
    if constexpr (Epi::NV == 1) {
        if (M.tiled.valid) {
            if (M.tiled.side_nblk > 0) {
                // long rows aside: their sums first (a few waves, stream kernel), then the tiled launch adds them
                CsrDev S = M;
                S.tiled.valid = false;
                S.blk = M.tiled.side_blk;
                S.nblk = M.tiled.side_nblk;
                S.longrows = M.tiled.side_long;
                S.nlong = M.tiled.side_nlong;
                S.long_partial = M.tiled.side_partial;
                SideEpi se{{e.gv[0]}, M.tiled.side_rows, M.tiled.base};
                hipLaunchKernelGGL(k_spmv_fused<SideEpi>, dim3(S.csr_grid()), dim3(kThreads), 0, s, S, se);
                if (S.nlong > 0) hipLaunchKernelGGL(k_long_finish<SideEpi>, dim3(S.finish_grid()), dim3(kThreads), 0, s, S, se, S.csr_grid());
                WithBase<Epi> wb{e, M.tiled.base};
                return launch_tiled(M, wb, s, far_ready);
            }
            return launch_tiled(M, e, s, far_ready);
        }
    }
    hipLaunchKernelGGL(k_spmv_fused<Epi>, dim3(M.csr_grid()), dim3(kThreads), 0, s, M, e);
    if (M.nlong > 0)
        hipLaunchKernelGGL(k_long_finish<Epi>, dim3(M.finish_grid()), dim3(kThreads), 0, s, M, e, M.csr_grid());
    return false;
}

template <class Epi>
static bool launch_tiled(const CsrDev &M, const Epi &e, hipStream_t s, bool far_ready) {
    if (M.tiled.n_groups > 0 && !far_ready)
        launch_far_products<false>(M.tiled, e.gv[0], M.cols, s);
    // copies with narrow tiles (tiled.h: kTileColsNarrow) run their own instantiation of the sweep, without the repeated-tile
    // shortcut (REP only saves work, it is not needed for correctness; measured on the banded 2e7 point: 0.098 / 0.084 ms with
    // it against 0.089 / 0.080 without -- re-staging 8 KiB costs less than the shortcut's control flow)
    const bool narrow = M.tiled.T == kTileColsNarrow;
    if (M.tiled.rem_cap == kPbRemCap) {  // all-remainder form (tiled.h): its own fused kernel
        if constexpr (Publishes<Epi>::value) {
            if (e.push.gptr) {
                hipLaunchKernelGGL((k_pb_fused<Epi, true>), dim3(M.tiled.grid), dim3(kTileThreads), 0, s, M, e);
                return true;
            }
        }
        hipLaunchKernelGGL((k_pb_fused<Epi, false>), dim3(M.tiled.grid), dim3(kTileThreads), 0, s, M, e);
        return false;
    }
    if (M.tiled.n_pieces > 0) {
        const dim3 grid((M.tiled.n_pieces + 7) / 8 * 8);
        if (narrow) hipLaunchKernelGGL((k_tiled_part<false, false, true>), grid, dim3(kTileThreads), 0, s, M, e.gv[0]);
        else if (M.tiled.stamps) hipLaunchKernelGGL((k_tiled_part<false, true>), grid, dim3(kTileThreads), 0, s, M, e.gv[0]);
        else if (M.tiled.repeats) hipLaunchKernelGGL(k_tiled_part<true>, grid, dim3(kTileThreads), 0, s, M, e.gv[0]);
        else hipLaunchKernelGGL(k_tiled_part<false>, grid, dim3(kTileThreads), 0, s, M, e.gv[0]);
        hipLaunchKernelGGL(k_tiled_finish<Epi>, dim3(M.tiled_finish_grid()), dim3(kThreads), 0, s, M, e);
        return false;
    }
    if constexpr (Publishes<Epi>::value) {
        if (e.push.gptr) {
            if (narrow) hipLaunchKernelGGL((k_tiled_fused<Epi, false, true, true>), dim3(M.tiled.grid), dim3(kTileThreads), 0, s, M, e);
            else if (M.tiled.repeats) hipLaunchKernelGGL((k_tiled_fused<Epi, true, true>), dim3(M.tiled.grid), dim3(kTileThreads), 0, s, M, e);
            else hipLaunchKernelGGL((k_tiled_fused<Epi, false, true>), dim3(M.tiled.grid), dim3(kTileThreads), 0, s, M, e);
            return true;
        }
    }
    if (narrow) hipLaunchKernelGGL((k_tiled_fused<Epi, false, false, true>), dim3(M.tiled.grid), dim3(kTileThreads), 0, s, M, e);
    else if (M.tiled.repeats) hipLaunchKernelGGL((k_tiled_fused<Epi, true>), dim3(M.tiled.grid), dim3(kTileThreads), 0, s, M, e);
    else hipLaunchKernelGGL((k_tiled_fused<Epi, false>), dim3(M.tiled.grid), dim3(kTileThreads), 0, s, M, e);
    return false;
}

FarPush far_push_of(const CsrDev &consumer) {
    FarPush f;
    if (consumer.tiled.valid && consumer.tiled.n_groups > 0) {
        f.gptr = consumer.tiled.f_gptr;
        f.val = consumer.tiled.f_val;
        f.pos = consumer.tiled.f_pos;
        f.lcol = consumer.tiled.f_lcol;
        f.P = consumer.tiled.P;
        if (consumer.tiled.f_rk) {
            f.rptr = consumer.tiled.f_rptr;
            f.rk = consumer.tiled.f_rk;
            f.rp = consumer.tiled.f_rp;
        }
    }
    return f;
}

bool launch_x_half(const CsrDev &AT, const XHalfArgs &a, bool check, hipStream_t s) {
    if (check) {
        XEpi<true> e{{a.y_full}, a.x, a.x_hat, a.l, a.u, a.c, a.last_x, a.x_bar, a.z_bar, a.x_temp, a.ctrl, a.partials, a.stride, 0, 0, 0, a.push, a.lu_code};
        return launch_fused(AT, e, s, a.far_ready);
    }
    if (AT.tiled.valid) {  // the tiled kernels: the row's own streams bypass the L2s (HPRLP_EPI_NT)
        XEpi<false, true> e{{a.y_full}, a.x, a.x_hat, a.l, a.u, a.c, a.last_x, nullptr, nullptr, nullptr, a.ctrl, nullptr, 0, 0, 0, 0, a.push, a.lu_code, a.x_mode};
        return launch_fused(AT, e, s, a.far_ready);
    }
    XEpi<false> e{{a.y_full}, a.x, a.x_hat, a.l, a.u, a.c, a.last_x, nullptr, nullptr, nullptr, a.ctrl, nullptr, 0, 0, 0, 0, a.push, a.lu_code};
    return launch_fused(AT, e, s, a.far_ready);
}

bool launch_y_half(const CsrDev &A, const YHalfArgs &a, bool check, hipStream_t s) {
    if (check) {
        YEpi<true> e{{a.xhat_full}, a.y, a.AL, a.AU, a.last_y, a.y_bar, a.y_obj, a.y_temp, a.ctrl, a.partials, a.stride, 0, 0, 0, 0, a.push, a.row_code};
        return launch_fused(A, e, s, a.far_ready);
    }
    if (A.tiled.valid) {
        YEpi<false, true> e{{a.xhat_full}, a.y, a.AL, a.AU, a.last_y, nullptr, nullptr, nullptr, a.ctrl, nullptr, 0, 0, 0, 0, 0, a.push, a.row_code};
        return launch_fused(A, e, s, a.far_ready);
    }
    YEpi<false> e{{a.xhat_full}, a.y, a.AL, a.AU, a.last_y, nullptr, nullptr, nullptr, a.ctrl, nullptr, 0, 0, 0, 0, 0, a.push, a.row_code};
    return launch_fused(A, e, s, a.far_ready);
}


void launch_x_half_base(const CsrDev &AT_remote, const XHalfArgs &a, const double *base, hipStream_t s) {
    WithBase<XEpi<false>> e{{{a.y_full}, a.x, a.x_hat, a.l, a.u, a.c, a.last_x, nullptr, nullptr, nullptr, a.ctrl, nullptr, 0, 0, 0, 0, FarPush{}, a.lu_code}, base};
    launch_fused(AT_remote, e, s);
}

void launch_y_half_base(const CsrDev &A_remote, const YHalfArgs &a, const double *base, hipStream_t s) {
    WithBase<YEpi<false>> e{{{a.xhat_full}, a.y, a.AL, a.AU, a.last_y, nullptr, nullptr, nullptr, a.ctrl, nullptr, 0, 0, 0, 0, 0, FarPush{}}, base};
    launch_fused(A_remote, e, s);
}

void launch_resid_d(const CsrDev &AT, const double *ybar_full, const double *c, const double *z_bar,
                    const double *col_norm, double *partials, hipStream_t s) {
    RdEpi e{{ybar_full}, c, z_bar, col_norm, partials, AT.grid()};
    launch_fused(AT, e, s);
}

void launch_resid_p(const CsrDev &A, const double *xbar_full, const double *xtemp_full, const double *AL,
                    const double *AU, const double *row_norm, const double *y_temp, bool with_gap,
                    double *partials, int stride, hipStream_t s) {
    if (with_gap && A.tiled.valid) {  // the tiled kernel stages one vector: two passes
        RpEpi<false> e{{xbar_full}, AL, AU, row_norm, nullptr, partials, stride};
        launch_fused(A, e, s);
        GapEpi g{{xtemp_full}, y_temp, partials + stride, stride};
        launch_fused(A, g, s);
    } else if (with_gap) {
        RpEpi<true> e{{xbar_full, xtemp_full}, AL, AU, row_norm, y_temp, partials, stride};
        launch_fused(A, e, s);
    } else {
        RpEpi<false> e{{xbar_full}, AL, AU, row_norm, nullptr, partials, stride};
        launch_fused(A, e, s);
    }
}

void launch_gap(const CsrDev &A, const double *xtemp_full, const double *y_temp, double *partials, hipStream_t s) {
    GapEpi e{{xtemp_full}, y_temp, partials, A.grid()};
    launch_fused(A, e, s);
}

bool launch_spmv_push(const CsrDev &M, const double *v_full, double *out, const FarPush &push, hipStream_t s) {
    PlainPushEpi e{{v_full}, out, push};
    return launch_fused(M, e, s);
}

void launch_spmv_plain(const CsrDev &M, const double *v_full, double *out, const double *q, bool with_dots,
                       double *partials, int stride, hipStream_t s, bool far_ready) {
    if (with_dots) {
        PlainEpi<true> e{{v_full}, out, q, partials, stride};
        launch_fused(M, e, s, far_ready);
    } else if (far_ready) {
        PlainEpi<false> e{{v_full}, out, nullptr, nullptr, 0};
        launch_fused(M, e, s, true);
    } else {
        PlainEpi<false> e{{v_full}, out, nullptr, nullptr, 0};
        launch_fused(M, e, s);
    }
}

// ------------------------------------------------------------------------------------------------
// scalar finalisation: out[slot] = sum of a partial array, one block per item, fixed order
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads) k_finalize(FinalizeArgs f, double *scalars) {
    const FinalizeItem it = f.item[blockIdx.x];
    double v = 0.0;
    for (int i = threadIdx.x; i < it.count; i += kThreads) v += it.partials[i];
    __shared__ double red[kWavesPerBlock];
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) scalars[it.slot] = ((red[0] + red[1]) + red[2]) + red[3];
}

void launch_finalize(const FinalizeArgs &f, double *scalars, hipStream_t s) {
    if (f.n <= 0) return;
    hipLaunchKernelGGL(k_finalize, dim3(f.n), dim3(kThreads), 0, s, f, scalars);
}

// ------------------------------------------------------------------------------------------------
// plain vector kernels (grid-stride, fixed grids so partial counts are static)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads) k_movement(int n, int m, const double *x_bar, const double *last_x,
                                                      double *x_temp, const double *y_bar, const double *last_y,
                                                      double *y_temp, double *partials, int stride) {
    double acc[2] = {0.0, 0.0};
    const int tid = blockIdx.x * kThreads + threadIdx.x, nth = gridDim.x * kThreads;
    for (int i = tid; i < n; i += nth) {
        const double d = 1.0 * x_bar[i] + (-1.0) * last_x[i];
        x_temp[i] = d;
        acc[0] += d * d;
    }
    for (int i = tid; i < m; i += nth) {
        const double d = 1.0 * y_bar[i] + (-1.0) * last_y[i];
        y_temp[i] = d;
        acc[1] += d * d;
    }
    block_store_partials<2>(acc, partials, stride);
}

void launch_movement(int n, int m, const double *x_bar, const double *last_x, double *x_temp, const double *y_bar,
                     const double *last_y, double *y_temp, double *partials, int stride, int nblocks, hipStream_t s) {
    hipLaunchKernelGGL(k_movement, dim3(nblocks), dim3(kThreads), 0, s, n, m, x_bar, last_x, x_temp, y_bar, last_y,
                       y_temp, partials, stride);
}

__global__ void __launch_bounds__(kThreads) k_restart_copy(int n, int m, const double *x_bar, double *x, double *last_x,
                                                          const double *y_bar, double *y, double *last_y, Ctrl *ctrl) {
    const int tid = blockIdx.x * kThreads + threadIdx.x, nth = gridDim.x * kThreads;
    for (int i = tid; i < n; i += nth) {
        const double v = x_bar[i];
        x[i] = v;
        last_x[i] = v;
    }
    for (int i = tid; i < m; i += nth) {
        const double v = y_bar[i];
        y[i] = v;
        last_y[i] = v;
    }
    if (tid == 0) {
        ctrl->kx = 0;
        ctrl->ky = 0;
    }
}

static int vec_grid(long n) {
    long g = (n + kThreads - 1) / kThreads;
    if (g < 1) g = 1;
    if (g > 2048) g = 2048;
    return static_cast<int>(g);
}

void launch_restart_copy(int n, int m, const double *x_bar, double *x, double *last_x, const double *y_bar,
                         double *y, double *last_y, Ctrl *ctrl, hipStream_t s) {
    hipLaunchKernelGGL(k_restart_copy, dim3(vec_grid(n > m ? n : m)), dim3(kThreads), 0, s, n, m, x_bar, x, last_x,
                       y_bar, y, last_y, ctrl);
}

__global__ void __launch_bounds__(kThreads) k_lu(int n, const double *x_bar, const double *l, const double *u,
                                                const double *col_norm, double *x_temp, double *partials) {
    double acc[1] = {0.0};
    const int tid = blockIdx.x * kThreads + threadIdx.x, nth = gridDim.x * kThreads;
    for (int i = tid; i < n; i += nth) {
        const double xb = x_bar[i], li = l[i], ui = u[i];
        const double t = (xb < li) ? (li - xb) : ((xb > ui) ? (xb - ui) : 0.0);
        const double r = t / col_norm[i];
        x_temp[i] = r;
        acc[0] += r * r;
    }
    block_store_partials<1>(acc, partials, gridDim.x);
}

void launch_lu(int n, const double *x_bar, const double *l, const double *u, const double *col_norm, double *x_temp,
               double *partials, int nblocks, hipStream_t s) {
    hipLaunchKernelGGL(k_lu, dim3(nblocks), dim3(kThreads), 0, s, n, x_bar, l, u, col_norm, x_temp, partials);
}

__global__ void k_set_ctrl(Ctrl *ctrl, double sigma, double lambda_max, int reset_k) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const double f = lambda_max * sigma;
        ctrl->sigma = sigma;
        ctrl->lam_sigma = f;
        ctrl->inv_lam_sigma = 1.0 / f;
        ctrl->inv_sigma = 1.0 / sigma;
        if (reset_k) {
            ctrl->kx = 0;
            ctrl->ky = 0;
        }
    }
}

void launch_set_ctrl(Ctrl *ctrl, double sigma, double lambda_max, int reset_k, hipStream_t s) {
    hipLaunchKernelGGL(k_set_ctrl, dim3(1), dim3(64), 0, s, ctrl, sigma, lambda_max, reset_k);
}

// ------------------------------------------------------------------------------------------------
// scaling kernels (reference src/scaling.cu:5-38, HPR_cuda_kernels.cu:34-43,91-157)
// ------------------------------------------------------------------------------------------------
// Curtis-Reid update (reference scaling.cu:5-38): result[r] = mean over the row's entries of (-log|a| - other[col]).
// Runs as an epilogue of the stream kernel (coalesced matrix reads; per row the terms are added in CSR order like
// the thread-per-row loop it replaces, which took 6.4 ms per pass on the 2e8-nnz matrix against 1.4 ms).  Through the
// tiled kernel when the caller holds the copy's log values (launch_tiled_refresh_log: -log|a| formed once instead of in each
// of the 20 passes, NaN = padding): the same terms, added in the tiled kernel's order.
struct CrEpi {
    static constexpr int NV = 1;
    static constexpr int NACC = 0;
    static constexpr bool kLogTerm = true;
    const double *gv[1];
    const int *rowptr;
    double *result;
    double *partials;
    int stride;
    struct Row {
        int cnt;
    };
    static __device__ __forceinline__ double term(double a, double g) { return -log(fmax(fabs(a), 1e-300)) - g; }
    __device__ __forceinline__ void begin() {}
    __device__ __forceinline__ Row load_row(int r) const { return Row{rowptr[r + 1] - rowptr[r]}; }
    __device__ __forceinline__ void apply(int r, const Row &w, const double (&s)[1], double (&)[1]) const {
        result[r] = w.cnt > 0 ? s[0] / static_cast<double>(w.cnt) : 0.0;
    }
};

bool cr_runs_tiled(const CsrDev &M) { return M.tiled.valid && M.tiled.n_pieces == 0 && M.tiled.side_nblk == 0; }

void launch_cr_log_update(const CsrDev &M, const double *other_full, double *result, hipStream_t s, const double *tval_log,
                          const double *fval_log) {
    if (M.rows <= 0 || M.nblk <= 0) return;
    CrEpi e{{other_full}, M.rowptr, result, nullptr, 0};
    if (tval_log && cr_runs_tiled(M)) {
        CsrDev L = M;
        L.tiled.tval = tval_log;
        L.tiled.f_val = fval_log;
        if (L.tiled.n_groups > 0)
            launch_far_products<true>(L.tiled, other_full, L.cols, s);
        if (L.tiled.rem_cap == kPbRemCap) hipLaunchKernelGGL((k_pb_fused<CrEpi, false>), dim3(L.tiled.grid), dim3(kTileThreads), 0, s, L, e);  // all-remainder form
        else if (L.tiled.T == kTileColsNarrow) hipLaunchKernelGGL((k_tiled_fused<CrEpi, false, false, true>), dim3(L.tiled.grid), dim3(kTileThreads), 0, s, L, e);
        else if (L.tiled.repeats) hipLaunchKernelGGL((k_tiled_fused<CrEpi, true>), dim3(L.tiled.grid), dim3(kTileThreads), 0, s, L, e);
        else hipLaunchKernelGGL((k_tiled_fused<CrEpi, false>), dim3(L.tiled.grid), dim3(kTileThreads), 0, s, L, e);
        return;
    }
    hipLaunchKernelGGL(k_spmv_fused<CrEpi>, dim3(M.csr_grid()), dim3(kThreads), 0, s, M, e);
    if (M.nlong > 0)
        hipLaunchKernelGGL(k_long_finish<CrEpi>, dim3(M.finish_grid()), dim3(kThreads), 0, s, M, e, M.csr_grid());
}

__global__ void __launch_bounds__(kThreads) k_exp_clamp(double *v, int n) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i < n) v[i] = fmin(fmax(exp(v[i]), 1e-30), 1e30);
}

void launch_exp_clamp(double *v, int n, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_exp_clamp, dim3((n + kThreads - 1) / kThreads), dim3(kThreads), 0, s, v, n);
}

__global__ void __launch_bounds__(kThreads) k_row_norm(int rows, const int *rowptr, const double *val, double *result,
                                                      int norm) {
    const int r = blockIdx.x * kThreads + threadIdx.x;
    if (r >= rows) return;
    double acc = 0.0;
    if (norm == 99) {
        for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) {
            const double a = fabs(val[k]);
            if (acc < a) acc = a;
        }
    } else {
        for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) acc += fabs(val[k]);
    }
    acc = sqrt(acc);
    if (acc < 1e-15) acc = 1.0;
    result[r] = acc;
}

// The max-norm passes (Ruiz: 10 per matrix) over the row blocks of the stream kernel: one wave per block, lanes
// over the entries (coalesced), |a| through LDS, one lane per row takes the maximum of its segment.  A maximum does
// not depend on the order, so the result equals the thread-per-row loop's bit for bit (2.2 -> 0.4 ms per pass on
// 2e8 nonzeros).  The sum norm: k_row_sum_blocks below.
__device__ __forceinline__ double norm_result(double acc) {
    acc = sqrt(acc);
    return acc < 1e-15 ? 1.0 : acc;
}

__global__ void __launch_bounds__(kThreads) k_row_max_blocks(CsrDev M, double *result) {
    __shared__ double lds[kWavesPerBlock][kStreamW];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x * kWavesPerBlock + wave;
    if (b >= M.nblk) return;
    const int4 d = M.blk[b];
    const int r0 = d.x, nr = d.y, k0 = d.z, nz = d.w;
    if (nr == 0) return;  // chunk of a split row: k_row_max_long
    const double *__restrict__ val = M.val + k0;
    if (nr == 1 && nz > kLongRow) {
        double acc = 0.0;
        for (int j = lane; j < nz; j += kWave) {
            const double a = fabs(val[j]);
            if (acc < a) acc = a;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double o = __shfl_xor(acc, off);
            if (acc < o) acc = o;
        }
        if (lane == 0) result[r0] = norm_result(acc);
        return;
    }
    int rs = 0, re = 0;
    if (lane < nr) {
        rs = M.rowptr[r0 + lane] - k0;
        re = M.rowptr[r0 + lane + 1] - k0;
    }
    for (int j = lane; j < nz; j += kWave) lds[wave][j] = fabs(val[j]);
    wave_lds_sync();
    if (lane < nr) {
        double acc = 0.0;
        for (int j = rs; j < re; ++j) {
            const double a = lds[wave][j];
            if (acc < a) acc = a;
        }
        result[r0 + lane] = norm_result(acc);
    }
}

__global__ void __launch_bounds__(kThreads) k_row_max_long(CsrDev M, double *result) {
    __shared__ double red[kWavesPerBlock];
    const int r = M.longrows[blockIdx.x].x;
    double acc = 0.0;
    for (int k = M.rowptr[r] + threadIdx.x; k < M.rowptr[r + 1]; k += kThreads) {
        const double a = fabs(M.val[k]);
        if (acc < a) acc = a;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(acc, off);
        if (acc < o) acc = o;
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kWavesPerBlock; ++w)
            if (acc < red[w]) acc = red[w];
        result[r] = norm_result(acc);
    }
}

// The sum norm (Pock-Chambolle, one pass per matrix) over the same row blocks: |a| through LDS (coalesced reads), one lane per
// row adds its segment in CSR order -- the order of the thread-per-row loop and of the oracle, so the sums are the same bit for
// bit (2.3 -> 0.5 ms per pass on 2e8 nonzeros).  Rows in vector mode or split into chunks: one lane, in order, from memory.
__global__ void __launch_bounds__(kThreads) k_row_sum_blocks(CsrDev M, double *result) {
    __shared__ double lds[kWavesPerBlock][kStreamW];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x * kWavesPerBlock + wave;
    if (b >= M.nblk) return;
    const int4 d = M.blk[b];
    const int r0 = d.x, nr = d.y, k0 = d.z, nz = d.w;
    if (nr == 0) return;  // chunk of a split row: k_row_sum_long
    const double *__restrict__ val = M.val + k0;
    if (nr == 1 && nz > kLongRow) {
        if (lane == 0) {
            double acc = 0.0;
            for (int j = 0; j < nz; ++j) acc += fabs(val[j]);
            result[r0] = norm_result(acc);
        }
        return;
    }
    int rs = 0, re = 0;
    if (lane < nr) {
        rs = M.rowptr[r0 + lane] - k0;
        re = M.rowptr[r0 + lane + 1] - k0;
    }
    for (int j = lane; j < nz; j += kWave) lds[wave][j] = fabs(val[j]);
    wave_lds_sync();
    if (lane < nr) {
        double acc = 0.0;
        for (int j = rs; j < re; ++j) acc += lds[wave][j];
        result[r0 + lane] = norm_result(acc);
    }
}

__global__ void __launch_bounds__(kWave) k_row_sum_long(CsrDev M, double *result) {
    const int i = blockIdx.x * kWave + threadIdx.x;
    if (i >= M.nlong) return;
    const int r = M.longrows[i].x;
    double acc = 0.0;
    for (int k = M.rowptr[r]; k < M.rowptr[r + 1]; ++k) acc += fabs(M.val[k]);
    result[r] = norm_result(acc);
}

// How well the stream kernel's gathers coalesce: distinct 64-byte lines of the gathered vector per matrix entry, over sampled
// windows of 64 consecutive rows (what one wave of k_spmv_fused gathers for at most).  1: every entry pulls its own line through
// the L2 (a banded-random row pattern); 1/8: eight entries share a line (stencil rows, incidence matrices).  One workgroup of 64
// lanes per window; lines within 2^16 of the window's lowest are marked in an LDS bitmap, lines beyond that count as distinct.
constexpr int kDensityWindowRows = 64;
__global__ void __launch_bounds__(kWave) k_line_density(const int *__restrict__ rowptr, const int *__restrict__ col, int rows, int nwin,
                                                        unsigned long long *out) {
    __shared__ unsigned int bits[2048];  // 65536 lines
    __shared__ int lo_s;
    const int lane = threadIdx.x;
    const int r0 = static_cast<int>(static_cast<long>(blockIdx.x) * (rows - kDensityWindowRows) / max(nwin - 1, 1));
    const int k0 = rowptr[r0], k1 = rowptr[min(r0 + kDensityWindowRows, rows)];
    for (int i = lane; i < 2048; i += kWave) bits[i] = 0u;
    int lo = 0x7fffffff;
    for (int k = k0 + lane; k < k1; k += kWave) lo = min(lo, col[k] >> 3);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lo = min(lo, __shfl_xor(lo, off));
    if (lane == 0) lo_s = lo;
    __syncthreads();
    lo = lo_s;
    unsigned long long far = 0;
    for (int k = k0 + lane; k < k1; k += kWave) {
        const int d = (col[k] >> 3) - lo;
        if (d < 65536) atomicOr(&bits[d >> 5], 1u << (d & 31));
        else ++far;
    }
    __syncthreads();
    unsigned long long cnt = far;
    for (int i = lane; i < 2048; i += kWave) cnt += __popc(bits[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
    if (lane == 0) {
        out[2 * blockIdx.x] = cnt;
        out[2 * blockIdx.x + 1] = static_cast<unsigned long long>(k1 - k0);
    }
}

double launch_line_density(const int *rowptr, const int *col, int rows, hipStream_t s) {
    if (rows < 4 * kDensityWindowRows) return 1.0;
    const int nwin = std::min(1024, rows / kDensityWindowRows);
    DBuf<unsigned long long> out;
    out.alloc(static_cast<size_t>(2) * nwin);
    hipLaunchKernelGGL(k_line_density, dim3(nwin), dim3(kWave), 0, s, rowptr, col, rows, nwin, out.p);
    std::vector<unsigned long long> h(static_cast<size_t>(2) * nwin);
    HIP_CHECK(hipMemcpyAsync(h.data(), out.p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    // entry-weighted mean over the windows, a window's weight capped at four times the median window's entries: the windows
    // are a 1-in-50 sample of the rows, and ONE window of heavy rows (forty budget rows of 900 random entries behind a
    // 3M-row bidiagonal matrix: 0.6 % of the entries, a fifth of the sample's) used to lift 0.13 lines per entry over the
    // stream kernel's threshold (held-out corpus of tools/form_regret.py, round 5)
    std::vector<unsigned long long> ent(static_cast<size_t>(nwin));
    for (int w = 0; w < nwin; ++w) ent[w] = h[2 * w + 1];
    std::nth_element(ent.begin(), ent.begin() + nwin / 2, ent.end());
    const double cap = 4.0 * static_cast<double>(std::max<unsigned long long>(ent[nwin / 2], 1));
    double num = 0.0, den = 0.0;
    for (int w = 0; w < nwin; ++w) {
        const double e = static_cast<double>(h[2 * w + 1]);
        if (e <= 0.0) continue;
        const double wt = std::min(e, cap);
        num += wt * static_cast<double>(h[2 * w]) / e;
        den += wt;
    }
    return den > 0.0 ? num / den : 1.0;
}

__global__ void __launch_bounds__(kThreads) k_longest_row(const int *__restrict__ rowptr, int rows, int limit, int *out, unsigned long long *in_long) {
    int best = 0;
    unsigned long long lg = 0;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < rows; i += gridDim.x * kThreads) {
        const int len = rowptr[i + 1] - rowptr[i];
        best = max(best, len);
        if (len > limit) lg += static_cast<unsigned long long>(len);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        best = max(best, __shfl_xor(best, off));
        lg += __shfl_xor(lg, off);
    }
    if ((threadIdx.x & 63) == 0) {
        if (best > 0) atomicMax(out, best);
        if (lg > 0) atomicAdd(in_long, lg);
    }
}

// entries of the longest row; entries_in_long (optional): the entries that lie in rows of more than `limit` entries
int launch_longest_row(const int *rowptr, int rows, hipStream_t s, int limit, long *entries_in_long) {
    if (entries_in_long) *entries_in_long = 0;
    if (rows <= 0) return 0;
    DBuf<unsigned long long> out;
    out.alloc(2);
    HIP_CHECK(hipMemsetAsync(out.p, 0, 2 * sizeof(unsigned long long), s));
    const int grid = std::min((rows + kThreads - 1) / kThreads, 2048);
    hipLaunchKernelGGL(k_longest_row, dim3(grid), dim3(kThreads), 0, s, rowptr, rows, limit, reinterpret_cast<int *>(out.p), out.p + 1);
    unsigned long long h[2] = {0, 0};
    HIP_CHECK(hipMemcpyAsync(h, out.p, sizeof(h), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    if (entries_in_long) *entries_in_long = static_cast<long>(h[1]);
    return static_cast<int>(h[0] & 0xffffffffull);
}

// the most entries any block of `height` consecutive rows holds (the heaviest super-block of a tiled copy of that height)
__global__ void __launch_bounds__(kThreads) k_heaviest_block(const int *__restrict__ rowptr, int rows, int height, int *out) {
    const int nb = (rows + height - 1) / height;
    int best = 0;
    for (int b = blockIdx.x * kThreads + threadIdx.x; b < nb; b += gridDim.x * kThreads)
        best = max(best, rowptr[min(rows, (b + 1) * height)] - rowptr[b * height]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) best = max(best, __shfl_xor(best, off));
    if ((threadIdx.x & 63) == 0 && best > 0) atomicMax(out, best);
}

int launch_heaviest_block(const int *rowptr, int rows, int height, hipStream_t s) {
    if (rows <= 0 || height <= 0) return 0;
    DBuf<int> out;
    out.alloc(1);
    HIP_CHECK(hipMemsetAsync(out.p, 0, sizeof(int), s));
    const int nb = (rows + height - 1) / height;
    hipLaunchKernelGGL(k_heaviest_block, dim3(std::min((nb + kThreads - 1) / kThreads, 1024)), dim3(kThreads), 0, s, rowptr, rows, height, out.p);
    int h = 0;
    HIP_CHECK(hipMemcpyAsync(&h, out.p, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    return h;
}

void launch_row_norm(const CsrDev &M, double *result, int norm, hipStream_t s) {
    if (M.rows <= 0) return;
    if (norm == 99 && M.nblk > 0) {
        hipLaunchKernelGGL(k_row_max_blocks, dim3(M.csr_grid()), dim3(kThreads), 0, s, M, result);
        if (M.nlong > 0) hipLaunchKernelGGL(k_row_max_long, dim3(M.nlong), dim3(kThreads), 0, s, M, result);
        return;
    }
    if (norm == 1 && M.nblk > 0) {
        hipLaunchKernelGGL(k_row_sum_blocks, dim3(M.csr_grid()), dim3(kThreads), 0, s, M, result);
        if (M.nlong > 0) hipLaunchKernelGGL(k_row_sum_long, dim3((M.nlong + kWave - 1) / kWave), dim3(kWave), 0, s, M, result);
        return;
    }
    hipLaunchKernelGGL(k_row_norm, dim3((M.rows + kThreads - 1) / kThreads), dim3(kThreads), 0, s, M.rows, M.rowptr,
                       M.val, result, norm);
}

// val = op(op(val, first), second) with first/second = the row's scale and the gathered column scale (reference
// scale_rows/scale_columns kernels, HPR_cuda_kernels.cu:91-157).  One wave per row block, lane = matrix entry
// (coalesced read-modify-write of val, coalesced col).  (The thread-per-row loop this replaces took 8.7 ms per pass on 2e8 nonzeros.)
template <bool ROW_FIRST, bool DIVIDE>
__device__ __forceinline__ double scale_entry(double v, double rv, double cv) {
    if (ROW_FIRST) {
        v = DIVIDE ? v / rv : v * rv;
        v = DIVIDE ? v / cv : v * cv;
    } else {
        v = DIVIDE ? v / cv : v * cv;
        v = DIVIDE ? v / rv : v * rv;
    }
    return v;
}

// Round 4: the entry's row scale is spread over the block's entries by the row's own lane first (rsv[j] = rowvec[row of j]: plain
// LDS stores, no dependent chain -- the binary search in the row offsets this replaces was six dependent LDS reads per entry and
// held the pass at 2.0 ms on 2e8 nonzeros), and with NEXT the pass also leaves norm_result(max |new value|) of every row in
// next_norm: the following Ruiz pass's row norm, which otherwise re-read the matrix.  The maximum does not depend on the order.
template <bool ROW_FIRST, bool DIVIDE, bool NEXT>
__global__ void __launch_bounds__(kThreads) k_scale_matrix(CsrDev M, const double *rowvec, const double *colvec, double *next_norm) {
    __shared__ double rsv[kWavesPerBlock][kStreamW];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x * kWavesPerBlock + wave;
    if (b >= M.nblk) return;
    const int4 d = M.blk[b];
    const int r0 = d.x, nr = d.y, k0 = d.z, nz = d.w;
    if (nr == 0) return;  // chunk of a split row: k_scale_long_rows
    const int *__restrict__ col = M.col + k0;
    double *__restrict__ val = M.val + k0;
    if (nr == 1 && nz > kLongRow) {
        const double rv = rowvec[r0];
        double acc = 0.0;
        for (int j = lane; j < nz; j += kWave) {
            const double nv = scale_entry<ROW_FIRST, DIVIDE>(val[j], rv, colvec[col[j]]);
            val[j] = nv;
            if (NEXT && acc < fabs(nv)) acc = fabs(nv);
        }
        if (NEXT) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const double o = __shfl_xor(acc, off);
                if (acc < o) acc = o;
            }
            if (lane == 0) next_norm[r0] = norm_result(acc);
        }
        return;
    }
    // all of the block's entries (at most kStreamW / kWave = 8 per lane) are requested before anything waits: with one entry per
    // trip of a loop the pass sat at 2 TB/s behind two dependent memory latencies per trip (index, then gathered scale)
    constexpr int PER = kStreamW / kWave;
    double v[PER], cv[PER];
    int c[PER];
#pragma unroll
    for (int t = 0; t < PER; ++t) {
        const int j = lane + kWave * t;
        if (j < nz) {
            c[t] = __builtin_nontemporal_load(col + j);
            v[t] = __builtin_nontemporal_load(val + j);
        }
    }
    int rs = 0, re = 0;
    if (lane < nr) {
        rs = M.rowptr[r0 + lane] - k0;
        re = M.rowptr[r0 + lane + 1] - k0;
    }
#pragma unroll
    for (int t = 0; t < PER; ++t)
        if (lane + kWave * t < nz) cv[t] = colvec[c[t]];
    if (lane < nr) {
        const double rv = rowvec[r0 + lane];
        for (int j = rs; j < re; ++j) rsv[wave][j] = rv;
    }
    wave_lds_sync();
#pragma unroll
    for (int t = 0; t < PER; ++t) {
        const int j = lane + kWave * t;
        if (j < nz) {
            const double nv = scale_entry<ROW_FIRST, DIVIDE>(v[t], rsv[wave][j], cv[t]);
            __builtin_nontemporal_store(nv, val + j);
            if (NEXT) rsv[wave][j] = fabs(nv);  // the slot's row scale has been read by this lane
        }
    }
    if (NEXT) {
        wave_lds_sync();
        if (lane < nr) {
            double acc = 0.0;
            for (int j = rs; j < re; ++j) {
                const double a = rsv[wave][j];
                if (acc < a) acc = a;
            }
            next_norm[r0 + lane] = norm_result(acc);
        }
    }
}

// rows longer than kSplitRow (their blocks carry chunk slots, not row numbers): one workgroup per row
template <bool ROW_FIRST, bool DIVIDE, bool NEXT>
__global__ void __launch_bounds__(kThreads) k_scale_long_rows(CsrDev M, const double *rowvec, const double *colvec, double *next_norm) {
    __shared__ double red[kWavesPerBlock];
    const int r = M.longrows[blockIdx.x].x;
    const double rv = rowvec[r];
    double acc = 0.0;
    for (int k = M.rowptr[r] + threadIdx.x; k < M.rowptr[r + 1]; k += kThreads) {
        const double nv = scale_entry<ROW_FIRST, DIVIDE>(M.val[k], rv, colvec[M.col[k]]);
        M.val[k] = nv;
        if (NEXT && acc < fabs(nv)) acc = fabs(nv);
    }
    if (NEXT) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double o = __shfl_xor(acc, off);
            if (acc < o) acc = o;
        }
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < kWavesPerBlock; ++w)
                if (acc < red[w]) acc = red[w];
            next_norm[r] = norm_result(acc);
        }
    }
}

template <bool ROW_FIRST, bool DIVIDE, bool NEXT>
static void launch_scale_matrix_t(const CsrDev &M, const double *rowvec, const double *colvec_full, double *next_norm, hipStream_t s) {
    hipLaunchKernelGGL((k_scale_matrix<ROW_FIRST, DIVIDE, NEXT>), dim3(M.csr_grid()), dim3(kThreads), 0, s, M, rowvec, colvec_full, next_norm);
    if (M.nlong > 0)
        hipLaunchKernelGGL((k_scale_long_rows<ROW_FIRST, DIVIDE, NEXT>), dim3(M.nlong), dim3(kThreads), 0, s, M, rowvec, colvec_full, next_norm);
}

// next_max_norm (optional; not rowvec): receives what launch_row_norm(M, ., 99) would give on the scaled matrix
void launch_scale_matrix(const CsrDev &M, const double *rowvec, const double *colvec_full, bool row_first,
                         bool divide, hipStream_t s, double *next_max_norm) {
    if (M.rows <= 0 || M.nblk <= 0) return;
    if (next_max_norm) {
        if (row_first && divide) launch_scale_matrix_t<true, true, true>(M, rowvec, colvec_full, next_max_norm, s);
        else if (row_first && !divide) launch_scale_matrix_t<true, false, true>(M, rowvec, colvec_full, next_max_norm, s);
        else if (!row_first && divide) launch_scale_matrix_t<false, true, true>(M, rowvec, colvec_full, next_max_norm, s);
        else launch_scale_matrix_t<false, false, true>(M, rowvec, colvec_full, next_max_norm, s);
        return;
    }
    if (row_first && divide) launch_scale_matrix_t<true, true, false>(M, rowvec, colvec_full, nullptr, s);
    else if (row_first && !divide) launch_scale_matrix_t<true, false, false>(M, rowvec, colvec_full, nullptr, s);
    else if (!row_first && divide) launch_scale_matrix_t<false, true, false>(M, rowvec, colvec_full, nullptr, s);
    else launch_scale_matrix_t<false, false, false>(M, rowvec, colvec_full, nullptr, s);
}

__global__ void __launch_bounds__(kThreads) k_vec_scale(double *x, const double *s, int n, int divide) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i < n) x[i] = divide ? x[i] / s[i] : x[i] * s[i];
}
void launch_vec_scale(double *x, const double *sv, int n, bool divide, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_vec_scale, dim3((n + kThreads - 1) / kThreads), dim3(kThreads), 0, s, x, sv, n, divide ? 1 : 0);
}

__global__ void __launch_bounds__(kThreads) k_vec_scal(double *x, double a, int n) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i < n) x[i] = x[i] * a;
}
void launch_vec_scal(double *x, double a, int n, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_vec_scal, dim3((n + kThreads - 1) / kThreads), dim3(kThreads), 0, s, x, a, n);
}

__global__ void __launch_bounds__(kThreads) k_fill(double *x, double a, int n) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i < n) x[i] = a;
}
void launch_fill(double *x, double a, int n, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_fill, dim3((n + kThreads - 1) / kThreads), dim3(kThreads), 0, s, x, a, n);
}

__global__ void __launch_bounds__(kThreads) k_bnorm2(const double *AL, const double *AU, int m, double *partials) {
    double acc[1] = {0.0};
    const int tid = blockIdx.x * kThreads + threadIdx.x, nth = gridDim.x * kThreads;
    for (int i = tid; i < m; i += nth) {
        double a = AL[i], b = AU[i];
        a = isinf(a) ? 0.0 : a;
        b = isinf(b) ? 0.0 : b;
        const double v = fmax(fabs(a), fabs(b));
        acc[0] += v * v;
    }
    block_store_partials<1>(acc, partials, gridDim.x);
}
void launch_bnorm2(const double *AL, const double *AU, int m, double *partials, int nblocks, hipStream_t s) {
    hipLaunchKernelGGL(k_bnorm2, dim3(nblocks), dim3(kThreads), 0, s, AL, AU, m, partials);
}

__global__ void __launch_bounds__(kThreads) k_norm2(const double *x, int n, double *partials) {
    double acc[1] = {0.0};
    const int tid = blockIdx.x * kThreads + threadIdx.x, nth = gridDim.x * kThreads;
    for (int i = tid; i < n; i += nth) acc[0] += x[i] * x[i];
    block_store_partials<1>(acc, partials, gridDim.x);
}
void launch_norm2(const double *x, int n, double *partials, int nblocks, hipStream_t s) {
    hipLaunchKernelGGL(k_norm2, dim3(nblocks), dim3(kThreads), 0, s, x, n, partials);
}

// which of l[j], u[j] the x-half has to read (kernels.h: XHalfArgs::lu_code); derived from the arrays as they are on the
// device (after scaling: zero stays zero, infinite stays infinite), so a coded bound is bit for bit the stored one
__global__ void __launch_bounds__(kThreads) k_bound_codes(int n, const double *l, const double *u, unsigned char *code) {
    const int j = blockIdx.x * kThreads + threadIdx.x;
    if (j >= n) return;
    const double lj = l[j], uj = u[j];
    const bool l_zero = __double_as_longlong(lj) == 0, l_minf = lj == -__builtin_huge_val(), u_pinf = uj == __builtin_huge_val();
    code[j] = static_cast<unsigned char>((l_zero || l_minf ? 0u : kLoadL) | (u_pinf ? 0u : kLoadU) | (l_zero ? kZeroL : 0u));
}
void launch_bound_codes(int n, const double *l, const double *u, unsigned char *code, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_bound_codes, dim3((n + kThreads - 1) / kThreads), dim3(kThreads), 0, s, n, l, u, code);
}

// which of AL[i], AU[i] the y-half has to read (kernels.h: YHalfArgs::row_code); derived from the arrays as they are on the device
__global__ void __launch_bounds__(kThreads) k_row_codes(int m, const double *AL, const double *AU, unsigned char *code) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= m) return;
    const double lo = AL[i], hi = AU[i];
    const bool lo_minf = lo == -__builtin_huge_val(), hi_pinf = hi == __builtin_huge_val();
    // (equal as bit patterns: the coded lower bound must be the stored one, -0.0 against +0.0 included)
    const bool eq = !hi_pinf && __double_as_longlong(lo) == __double_as_longlong(hi);
    code[i] = static_cast<unsigned char>((lo_minf || eq ? 0u : kLoadLo) | (hi_pinf ? 0u : kLoadHi) | (eq ? kRowEq : 0u));
}
void launch_row_codes(int m, const double *AL, const double *AU, unsigned char *code, hipStream_t s) {
    if (m > 0) hipLaunchKernelGGL(k_row_codes, dim3((m + kThreads - 1) / kThreads), dim3(kThreads), 0, s, m, AL, AU, code);
}

// ------------------------------------------------------------------------------------------------
// power iteration helpers (reference src/power_iteration.cu:60-100)
// ------------------------------------------------------------------------------------------------
// *bad = 1 if some column index lies outside [0, cols) (DeviceMatrix::upload: the kernels index without bounds checks)
__global__ void __launch_bounds__(kThreads) k_check_columns(long nnz, int cols, const int *__restrict__ col, int *bad) {
    const long stride = static_cast<long>(gridDim.x) * kThreads;
    int b = 0;
    for (long k = static_cast<long>(blockIdx.x) * kThreads + threadIdx.x; k < nnz; k += stride) b |= (col[k] < 0 || col[k] >= cols);
    if (b) *bad = 1;
}
void launch_check_columns(long nnz, int cols, const int *col, int *bad, hipStream_t s) {
    if (nnz > 0) hipLaunchKernelGGL(k_check_columns, dim3(static_cast<unsigned>(std::min<long>((nnz + kThreads - 1) / kThreads, 8192))), dim3(kThreads), 0, s, nnz, cols, col, bad);
}

// The power iteration's start vector formed where it is used (host_model.cpp: power_start_vector, same counter generator and
// the same formula; log / cos are the device library's, so elements may differ from the host's in the last place).  For
// vectors of millions of rows: the host fill + copy of config 5's 1e7 rows was 50 ms of a 0.43 s power iteration.
__global__ void __launch_bounds__(kThreads) k_pw_start(int m, unsigned long long seed, long long offset, double *z) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= m) return;
    auto mix = [](unsigned long long x) {
        unsigned long long v = x + 0x9E3779B97F4A7C15ULL;
        v = (v ^ (v >> 30)) * 0xBF58476D1CE4E5B9ULL;
        v = (v ^ (v >> 27)) * 0x94D049BB133111EBULL;
        return v ^ (v >> 31);
    };
    const unsigned long long base = seed * 0x100000001B3ULL, g = static_cast<unsigned long long>(offset + i);
    const unsigned long long h1 = mix(base + 2 * g), h2 = mix(base + 2 * g + 1);
    const double two53 = 1.0 / 9007199254740992.0, twopi = 6.283185307179586476925286766559;
    const double u1 = static_cast<double>((h1 >> 11) + 1) * two53, u2 = static_cast<double>(h2 >> 11) * two53;
    z[i] = sqrt(-2.0 * log(u1)) * cos(twopi * u2) + 1e-8;
}
void launch_pw_start(int m, unsigned long long seed, long long offset, double *z, hipStream_t s) {
    if (m > 0) hipLaunchKernelGGL(k_pw_start, dim3((m + kThreads - 1) / kThreads), dim3(kThreads), 0, s, m, seed, offset, z);
}

__global__ void __launch_bounds__(kThreads) k_pw_normalize(const double *z, double *q, int m, const double *scalars) {
    const double invn = 1.0 / sqrt(scalars[S_PW_ZZ] + 2.220446049250313e-16);
    const int tid = blockIdx.x * kThreads + threadIdx.x, nth = gridDim.x * kThreads;
    for (int i = tid; i < m; i += nth) q[i] = invn * z[i];
}
void launch_pw_normalize(const double *z, double *q, int m, const double *scalars, hipStream_t s) {
    hipLaunchKernelGGL(k_pw_normalize, dim3(vec_grid(m)), dim3(kThreads), 0, s, z, q, m, scalars);
}

__global__ void __launch_bounds__(kThreads) k_pw_err(const double *z, const double *q, int m, const double *scalars,
                                                    double *partials) {
    const double lambda = scalars[S_PW_QZ];
    double acc[1] = {0.0};
    const int tid = blockIdx.x * kThreads + threadIdx.x, nth = gridDim.x * kThreads;
    for (int i = tid; i < m; i += nth) {
        const double d = -lambda * q[i] + 1.0 * z[i];
        acc[0] += d * d;
    }
    block_store_partials<1>(acc, partials, gridDim.x);
}
void launch_pw_err(const double *z, const double *q, int m, const double *scalars, double *partials, int nblocks,
                   hipStream_t s) {
    hipLaunchKernelGGL(k_pw_err, dim3(nblocks), dim3(kThreads), 0, s, z, q, m, scalars, partials);
}

// ------------------------------------------------------------------------------------------------
// unscale (reference collect_solution, src/utils.cu:172-189)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads) k_unscale(int n, int m, const double *x_bar, const double *y_bar,
                                                     const double *z_bar, const double *col_norm,
                                                     const double *row_norm, double b_scale, double c_scale, double *xo,
                                                     double *yo, double *zo) {
    const int tid = blockIdx.x * kThreads + threadIdx.x, nth = gridDim.x * kThreads;
    for (int i = tid; i < n; i += nth) {
        xo[i] = (x_bar[i] / col_norm[i]) * b_scale;
        zo[i] = (z_bar[i] * col_norm[i]) * c_scale;
    }
    for (int i = tid; i < m; i += nth) yo[i] = (y_bar[i] / row_norm[i]) * c_scale;
}
void launch_unscale(int n, int m, const double *x_bar, const double *y_bar, const double *z_bar,
                    const double *col_norm, const double *row_norm, double b_scale, double c_scale, double *xo,
                    double *yo, double *zo, hipStream_t s) {
    hipLaunchKernelGGL(k_unscale, dim3(vec_grid(n > m ? n : m)), dim3(kThreads), 0, s, n, m, x_bar, y_bar, z_bar,
                       col_norm, row_norm, b_scale, c_scale, xo, yo, zo);
}

// ------------------------------------------------------------------------------------------------
// multi-GPU neighbour exchange (HaloPlan, solver.h): the index lists are sorted, so both kernels walk
// the big vector monotonically
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads) k_pack(const double *__restrict__ src, const int *__restrict__ idx,
                                                  double *__restrict__ dst, int n) {
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) dst[i] = src[idx[i]];
}
__global__ void __launch_bounds__(kThreads) k_scatter(double *__restrict__ dst, const int *__restrict__ idx,
                                                     const double *__restrict__ src, int n) {
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) dst[idx[i]] = src[i];
}
void launch_pack(const double *src, const int *idx, double *dst, int n, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_pack, dim3(vec_grid(n)), dim3(kThreads), 0, s, src, idx, dst, n);
}
void launch_scatter(double *dst, const int *idx, const double *src, int n, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_scatter, dim3(vec_grid(n)), dim3(kThreads), 0, s, dst, idx, src, n);
}

// warm-up (abi.cpp: hprlp_warmup): an attribute query makes the runtime load this translation unit's code object now instead
// of at the first launch of one of its kernels
void warm_kernels_tu() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&k_finalize));
}

}  // namespace hprlp
