// batched.hip -- solve_batched: B linear programs that share the sparse matrix A, solved together on
// one MI355X.  Replaces reference src/batched_solver.cu (kernels :122-323, SpMM wrappers :428-477,
// loop :1017-1084, host restart/sigma logic :667-762, set-up :792-885, results :887-935).
//
// Layout: the batch (padded to Bp = a power of two <= 64, or a multiple of 64) is cut into chunks of Bc problems
// (Bc = Bp below 64, else 8..64: choose_chunk); a panel of `rows` rows is stored chunk after chunk, each chunk
// ROW-major in the batch index: element (row j, problem k) lives at P[((k / Bc) * rows + j) * Bc + k % Bc].
// A gathered row of a chunk is one contiguous Bc*8-byte run: the SpMM is a CSR row loop in which a lane follows
// one problem, 64 / Bc rows per wave.  A workgroup works on ONE chunk, chunk = blockIdx.x % (number of chunks):
// workgroups go round-robin to the 8 XCDs, so with 8 chunks every XCD gathers from the same eighth of the
// gathered panel (config 4: 2.2 MB of Y instead of 17 MB -- it stays in the XCD's 4 MiB L2).  Each problem's row sums are accumulated sequentially in CSR order, exactly like
// the single-LP stream kernel, so the result is bit-identical to the oracle's batched restatement.
// The half-step update (projection, reflection, Halpern average, per-problem sigma / inner counter /
// active mask) is fused into the SpMM epilogue: one launch per half-step, no per-iteration host sync
// (the reference synchronises the stream and uploads 2B doubles every iteration, :1070-1073).
// The MFMA f64 tile (v_mfma_f64_16x16x4) is not used: with ~2-7 nonzeros per row there is no dense
// A-tile to feed it and the kernel is bound by panel traffic, not by FMA rate.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <iostream>
#include <limits>
#include <string>
#include <type_traits>

#include "HPRLP.h"
#include "env.h"
#include "solver.h"

namespace hprlp {
namespace {

constexpr double kInfReplacement = 1.0e100;  // reference batched_solver.cu:17

enum BSlot : int {  // per-problem scalar slots, SC[slot*Bp + k]
    B_CX = 0, B_YOBJ_Y, B_XZ, B_RD2, B_RP2, B_ADX_DY, B_DY2, B_DX2, B_MOVE_X2, B_MOVE_Y2, B_LU2, B_NSLOT
};

struct BatchCtl {  // per-problem device scalars
    double *sigma;
    int *active;
    int *kx, *ky;
    int *restart_flag;
};

// thread -> (row slot, problem): lane l of a wave handles sub-row l / Bw and problem chunk*Bw + l % Bw, Bw = Bc = the
// chunk width; a 256-thread block covers 4 * (64/Bw) rows of one chunk.
struct Geo {
    int Bp, Bw, nchunk, rows_per_wave, rows_per_block;
};
inline Geo make_geo(int Bp, int Bc) {
    Geo g;
    g.Bp = Bp;
    g.Bw = Bc;
    g.nchunk = Bp / Bc;
    g.rows_per_wave = 64 / g.Bw;
    g.rows_per_block = 4 * g.rows_per_wave;
    return g;
}
// which chunk / which block of rows a workgroup works on (1-D grids of nchunk * row blocks, chunk fastest)
struct Blk {
    int chunk, rb, nrb;
};
__device__ __forceinline__ Blk decode_block(const Geo &g) {
    Blk b;
    b.chunk = blockIdx.x % g.nchunk;
    b.rb = blockIdx.x / g.nchunk;
    b.nrb = gridDim.x / g.nchunk;
    return b;
}
// element (row r, local problem kl) of a chunk of a panel with `rows` rows
__device__ __forceinline__ size_t pidx(const Geo &g, int chunk, int rows, int r, int kl) {
    return (static_cast<size_t>(chunk) * rows + r) * g.Bw + kl;
}

// Sum `NACC` per-thread accumulators over all threads of the block that share a problem index and
// store them to partials[(rb * NACC + i) * Bp + k] (rb: the workgroup's row block).  Fixed order => deterministic.
template <int NACC>
__device__ __forceinline__ void block_store_per_problem(double (&acc)[NACC], const Geo &g, int rb, int k, bool kvalid,
                                                        double *partials) {
    __shared__ double red[4][NACC][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
        double v = acc[i];
        for (int off = 32; off >= g.Bw; off >>= 1) v += __shfl_xor(v, off, 64);  // combine sub-rows of the wave
        red[wave][i][lane] = v;
    }
    __syncthreads();
    if (wave == 0 && lane < g.Bw && kvalid) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            const double v = ((red[0][i][lane] + red[1][i][lane]) + red[2][i][lane]) + red[3][i][lane];
            partials[(static_cast<size_t>(rb) * NACC + i) * g.Bp + k] = v;
        }
    }
}

// ---- fused SpMM + half-step ---------------------------------------------------------------------
// XHALF: M = A^T (n rows), V = Y.  reference update_x_z_{check,normal}_batched_kernel :122-178
// else : M = A   (m rows), V = X_hat.  reference update_y_{check,normal}_batched_kernel :180-236
// cache policy of the panel streams (measured on config 4, profiles/r02_pmc_summary.md)
#ifndef HPRLP_BATCH_NT
#define HPRLP_BATCH_NT 1  // 1: nontemporal loads of the panel streams (6390 -> 6800 batch-it/s on config 4); 2: also store X nontemporal (no difference)
#endif
constexpr bool kNtPanels = HPRLP_BATCH_NT != 0;
constexpr bool kNtStoreX = HPRLP_BATCH_NT >= 2;

struct HalfArgs {
    const double *V;                   // gathered panel
    int vrows;                         // its rows (= columns of the matrix)
    double *P, *P_hat;                 // X / X_hat  or  Y / (unused)
    const double *lo, *hi, *cost;      // L,U,C  or  AL,AU,(unused)
    const double *last;
    double *bar, *aux, *delta;         // X_bar, Z_bar, DX  or  Y_bar, Y_obj, DY  (check only)
    BatchCtl ctl;
    double lambda_max;
    double *partials;
    const int *order;                  // kb_half64: row group handled at position g of the launch (null: g itself)
};

// the half-step update of one (row, problem) element given the SpMM row sum s
template <bool XHALF, bool CHECK, int NACC>
__device__ __forceinline__ void half_update(const HalfArgs &a, size_t t, double s, double p_i, double p_lo, double p_hi,
                                            double p_last, double p_cost, double sig, double fact1, double f1,
                                            double f2, double (&acc)[NACC]) {
    if (XHALF) {
        const double xi = p_i;
        const double zt = xi + sig * (s - p_cost);
        const double xb = fmin(fmax(zt, p_lo), p_hi);
        const double xh = 2.0 * xb - xi;
        a.P_hat[t] = xh;  // gathered by the y-half that follows: default policy
        if (kNtStoreX) __builtin_nontemporal_store(f2 * xh + f1 * p_last, a.P + t);  // next read: the next iteration's x-half
        else a.P[t] = f2 * xh + f1 * p_last;
        if (CHECK) {
            const double zb = (xb - zt) / sig, dx = xb - xh;
            a.delta[t] = dx;
            a.aux[t] = zb;
            a.bar[t] = xb;
            acc[0] += p_cost * xb;
            acc[1 % NACC] += xb * zb;
            acc[2 % NACC] += dx * dx;
        }
    } else {
        const double yi = p_i;
        const double v = s - fact1 * yi;
        const double d = fmax(p_lo - v, fmin(p_hi - v, 0.0));
        const double yb = d / fact1;
        const double yh = 2.0 * yb - yi;
        a.P[t] = f2 * yh + f1 * p_last;
        if (CHECK) {
            const double dy = yb - yh, yo = v + d;
            a.delta[t] = dy;
            a.bar[t] = yb;
            a.aux[t] = yo;
            acc[0] += yo * yb;
            acc[1 % NACC] += dy * dy;
        }
    }
}

template <bool XHALF, bool CHECK>
__global__ void __launch_bounds__(256) kb_half(int rows, const int *__restrict__ rowptr, const int *__restrict__ col,
                                               const double *__restrict__ val, Geo g, HalfArgs a) {
    const Blk blk = decode_block(g);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int kl = lane % g.Bw, k = blk.chunk * g.Bw + kl;
    const int sub = lane / g.Bw;
    constexpr int NACC = CHECK ? (XHALF ? 3 : 2) : 1;
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0.0;

    const bool act = a.ctl.active[k] != 0;
    const int kk = XHALF ? a.ctl.kx[k] : a.ctl.ky[k];
    const double sig = a.ctl.sigma[k];
    const double f1 = 1.0 / (static_cast<double>(kk) + 2.0), f2 = 1.0 - f1;
    const double fact1 = a.lambda_max * sig;
    // counter hand-off (see Ctrl in kernels.h): the x-half publishes ky, the y-half advances kx
    if (blk.rb == 0 && wave == 0 && sub == 0) {
        if (XHALF) a.ctl.ky[k] = kk;
        else if (act) a.ctl.kx[k] = kk + 1;
    }
    const double *__restrict__ V = a.V + pidx(g, blk.chunk, a.vrows, 0, kl);
    for (int r = blk.rb * g.rows_per_block + wave * g.rows_per_wave + sub; r < rows; r += blk.nrb * g.rows_per_block) {
        if (!act) continue;
        // the panel operands do not depend on the row sum: issue their loads first so that they
        // overlap the gather chain of the SpMM loop
        const size_t t = pidx(g, blk.chunk, rows, r, kl);
        const double p_i = a.P[t], p_lo = a.lo[t], p_hi = a.hi[t], p_last = a.last[t];
        const double p_cost = XHALF ? a.cost[t] : 0.0;
        double s = 0.0;
        const int e = rowptr[r + 1];
        for (int p = rowptr[r]; p < e; p += 4) {  // four gathers in flight, summed in CSR order
            double av[4], gv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = min(p + u, e - 1);
                av[u] = val[q];
                gv[u] = V[static_cast<size_t>(col[q]) * g.Bw];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (p + u < e) s += av[u] * gv[u];
        }
        half_update<XHALF, CHECK, NACC>(a, t, s, p_i, p_lo, p_hi, p_last, p_cost, sig, fact1, f1, f2, acc);
    }
    if (CHECK) block_store_per_problem<NACC>(acc, g, blk.rb, k, true, a.partials);
}

// Chunks of 64 problems (a wave = 64 problems of ONE row): the row index is wave-uniform, so row pointers, column
// indices and values come through the scalar cache, and a wave works on kRowsPerWave consecutive rows
// at once -- their nonzeros are one contiguous CSR range -- with all panel loads and up to 8 gathers
// in flight before the first use.  Each row is still summed in CSR order.
constexpr int kRowsPerWave = 4;

template <bool XHALF, bool CHECK>
__global__ void __launch_bounds__(256) kb_half64(int rows, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                 const double *__restrict__ val, Geo g, HalfArgs a) {
    constexpr int RW = kRowsPerWave, G = 8;
    static_assert(RW == 4, "the row select below is written for 4 rows");
    const Blk blk = decode_block(g);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int k = blk.chunk * 64 + lane;
    constexpr int NACC = CHECK ? (XHALF ? 3 : 2) : 1;
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0.0;

    const bool act = a.ctl.active[k] != 0;
    const int kk = XHALF ? a.ctl.kx[k] : a.ctl.ky[k];
    const double sig = a.ctl.sigma[k];
    const double f1 = 1.0 / (static_cast<double>(kk) + 2.0), f2 = 1.0 - f1;
    const double fact1 = a.lambda_max * sig;
    if (blk.rb == 0 && wave == 0) {
        if (XHALF) a.ctl.ky[k] = kk;
        else if (act) a.ctl.kx[k] = kk + 1;
    }
    const double *__restrict__ V = a.V + pidx(g, blk.chunk, a.vrows, 0, lane);
    // Row groups in launch order: the few groups with long rows first (BatchWS::order_*).  A 200-entry row is 26 dependent
    // trips to memory for its wave (about 50 us): dispatched wherever it falls in the row order it ends up as the launch's
    // tail (config 4: 53 such rows of A^T, x-half 101 -> 71 us without them); dispatched first it runs beside everything else.
    const int ngroups = (rows + RW - 1) / RW;
    for (int gi = __builtin_amdgcn_readfirstlane(blk.rb * 4 + wave); gi < ngroups; gi += blk.nrb * 4) {
        const int rb = (a.order ? a.order[gi] : gi) * RW;
        int pb[RW + 1];
#pragma unroll
        for (int i = 0; i <= RW; ++i) pb[i] = rowptr[min(rb + i, rows)];
        double p_i[RW], p_lo[RW], p_hi[RW], p_last[RW], p_cost[RW], s[RW];
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            const size_t t = pidx(g, blk.chunk, rows, min(rb + i, rows - 1), lane);
            if (kNtPanels) {  // the panel streams are read once per half-step: keep them out of the gathered panel's way in the caches
                p_i[i] = __builtin_nontemporal_load(a.P + t), p_lo[i] = __builtin_nontemporal_load(a.lo + t);
                p_hi[i] = __builtin_nontemporal_load(a.hi + t), p_last[i] = __builtin_nontemporal_load(a.last + t);
                p_cost[i] = XHALF ? __builtin_nontemporal_load(a.cost + t) : 0.0;
            } else {
                p_i[i] = a.P[t], p_lo[i] = a.lo[t], p_hi[i] = a.hi[t], p_last[i] = a.last[t];
                p_cost[i] = XHALF ? a.cost[t] : 0.0;
            }
            s[i] = 0.0;
        }
        const int pend = pb[RW];
        for (int p = pb[0]; p < pend; p += G) {
            double gv[G], av[G];
#pragma unroll
            for (int u = 0; u < G; ++u) {
                const int q = min(p + u, pend - 1);
                av[u] = val[q];
                gv[u] = V[static_cast<size_t>(col[q]) * 64];
            }
#pragma unroll
            for (int u = 0; u < G; ++u) {
                const int q = p + u;
                if (q < pend) {
                    const double prod = av[u] * gv[u];
                    if (q < pb[1]) s[0] += prod;
                    else if (q < pb[2]) s[1] += prod;
                    else if (q < pb[3]) s[2] += prod;
                    else s[3] += prod;
                }
            }
        }
        if (act) {
#pragma unroll
            for (int i = 0; i < RW; ++i)
                if (rb + i < rows)
                    half_update<XHALF, CHECK, NACC>(a, pidx(g, blk.chunk, rows, rb + i, lane), s[i], p_i[i], p_lo[i], p_hi[i],
                                                    p_last[i], p_cost[i], sig, fact1, f1, f2, acc);
        }
    }
    if (CHECK) block_store_per_problem<NACC>(acc, g, blk.rb, k, true, a.partials);
}

// Chunks of BW = 8 / 16 / 32 problems: a wave is SUBS = 64 / BW lane groups; lane group `sub` of a wave that works on the
// 4 * SUBS consecutive rows from r0 takes rows r0 + i * SUBS + sub, i = 0..3, so that every panel load of the wave (fixed i)
// reads SUBS consecutive rows = one contiguous 512-byte run.  Same structure as kb_half64 otherwise -- all panel loads and up
// to 8 gathers in flight per lane before the first use, every row summed in CSR order -- but the CSR arrays come through
// vector loads (the rows are uniform per lane group, not per wave) and a lane group's entries are four separate CSR ranges,
// walked as one concatenated range.
template <bool XHALF, bool CHECK, int BW>
__global__ void __launch_bounds__(256) kb_halfN(int rows, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                const double *__restrict__ val, Geo g, HalfArgs a) {
    constexpr int RW = kRowsPerWave, G = 8, SUBS = 64 / BW;
    static_assert(RW == 4, "the row select below is written for 4 rows");
    const Blk blk = decode_block(g);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int kl = lane % BW, sub = lane / BW, k = blk.chunk * BW + kl;
    constexpr int NACC = CHECK ? (XHALF ? 3 : 2) : 1;
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0.0;

    const bool act = a.ctl.active[k] != 0;
    const int kk = XHALF ? a.ctl.kx[k] : a.ctl.ky[k];
    const double sig = a.ctl.sigma[k];
    const double f1 = 1.0 / (static_cast<double>(kk) + 2.0), f2 = 1.0 - f1;
    const double fact1 = a.lambda_max * sig;
    if (blk.rb == 0 && wave == 0 && sub == 0) {
        if (XHALF) a.ctl.ky[k] = kk;
        else if (act) a.ctl.kx[k] = kk + 1;
    }
    const double *__restrict__ V = a.V + pidx(g, blk.chunk, a.vrows, 0, kl);
    const int ngroups = (rows + RW * SUBS - 1) / (RW * SUBS);
    for (int gi = blk.rb * 4 + wave; gi < ngroups; gi += blk.nrb * 4) {
        const int r0 = (a.order ? a.order[gi] : gi) * (RW * SUBS) + sub;
        int pb[RW], pe[RW];
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            const int r = min(r0 + i * SUBS, rows);
            pb[i] = rowptr[r];
            pe[i] = rowptr[min(r + 1, rows)];
        }
        double p_i[RW], p_lo[RW], p_hi[RW], p_last[RW], p_cost[RW], s[RW];
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            const size_t t = pidx(g, blk.chunk, rows, min(r0 + i * SUBS, rows - 1), kl);
            if (kNtPanels) {
                p_i[i] = __builtin_nontemporal_load(a.P + t), p_lo[i] = __builtin_nontemporal_load(a.lo + t);
                p_hi[i] = __builtin_nontemporal_load(a.hi + t), p_last[i] = __builtin_nontemporal_load(a.last + t);
                p_cost[i] = XHALF ? __builtin_nontemporal_load(a.cost + t) : 0.0;
            } else {
                p_i[i] = a.P[t], p_lo[i] = a.lo[t], p_hi[i] = a.hi[t], p_last[i] = a.last[t];
                p_cost[i] = XHALF ? a.cost[t] : 0.0;
            }
            s[i] = 0.0;
        }
        // position e of the concatenated range -> CSR position e + (offset of the row e falls in)
        const int c1 = pe[0] - pb[0], c2 = c1 + (pe[1] - pb[1]), c3 = c2 + (pe[2] - pb[2]), total = c3 + (pe[3] - pb[3]);
        const int d0 = pb[0], d1 = pb[1] - c1, d2 = pb[2] - c2, d3 = pb[3] - c3;
        for (int p = 0; p < total; p += G) {
            double gv[G], av[G];
#pragma unroll
            for (int u = 0; u < G; ++u) {
                const int e = min(p + u, total - 1);
                const int q = e + (e < c1 ? d0 : e < c2 ? d1 : e < c3 ? d2 : d3);
                av[u] = val[q];
                gv[u] = V[static_cast<size_t>(col[q]) * BW];
            }
#pragma unroll
            for (int u = 0; u < G; ++u) {
                const int e = p + u;
                if (e < total) {
                    const double prod = av[u] * gv[u];
                    if (e < c1) s[0] += prod;
                    else if (e < c2) s[1] += prod;
                    else if (e < c3) s[2] += prod;
                    else s[3] += prod;
                }
            }
        }
        if (act) {
#pragma unroll
            for (int i = 0; i < RW; ++i)
                if (r0 + i * SUBS < rows)
                    half_update<XHALF, CHECK, NACC>(a, pidx(g, blk.chunk, rows, r0 + i * SUBS, kl), s[i], p_i[i], p_lo[i], p_hi[i],
                                                    p_last[i], p_cost[i], sig, fact1, f1, f2, acc);
        }
    }
    if (CHECK) block_store_per_problem<NACC>(acc, g, blk.rb, k, true, a.partials);
}

// ---- residual SpMMs (reference compute_batched_Rd/Rp_kernel :238-263 + SpMM) ---------------------
// WHICH 0: |(C - A^T Ybar - Zbar) .* col_norm|^2 ; 1: |Rp|^2 ; 2: |Rp|^2 and <A DX, DY> ; 3: <A DX, DY>
template <int WHICH>
__global__ void __launch_bounds__(256) kb_resid(int rows, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                const double *__restrict__ val, Geo g, int vrows, const double *V,
                                                const double *V2, const double *p0, const double *p1,
                                                const double *norm, const double *dvec, double *partials) {
    const Blk blk = decode_block(g);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int kl = lane % g.Bw, k = blk.chunk * g.Bw + kl;
    const int sub = lane / g.Bw;
    const size_t vbase = pidx(g, blk.chunk, vrows, 0, kl);
    constexpr int NACC = (WHICH == 2) ? 2 : 1;
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
    for (int r = blk.rb * g.rows_per_block + wave * g.rows_per_wave + sub; r < rows; r += blk.nrb * g.rows_per_block) {
        double s = 0.0, s2 = 0.0;
        const int e = rowptr[r + 1];
        for (int p = rowptr[r]; p < e; p += 4) {  // four entries in flight (a dependent trip per entry made long rows a 300 us tail); summed in CSR order
            double av[4], g1[4], g2[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = min(p + u, e - 1);
                const size_t gi = vbase + static_cast<size_t>(col[q]) * g.Bw;
                av[u] = val[q];
                g1[u] = WHICH != 3 ? V[gi] : 0.0;
                g2[u] = WHICH >= 2 ? V2[gi] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (p + u < e) {
                    if (WHICH != 3) s += av[u] * g1[u];
                    if (WHICH >= 2) s2 += av[u] * g2[u];
                }
        }
        const size_t t = pidx(g, blk.chunk, rows, r, kl);
        if (WHICH == 0) {
            const double rd = (p0[t] - s - p1[t]) * norm[r];
            acc[0] += rd * rd;
        } else if (WHICH == 1 || WHICH == 2) {
            const double rp = norm[r] * fmax(fmin(p1[t] - s, 0.0), p0[t] - s);
            acc[0] += rp * rp;
            if (WHICH == 2) acc[1 % NACC] += s2 * dvec[t];
        } else {
            acc[0] += s2 * dvec[t];
        }
    }
    block_store_per_problem<NACC>(acc, g, blk.rb, k, true, partials);
}

// iteration-0 bound violation (reference compute_batched_lu_violation_kernel :265-278)
__global__ void __launch_bounds__(256) kb_lu(int n, Geo g, const double *Xb, const double *L, const double *U,
                                             const double *col_norm, double *DX, double *partials) {
    const Blk blk = decode_block(g);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int kl = lane % g.Bw, k = blk.chunk * g.Bw + kl;
    const int sub = lane / g.Bw;
    double acc[1] = {0.0};
    for (int r = blk.rb * g.rows_per_block + wave * g.rows_per_wave + sub; r < n; r += blk.nrb * g.rows_per_block) {
        const size_t t = pidx(g, blk.chunk, n, r, kl);
        const double x = Xb[t];
        const double viol = x < L[t] ? L[t] - x : (x > U[t] ? x - U[t] : 0.0);
        const double v = viol / col_norm[r];
        DX[t] = v;
        acc[0] += v * v;
    }
    block_store_per_problem<1>(acc, g, blk.rb, k, true, partials);
}

// DX = Xbar - lastX, DY = Ybar - lastY for ALL problems, with their squared norms
// (reference batched_restart_movement_kernel :280-294 + the per-problem nrm2 calls :662-663)
__global__ void __launch_bounds__(256) kb_movement(int n, int m, Geo g, const double *Xb, const double *lastX,
                                                   double *DX, const double *Yb, const double *lastY, double *DY,
                                                   double *partials) {
    const Blk blk = decode_block(g);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int kl = lane % g.Bw, k = blk.chunk * g.Bw + kl;
    const int sub = lane / g.Bw;
    double acc[2] = {0.0, 0.0};
    const int r0 = blk.rb * g.rows_per_block + wave * g.rows_per_wave + sub, rs = blk.nrb * g.rows_per_block;
    for (int r = r0; r < n; r += rs) {
        const size_t t = pidx(g, blk.chunk, n, r, kl);
        const double d = Xb[t] - lastX[t];
        DX[t] = d;
        acc[0] += d * d;
    }
    for (int r = r0; r < m; r += rs) {
        const size_t t = pidx(g, blk.chunk, m, r, kl);
        const double d = Yb[t] - lastY[t];
        DY[t] = d;
        acc[1] += d * d;
    }
    block_store_per_problem<2>(acc, g, blk.rb, k, true, partials);
}

// where restart_flag[k]: X = lastX = Xbar, Y = lastY = Ybar, inner counter reset
// (reference do_batched_restart_kernel :296-323)
__global__ void __launch_bounds__(256) kb_restart(int n, int m, Geo g, double *X, double *lastX, const double *Xb,
                                                  double *Y, double *lastY, const double *Yb, BatchCtl ctl) {
    const Blk blk = decode_block(g);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int kl = lane % g.Bw, k = blk.chunk * g.Bw + kl;
    const int sub = lane / g.Bw;
    if (!ctl.restart_flag[k]) return;
    const int r0 = blk.rb * g.rows_per_block + wave * g.rows_per_wave + sub, rs = blk.nrb * g.rows_per_block;
    for (int r = r0; r < n; r += rs) {
        const size_t t = pidx(g, blk.chunk, n, r, kl);
        const double v = Xb[t];
        X[t] = v;
        lastX[t] = v;
    }
    for (int r = r0; r < m; r += rs) {
        const size_t t = pidx(g, blk.chunk, m, r, kl);
        const double v = Yb[t];
        Y[t] = v;
        lastY[t] = v;
    }
    if (blk.rb == 0 && wave == 0 && sub == 0 && ctl.active[k]) {
        ctl.kx[k] = 0;
        ctl.ky[k] = 0;
    }
}

// SC[slot[i]*Bp + k] = sum over blocks of partials[(b*nacc + i)*Bp + k]
struct BFin {
    int slot[3];
    int nacc;
};
// grid (ceil(Bp/64), nacc): a block sums one accumulator for 64 problems; its 4 waves stride over the
// producer blocks (lane = problem: coalesced), then combine in a fixed order
__global__ void __launch_bounds__(1024) kb_finalize(const double *partials, int nblocks, int Bp, BFin f, double *SC) {
    // 16 waves, four independent load chains per wave; fixed combination order => deterministic
    __shared__ double red[16][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int k = blockIdx.x * 64 + lane, i = blockIdx.y;
    double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
    if (k < Bp) {
        const double *__restrict__ p = partials + static_cast<size_t>(i) * Bp + k;
        const size_t st = static_cast<size_t>(f.nacc) * Bp;
        int b = wave;
        for (; b + 48 < nblocks; b += 64) {
            v0 += p[b * st];
            v1 += p[(b + 16) * st];
            v2 += p[(b + 32) * st];
            v3 += p[(b + 48) * st];
        }
        for (; b < nblocks; b += 16) v0 += p[b * st];
    }
    red[wave][lane] = (v0 + v1) + (v2 + v3);
    __syncthreads();
    if (wave == 0 && k < Bp) {
        double v = red[0][lane];
#pragma unroll
        for (int w = 1; w < 16; ++w) v += red[w][lane];
        SC[static_cast<size_t>(f.slot[i]) * Bp + k] = v;
    }
}

// ---- host side ----------------------------------------------------------------------------------
struct BatchWS {
    int m = 0, n = 0, B = 0, Bp = 0;
    Solver *shared = nullptr;  // scaled A, A^T, row_norm, col_norm
    DBuf<double> C, AL, AU, L, U;
    DBuf<double> X, Xh, Xb, DX, Zb, lastX, Y, Yb, DY, Yobj, lastY;
    DBuf<double> sigma, SC, partials;
    DBuf<int> active, kx, ky, rflag;
    DBuf<int> order_x, order_y;  // launch order of the 4-row groups of A^T / A in kb_half64 (groups with long rows first)
    HBuf<double> SC_h;
    BatchCtl ctl{};
    double lambda_max = 1.0;
    int gx = 1, gy = 1;  // row blocks of the n- / m-row launches with partials (the grid is geo.nchunk times that)
    Geo geo{};
    hipStream_t stream = nullptr;
    std::map<int, hipGraphExec_t> graphs;
    ~BatchWS() {
        for (auto &kv : graphs) (void)hipGraphExecDestroy(kv.second);
    }
};

int padded_batch(int B) {
    if (B <= 64) {
        int p = 1;
        while (p < B) p <<= 1;
        return p;
    }
    return (B + 63) / 64 * 64;
}

int grid_for(int rows, const Geo &g) {
    long need = (static_cast<long>(rows) + g.rows_per_block - 1) / g.rows_per_block;
    return static_cast<int>(std::max(1L, std::min(need, 2048L)));
}

void finalize(BatchWS &w, int nblocks, std::initializer_list<int> slots) {
    BFin f{};
    f.nacc = 0;
    for (int s : slots) f.slot[f.nacc++] = s;
    hipLaunchKernelGGL(kb_finalize, dim3((w.Bp + 63) / 64, f.nacc), dim3(1024), 0, w.stream, w.partials.p, nblocks, w.Bp, f,
                       w.SC.p);
}

void launch_half_pair(BatchWS &w, bool check) {
    const CsrDev &A = w.shared->A.view, &AT = w.shared->AT.view;
    HalfArgs xa{w.Y.p, w.m, w.X.p, w.Xh.p, w.L.p, w.U.p, w.C.p, w.lastX.p, w.Xb.p, w.Zb.p, w.DX.p, w.ctl, w.lambda_max, w.partials.p, w.order_x.p};
    HalfArgs ya{w.Xh.p, w.n, w.Y.p, nullptr, w.AL.p, w.AU.p, nullptr, w.lastY.p, w.Yb.p, w.Yobj.p, w.DY.p, w.ctl, w.lambda_max, w.partials.p, w.order_y.p};
    const Geo &g = w.geo;
    const dim3 gxd(w.gx * g.nchunk), gyd(w.gy * g.nchunk), blk(256);
    // kernel by chunk width: 64 -> a wave = one row (kb_half64); 8 / 16 / 32 -> lane groups with their own rows (kb_halfN);
    // below: the plain row loop (kb_half)
    auto launch = [&](auto xhalf, auto check, dim3 grid, const CsrDev &M, const HalfArgs &ha) {
        constexpr bool X = decltype(xhalf)::value, C = decltype(check)::value;
        switch (g.Bw) {
            case 64: hipLaunchKernelGGL((kb_half64<X, C>), grid, blk, 0, w.stream, M.rows, M.rowptr, M.col, M.val, g, ha); break;
            case 32: hipLaunchKernelGGL((kb_halfN<X, C, 32>), grid, blk, 0, w.stream, M.rows, M.rowptr, M.col, M.val, g, ha); break;
            case 16: hipLaunchKernelGGL((kb_halfN<X, C, 16>), grid, blk, 0, w.stream, M.rows, M.rowptr, M.col, M.val, g, ha); break;
            case 8: hipLaunchKernelGGL((kb_halfN<X, C, 8>), grid, blk, 0, w.stream, M.rows, M.rowptr, M.col, M.val, g, ha); break;
            default: hipLaunchKernelGGL((kb_half<X, C>), grid, blk, 0, w.stream, M.rows, M.rowptr, M.col, M.val, g, ha); break;
        }
    };
    using T = std::true_type;
    using F = std::false_type;
    if (check) {
        launch(T{}, T{}, gxd, AT, xa);
        finalize(w, w.gx, {B_CX, B_XZ, B_DX2});
        launch(F{}, T{}, gyd, A, ya);
        finalize(w, w.gy, {B_YOBJ_Y, B_DY2});
    } else {
        // no reduction partials in the normal variant: one pass over the rows, as many workgroups as rows need
        const int rpb = g.Bw >= 8 ? 4 * kRowsPerWave * (64 / g.Bw) : g.rows_per_block;
        static const int grid_cap = env_get("HPRLP_BATCH_GRID") ? std::atoi(env_get("HPRLP_BATCH_GRID")) : 0;  // experiment knob
        auto cap = [&](int gr) { return grid_cap > 0 ? std::min(gr, grid_cap) : gr; };
        launch(T{}, F{}, dim3(cap((AT.rows + rpb - 1) / rpb) * g.nchunk), AT, xa);
        launch(F{}, F{}, dim3(cap((A.rows + rpb - 1) / rpb) * g.nchunk), A, ya);
    }
}

void run_normal(BatchWS &w, int count) {
    while (count > 0) {
        const int len = std::min(count, 32);
        auto it = w.graphs.find(len);
        hipGraphExec_t ge;
        if (it == w.graphs.end()) {
            hipGraph_t g = nullptr;
            HIP_CHECK(hipStreamBeginCapture(w.stream, hipStreamCaptureModeThreadLocal));
            for (int i = 0; i < len; ++i) launch_half_pair(w, false);
            HIP_CHECK(hipStreamEndCapture(w.stream, &g));
            HIP_CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            HIP_CHECK(hipGraphDestroy(g));
            w.graphs[len] = ge;
        } else {
            ge = it->second;
        }
        HIP_CHECK(hipGraphLaunch(ge, w.stream));
        count -= len;
    }
}

void fetch(BatchWS &w) {
    HIP_CHECK(hipMemcpyAsync(w.SC_h.p, w.SC.p, sizeof(double) * B_NSLOT * w.Bp, hipMemcpyDeviceToHost, w.stream));
    HIP_CHECK(hipStreamSynchronize(w.stream));
}
inline double sc(const BatchWS &w, int slot, int k) { return w.SC_h.p[static_cast<size_t>(slot) * w.Bp + k]; }

// reference compute_weighted_norm :626-666.  DX/DY norms come from the slots the check step filled,
// unless a movement pass has overwritten DX/DY since (then B_MOVE_* hold the matching norms).
void weighted_norm(BatchWS &w, bool dxdy_from_movement, std::vector<double> &sigma, std::vector<double> &out) {
    const CsrDev &A = w.shared->A.view;
    hipLaunchKernelGGL((kb_resid<3>), dim3(w.gy * w.geo.nchunk), dim3(256), 0, w.stream, A.rows, A.rowptr, A.col, A.val, w.geo,
                       w.n, static_cast<const double *>(nullptr), w.DX.p, static_cast<const double *>(nullptr),
                       static_cast<const double *>(nullptr), static_cast<const double *>(nullptr), w.DY.p, w.partials.p);
    finalize(w, w.gy, {B_ADX_DY});
    fetch(w);
    out.assign(w.B, 0.0);
    for (int k = 0; k < w.B; ++k) {
        const double dot_prod = 2.0 * sc(w, B_ADX_DY, k);
        // the reference squares cublasDnrm2 results (:653-654); sqrt-then-square keeps that rounding
        const double dyn = std::sqrt(sc(w, dxdy_from_movement ? B_MOVE_Y2 : B_DY2, k));
        const double dxn = std::sqrt(sc(w, dxdy_from_movement ? B_MOVE_X2 : B_DX2, k));
        const double dy_sq = dyn * dyn, dx_sq = dxn * dxn;
        double value = sigma[k] * (w.lambda_max * dy_sq) + dx_sq / sigma[k] + dot_prod;
        if (value < 0.0 && dy_sq > 0.0) {
            const double cand = -(dot_prod + dx_sq / sigma[k]) / (sigma[k] * dy_sq) * 1.05;
            w.lambda_max = std::max(w.lambda_max, cand);
            value = sigma[k] * (w.lambda_max * dy_sq) + dx_sq / sigma[k] + dot_prod;
        }
        out[k] = std::sqrt(std::max(value, 0.0));
    }
}

double bound_norm_host(const double *AL, const double *AU, int m, size_t off) {  // :332-345
    long double sum = 0.0;
    for (int i = 0; i < m; ++i) {
        const double lo = AL[off + i], hi = AU[off + i];
        const double a = (std::isinf(lo) && lo < 0) ? 0.0 : std::abs(lo);
        const double b = (std::isinf(hi) && hi > 0) ? 0.0 : std::abs(hi);
        const double v = std::max(a, b);
        sum += static_cast<long double>(v) * v;
    }
    return std::sqrt(static_cast<double>(sum));
}
double column_norm_host(const double *X, int n, size_t off) {  // :347-354
    long double sum = 0.0;
    for (int i = 0; i < n; ++i) sum += static_cast<long double>(X[off + i]) * X[off + i];
    return std::sqrt(static_cast<double>(sum));
}

// element (row i, problem k) of a device panel with `rows` rows (the host's copy of pidx)
inline size_t panel_index(const Geo &g, int rows, int i, int k) {
    return (static_cast<size_t>(k / g.Bw) * rows + i) * g.Bw + k % g.Bw;
}
// column-major (ABI) rows x B -> padded device panel
void to_panel(const std::vector<double> &cm, int rows, int B, const Geo &g, double pad, std::vector<double> &out) {
    out.assign(static_cast<size_t>(rows) * g.Bp, pad);
    for (int k = 0; k < B; ++k)
        for (int i = 0; i < rows; ++i) out[panel_index(g, rows, i, k)] = cm[static_cast<size_t>(k) * rows + i];
}

// Chunk width.  Below 64 problems: one chunk.  From 64 up: 8 problems per chunk when the gathered panels are too large for
// an XCD's L2 as a whole but an eighth (a sixteenth, ..) of them is not -- every XCD then works on its own chunks and
// gathers from 1 / nchunk-th of the panel -- else 64 (a wave = one row, scalar CSR loads).  HPRLP_BATCH_CHUNK overrides.
int choose_chunk(int m, int n, int Bp) {
    if (Bp < 64) return Bp;
    if (const char *e = env_get("HPRLP_BATCH_CHUNK")) {
        const int c = std::atoi(e);
        if (c == 8 || c == 16 || c == 32 || c == 64) return c;
    }
    (void)m; (void)n;
    return 64;
}

HPRLP_batched_results make_batched_error(const char *status, int m, int n, int B) {  // :356-368
    HPRLP_batched_results r;
    r.m = m;
    r.n = n;
    r.batch_size = B;
    if (B > 0) {
        r.status = static_cast<char *>(std::calloc(static_cast<size_t>(B) * 64, sizeof(char)));
        if (r.status)
            for (int k = 0; k < B; ++k) std::strncpy(r.status + 64 * k, status, 63);
    }
    return r;
}

}  // namespace
// warm-up (abi.cpp: hprlp_warmup): an attribute query makes the runtime load this translation unit's code object now instead
// of at the first launch of one of its kernels
void warm_batched_tu() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&kb_finalize));
}

}  // namespace hprlp

using namespace hprlp;

extern "C" HPRLP_batched_results solve_batched(const LP_info_cpu *model, int batch_size, const HPRLP_FLOAT *C_in,
                                               const HPRLP_FLOAT *AL_in, const HPRLP_FLOAT *AU_in,
                                               const HPRLP_FLOAT *l_in, const HPRLP_FLOAT *u_in,
                                               const HPRLP_FLOAT *obj_constants, const HPRLP_parameters *param) {
    if (!model || !model->A || batch_size <= 0 || !C_in || !AL_in || !AU_in || !l_in || !u_in)
        return make_batched_error("ERROR", model ? model->m : 0, model ? model->n : 0, std::max(batch_size, 0));
    const int m = model->m, n = model->n, B = batch_size;
    try {
        HPRLP_parameters actual = param ? *param : HPRLP_parameters();
        actual.use_presolve = false;
        const auto setup_start = time_now();

        // shared-A scaling with zero vectors and b/c scaling off (:959-981)
        std::vector<double> zero_m(m, 0.0), zero_n(n, 0.0);
        LP_info_cpu mat{};
        mat.m = m; mat.n = n; mat.A = model->A;
        mat.AL = zero_m.data(); mat.AU = zero_m.data(); mat.c = zero_n.data(); mat.l = zero_n.data(); mat.u = zero_n.data();
        HPRLP_parameters mp = actual;
        mp.use_bc_scaling = false;
        Solver shared;
        shared.verbose = false;
        shared.allow_reorder = false;  // the panels and the returned X / Y / Z are in the caller's numbering
        shared.setup(&mat, &mp);
        shared.scale();
        std::vector<double> rn(m), cn(n);
        shared.row_norm.download(rn.data(), m);
        shared.col_norm.download(cn.data(), n);

        // per-column vector scaling on the host (:792-885)
        std::vector<double> hC(C_in, C_in + static_cast<size_t>(n) * B), hAL(AL_in, AL_in + static_cast<size_t>(m) * B),
            hAU(AU_in, AU_in + static_cast<size_t>(m) * B), hL(l_in, l_in + static_cast<size_t>(n) * B),
            hU(u_in, u_in + static_cast<size_t>(n) * B);
        std::vector<double> b_scale(B, 1.0), c_scale(B, 1.0), norm_b(B), norm_c(B), norm_b_org(B), norm_c_org(B), objc(B);
        for (int k = 0; k < B; ++k) {
            const size_t om = static_cast<size_t>(k) * m, on = static_cast<size_t>(k) * n;
            norm_b_org[k] = 1.0 + bound_norm_host(hAL.data(), hAU.data(), m, om);
            norm_c_org[k] = 1.0 + column_norm_host(hC.data(), n, on);
            for (int i = 0; i < m; ++i) { hAL[om + i] /= rn[i]; hAU[om + i] /= rn[i]; }
            for (int i = 0; i < n; ++i) { hC[on + i] /= cn[i]; hL[on + i] *= cn[i]; hU[on + i] *= cn[i]; }
        }
        if (actual.use_bc_scaling) {
            for (int k = 0; k < B; ++k) {
                const size_t om = static_cast<size_t>(k) * m, on = static_cast<size_t>(k) * n;
                b_scale[k] = 1.0 + bound_norm_host(hAL.data(), hAU.data(), m, om);
                c_scale[k] = 1.0 + column_norm_host(hC.data(), n, on);
                for (int i = 0; i < m; ++i) { hAL[om + i] /= b_scale[k]; hAU[om + i] /= b_scale[k]; }
                for (int i = 0; i < n; ++i) { hC[on + i] /= c_scale[k]; hL[on + i] /= b_scale[k]; hU[on + i] /= b_scale[k]; }
            }
        }
        for (int k = 0; k < B; ++k) {
            const size_t om = static_cast<size_t>(k) * m, on = static_cast<size_t>(k) * n;
            norm_b[k] = bound_norm_host(hAL.data(), hAU.data(), m, om);
            norm_c[k] = column_norm_host(hC.data(), n, on);
            for (int i = 0; i < m; ++i) {
                if (std::isinf(hAL[om + i]) && hAL[om + i] < 0) hAL[om + i] = -kInfReplacement;
                if (std::isinf(hAU[om + i]) && hAU[om + i] > 0) hAU[om + i] = kInfReplacement;
            }
            for (int i = 0; i < n; ++i) {
                if (std::isinf(hL[on + i]) && hL[on + i] < 0) hL[on + i] = -kInfReplacement;
                if (std::isinf(hU[on + i]) && hU[on + i] > 0) hU[on + i] = kInfReplacement;
            }
            objc[k] = obj_constants ? obj_constants[k] : model->obj_constant;
        }

        // lambda_max on the scaled shared matrix (:994-1001)
        const double lambda0 = shared.power_iteration(5000, 1.0e-4, nullptr) * 1.01;
        const double power_time = shared.power_time;

        // workspace (:479-532): row-major padded panels
        BatchWS w;
        w.m = m; w.n = n; w.B = B; w.Bp = padded_batch(B);
        w.shared = &shared;
        w.stream = shared.stream;
        w.lambda_max = lambda0;
        w.geo = make_geo(w.Bp, choose_chunk(m, n, w.Bp));
        const Geo &geo = w.geo;
        w.gx = grid_for(n, geo);
        w.gy = grid_for(m, geo);
        if (geo.Bw >= 8) {
            // groups of kRowsPerWave rows with more than kLongGroup nonzeros go first, longest first; the rest keep their order
            // a wave's group: 64 / Bw lane groups of kRowsPerWave rows each; its length = the longest lane group's entry count
            const int subs = 64 / geo.Bw, gr = kRowsPerWave * subs;
            auto build_order = [gr, subs](const DBuf<int> &rowptr_dev, int rows, DBuf<int> &out) {
                constexpr int kLongGroup = 32;
                std::vector<int> rp(static_cast<size_t>(rows) + 1);
                rowptr_dev.download(rp.data(), rp.size());
                const int ng = (rows + gr - 1) / gr;
                std::vector<int> longg, order;
                order.reserve(static_cast<size_t>(ng));
                auto len = [&](int g) {
                    int longest = 0;
                    for (int sb = 0; sb < subs; ++sb) {  // lane group sb: rows g * gr + i * subs + sb (kb_halfN; kb_half64: subs = 1)
                        int cnt = 0;
                        for (int i = 0; i < kRowsPerWave; ++i) {
                            const int r = g * gr + i * subs + sb;
                            if (r < rows) cnt += rp[r + 1] - rp[r];
                        }
                        longest = std::max(longest, cnt);
                    }
                    return longest;
                };
                for (int g = 0; g < ng; ++g)
                    if (len(g) > kLongGroup) longg.push_back(g);
                if (longg.empty()) return;  // identity: no table
                std::stable_sort(longg.begin(), longg.end(), [&](int x, int y) { return len(x) > len(y); });
                order = longg;
                for (int g = 0; g < ng; ++g)
                    if (len(g) <= kLongGroup) order.push_back(g);
                out.alloc(order.size());
                out.upload(order.data(), order.size());
            };
            build_order(shared.AT.rowptr, n, w.order_x);
            build_order(shared.A.rowptr, m, w.order_y);
        }
        const size_t nB = static_cast<size_t>(n) * w.Bp, mB = static_cast<size_t>(m) * w.Bp;
        {
            std::vector<double> panel;
            to_panel(hC, n, B, geo, 0.0, panel); w.C.alloc(nB); w.C.upload(panel.data(), nB);
            to_panel(hL, n, B, geo, 0.0, panel); w.L.alloc(nB); w.L.upload(panel.data(), nB);
            to_panel(hU, n, B, geo, 0.0, panel); w.U.alloc(nB); w.U.upload(panel.data(), nB);
            to_panel(hAL, m, B, geo, 0.0, panel); w.AL.alloc(mB); w.AL.upload(panel.data(), mB);
            to_panel(hAU, m, B, geo, 0.0, panel); w.AU.alloc(mB); w.AU.upload(panel.data(), mB);
        }
        for (DBuf<double> *p : {&w.X, &w.Xh, &w.Xb, &w.DX, &w.Zb, &w.lastX}) p->alloc_zero(nB);
        for (DBuf<double> *p : {&w.Y, &w.Yb, &w.DY, &w.Yobj, &w.lastY}) p->alloc_zero(mB);
        w.SC.alloc_zero(static_cast<size_t>(B_NSLOT) * w.Bp);
        w.SC_h.alloc(static_cast<size_t>(B_NSLOT) * w.Bp);
        w.partials.alloc_zero(static_cast<size_t>(std::max(w.gx, w.gy)) * 3 * w.Bp);
        w.sigma.alloc(w.Bp); w.active.alloc(w.Bp); w.kx.alloc_zero(w.Bp); w.ky.alloc_zero(w.Bp); w.rflag.alloc_zero(w.Bp);
        w.ctl = BatchCtl{w.sigma.p, w.active.p, w.kx.p, w.ky.p, w.rflag.p};
        std::vector<double> sigma(w.Bp, 1.0);
        std::vector<int> active(w.Bp, 0), flags(w.Bp, 0);
        for (int k = 0; k < B; ++k) {
            if (norm_b[k] > 1.0e-8 && norm_c[k] > 1.0e-8) sigma[k] = norm_b[k] / norm_c[k];
            active[k] = 1;
        }
        w.sigma.upload(sigma.data(), w.Bp);
        w.active.upload(active.data(), w.Bp);
        HIP_CHECK(hipDeviceSynchronize());
        const double setup_time = time_since(setup_start);

        // restart state (:534-556)
        const auto solve_start = time_now();
        const double INF = std::numeric_limits<double>::infinity();
        std::vector<int> rflag(B, 0), inner(B, 0), final_iter(B, actual.max_iter);
        std::vector<unsigned char> first(B, 1);
        std::vector<double> last_gap(B, INF), cur_gap(B, INF), save_gap(B, INF), best_gap(B, INF), best_sigma(sigma.begin(), sigma.begin() + B);
        std::vector<double> r_pobj(B, 0.0), r_dobj(B, 0.0), r_rp(B, 0.0), r_rd(B, 0.0), r_gap(B, 0.0), r_kkt(B, INF), tmp;
        std::vector<std::string> status(B, "CONTINUE");
        const int check_iter = std::max(actual.check_iter, 1);
        const CsrDev &A = shared.A.view, &AT = shared.AT.view;
        bool dxdy_from_movement = false;

        int iter = 0;
        while (true) {  // one pass per event iteration (periodic check or iteration limit), :1017-1084
            const bool periodic = (iter % check_iter) == 0;
            const double elapsed = time_since(solve_start);
            if (periodic) {
                if (iter > 0) weighted_norm(w, dxdy_from_movement, sigma, cur_gap);
                // compute_residuals :578-624
                hipLaunchKernelGGL((kb_resid<0>), dim3(w.gx * geo.nchunk), dim3(256), 0, w.stream, AT.rows, AT.rowptr, AT.col,
                                   AT.val, geo, m, w.Yb.p, static_cast<const double *>(nullptr), w.C.p, w.Zb.p,
                                   shared.col_norm.p, static_cast<const double *>(nullptr), w.partials.p);
                finalize(w, w.gx, {B_RD2});
                hipLaunchKernelGGL((kb_resid<1>), dim3(w.gy * geo.nchunk), dim3(256), 0, w.stream, A.rows, A.rowptr, A.col,
                                   A.val, geo, n, w.Xb.p, static_cast<const double *>(nullptr), w.AL.p, w.AU.p,
                                   shared.row_norm.p, static_cast<const double *>(nullptr), w.partials.p);
                finalize(w, w.gy, {B_RP2});
                if (iter == 0) {
                    hipLaunchKernelGGL(kb_lu, dim3(w.gx * geo.nchunk), dim3(256), 0, w.stream, n, geo, w.Xb.p, w.L.p, w.U.p,
                                       shared.col_norm.p, w.DX.p, w.partials.p);
                    finalize(w, w.gx, {B_LU2});
                }
                fetch(w);
                for (int k = 0; k < B; ++k) {
                    // a frozen member's X_bar/Y_bar/Z_bar no longer change, so its residuals keep the
                    // values of the check that froze it (the fused dot slots only cover active members)
                    if (!active[k]) continue;
                    const double obj_scale = b_scale[k] * c_scale[k];
                    r_pobj[k] = obj_scale * sc(w, B_CX, k) + objc[k];
                    r_dobj[k] = obj_scale * (sc(w, B_YOBJ_Y, k) + sc(w, B_XZ, k)) + objc[k];
                    r_rd[k] = c_scale[k] * std::sqrt(sc(w, B_RD2, k)) / norm_c_org[k];
                    r_rp[k] = b_scale[k] * std::sqrt(sc(w, B_RP2, k)) / norm_b_org[k];
                    if (iter == 0) r_rp[k] = std::max(r_rp[k], b_scale[k] * std::sqrt(sc(w, B_LU2, k)));
                    r_gap[k] = std::abs(r_pobj[k] - r_dobj[k]) / (1.0 + std::abs(r_pobj[k]) + std::abs(r_dobj[k]));
                    r_kkt[k] = std::max(r_rp[k], std::max(r_rd[k], r_gap[k]));
                }
                for (int k = 0; k < B; ++k)
                    if (active[k] && r_kkt[k] <= actual.stop_tol) {
                        status[k] = "OPTIMAL";
                        final_iter[k] = iter;
                        active[k] = 0;
                    }
                w.active.upload(active.data(), w.Bp);
            }
            bool all_done = true;
            for (const std::string &s : status) all_done = all_done && (s != "CONTINUE");
            if (all_done) break;
            if (iter >= actual.max_iter || elapsed >= actual.time_limit) {
                const char *fs = elapsed >= actual.time_limit ? "TIME_LIMIT" : "ITER_LIMIT";
                for (int k = 0; k < B; ++k)
                    if (status[k] == "CONTINUE") {
                        status[k] = fs;
                        final_iter[k] = iter;
                        active[k] = 0;
                    }
                break;
            }
            std::fill(rflag.begin(), rflag.end(), 0);
            if (periodic) {  // check_restart :667-700
                for (int k = 0; k < B; ++k) {
                    if (!active[k]) continue;
                    if (first[k]) {
                        if (iter == check_iter) {
                            first[k] = 0; rflag[k] = 1;
                            best_gap[k] = cur_gap[k]; best_sigma[k] = sigma[k];
                        }
                    } else {
                        if (cur_gap[k] < 0.0) cur_gap[k] = 1.0e-6;
                        if (cur_gap[k] <= 0.2 * last_gap[k]) rflag[k] = 1;
                        if (cur_gap[k] <= 0.6 * last_gap[k] && cur_gap[k] > save_gap[k]) rflag[k] = 2;
                        if (inner[k] >= 0.2 * iter) rflag[k] = 3;
                        if (best_gap[k] > cur_gap[k]) { best_gap[k] = cur_gap[k]; best_sigma[k] = sigma[k]; }
                        save_gap[k] = cur_gap[k];
                    }
                }
            }
            bool restarted = false;
            for (int k = 0; k < B; ++k) restarted = restarted || rflag[k] > 0;
            if (restarted) {
                // update_sigma :702-745 (movement norms for every problem, formula for the flagged ones)
                hipLaunchKernelGGL(kb_movement, dim3(std::max(w.gx, w.gy) * geo.nchunk), dim3(256), 0, w.stream, n, m, geo,
                                   w.Xb.p, w.lastX.p, w.DX.p, w.Yb.p, w.lastY.p, w.DY.p, w.partials.p);
                finalize(w, std::max(w.gx, w.gy), {B_MOVE_X2, B_MOVE_Y2});
                fetch(w);
                dxdy_from_movement = true;
                const double sqrt_lambda = std::sqrt(w.lambda_max);
                for (int k = 0; k < B; ++k) {
                    if (!active[k] || rflag[k] < 1) continue;
                    const double pm = std::sqrt(sc(w, B_MOVE_X2, k)), dm = std::sqrt(sc(w, B_MOVE_Y2, k));
                    if (pm > 1.0e-16 && dm > 1.0e-16 && pm < 1.0e12 && dm < 1.0e12) {
                        const double ratio = (pm / dm) / sqrt_lambda;
                        const double fact = std::exp(-0.05 * (cur_gap[k] / best_gap[k]));
                        const double temp1 = std::max(std::min(r_rd[k], r_rp[k]), std::min(r_gap[k], cur_gap[k]));
                        const double sigma_cand = std::exp(fact * std::log(ratio) + (1.0 - fact) * std::log(best_sigma[k]));
                        const double ratio_infeas = r_rd[k] / r_rp[k];
                        double kappa = 1.0;
                        if (temp1 > 9.0e-10) kappa = 1.0;
                        else if (temp1 > 5.0e-10) kappa = std::max(std::min(std::sqrt(ratio_infeas), 100.0), 1.0e-2);
                        else kappa = std::max(std::min(ratio_infeas, 100.0), 1.0e-2);
                        sigma[k] = kappa * sigma_cand;
                    } else {
                        sigma[k] = 1.0;
                    }
                }
                w.sigma.upload(sigma.data(), w.Bp);
                // do_restart :747-769
                for (int k = 0; k < w.Bp; ++k) flags[k] = (k < B && rflag[k] > 0) ? 1 : 0;
                w.rflag.upload(flags.data(), w.Bp);
                hipLaunchKernelGGL(kb_restart, dim3(std::max(w.gx, w.gy) * geo.nchunk), dim3(256), 0, w.stream, n, m, geo,
                                   w.X.p, w.lastX.p, w.Xb.p, w.Y.p, w.lastY.p, w.Yb.p, w.ctl);
                for (int k = 0; k < B; ++k)
                    if (active[k] && rflag[k] > 0) {
                        inner[k] = 0;
                        save_gap[k] = INF;
                    }
            }
            // iterations iter .. next-1; check variant where the reference's to_check holds (:1067-1068)
            int next = iter + 1;
            while (next % check_iter != 0 && next < actual.max_iter) ++next;
            int it = iter;
            while (it < next) {
                const bool first_after_restart = (it == iter) && restarted;
                int run = 0;  // normal iterations before the next check-variant one
                while (it + run < next && !(((it + run + 1) % check_iter) == 0 || ((it + run + 1) % log_step(it + run + 1)) == 0 ||
                                            (first_after_restart && run == 0)))
                    ++run;
                run_normal(w, run);
                it += run;
                if (it < next) {
                    launch_half_pair(w, true);
                    dxdy_from_movement = false;
                    ++it;
                    if (first_after_restart) {
                        weighted_norm(w, false, sigma, tmp);
                        for (int k = 0; k < B; ++k)
                            if (rflag[k] > 0) last_gap[k] = tmp[k];
                    }
                }
            }
            for (int k = 0; k < B; ++k)
                if (active[k]) inner[k] += next - iter;
            iter = next;
        }
        const double solve_time = time_since(solve_start);

        // collect_results :887-935
        std::vector<double> hX(nB), hY(mB), hZ(nB);
        HIP_CHECK(hipStreamSynchronize(w.stream));
        w.Xb.download(hX.data(), nB);
        w.Yb.download(hY.data(), mB);
        w.Zb.download(hZ.data(), nB);
        HPRLP_batched_results out;
        out.m = m; out.n = n; out.batch_size = B;
        out.x = static_cast<double *>(std::malloc(sizeof(double) * static_cast<size_t>(n) * B));
        out.y = static_cast<double *>(std::malloc(sizeof(double) * static_cast<size_t>(m) * B));
        out.z = static_cast<double *>(std::malloc(sizeof(double) * static_cast<size_t>(n) * B));
        out.primal_obj = static_cast<double *>(std::malloc(sizeof(double) * B));
        out.residuals = static_cast<double *>(std::malloc(sizeof(double) * B));
        out.gap = static_cast<double *>(std::malloc(sizeof(double) * B));
        out.iter = static_cast<int *>(std::malloc(sizeof(int) * B));
        out.status = static_cast<char *>(std::calloc(static_cast<size_t>(B) * 64, sizeof(char)));
        if (!out.x || !out.y || !out.z || !out.primal_obj || !out.residuals || !out.gap || !out.iter || !out.status) {
            free_batched_results(&out);
            throw std::runtime_error("host allocation of the batched results failed");
        }
        for (int k = 0; k < B; ++k) {
            for (int i = 0; i < n; ++i) {
                const size_t src = panel_index(geo, n, i, k), dst = static_cast<size_t>(k) * n + i;
                out.x[dst] = (hX[src] / cn[i]) * b_scale[k];
                out.z[dst] = (hZ[src] * cn[i]) * c_scale[k];
            }
            for (int i = 0; i < m; ++i) {
                const size_t src = panel_index(geo, m, i, k), dst = static_cast<size_t>(k) * m + i;
                out.y[dst] = (hY[src] / rn[i]) * c_scale[k];
            }
            out.primal_obj[k] = r_pobj[k];
            out.residuals[k] = r_kkt[k];
            out.gap[k] = r_gap[k];
            out.iter[k] = final_iter[k];
            std::strncpy(out.status + 64 * k, status[k].c_str(), 63);
        }
        out.setup_time = setup_time;
        out.solve_time = solve_time;
        out.power_time = power_time;
        out.time = setup_time + solve_time;
        return out;
    } catch (const std::exception &e) {
        set_last_error(e.what());
        std::cerr << "[error] solve_batched failed: " << e.what() << std::endl;
        return make_batched_error("ERROR", m, n, B);
    }
}

extern "C" void free_batched_results(HPRLP_batched_results *results) {  // :1094-1105
    if (!results) return;
    std::free(results->x); std::free(results->y); std::free(results->z);
    std::free(results->primal_obj); std::free(results->residuals); std::free(results->gap);
    std::free(results->iter); std::free(results->status);
    *results = HPRLP_batched_results();
}
