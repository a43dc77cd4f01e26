// tiled_build.hip -- the column-tiled matrix copy (tiled.h) built ON THE DEVICE from the device CSR arrays
// (the set-up no longer waits for ~0.5 s of host work per 2e8-nonzero matrix, and the index
// arrays of a device-built A^T never travel to the host).  The result is array-for-array identical to the host
// builder's (tiled.cpp, kept as the reference and for matrices the device path declines): same packing rule --
// entries of a (super-block, tile) pair in (row, CSR order), chunks of 4 that no row segment straddles, padding
// that repeats the previous row, tiles cut into steps of <= 2048 entries, sparse tiles and segments longer than 4
// in a remainder list in (row, CSR order) -- tests/test_gpu_setup.py compares the two.
//
// Pipeline (all on `s`): key = (super-block, tile, local row) per entry; STABLE radix sort of (key, entry index)
// [hipCUB]; runs of equal (super-block, tile) by head flags + scan; one thread per run walks its entries once to
// size the padded list (or to flag a sparse run / long segments for the remainder) and, after an exclusive scan
// of the sizes, a second time to write tidx / tperm / the steps; remainder entries are the flagged entries in
// original order [hipCUB select]; the per-super-block step tables (1221 entries on config 5) are finished on the host.
#include <algorithm>
#include <functional>
#include <cstdlib>
#include <hipcub/hipcub.hpp>

#include <vector>

#include "common.h"
#include "env.h"
#include "kernels.h"
#include "tiled.h"

namespace hprlp {

namespace {

constexpr int K = kTileChunk;  // (the tile width T is a property of the copy: kTileCols or kTileColsNarrow)
constexpr int kRowBits = kTileRowBits;  // the key and the entry codes pack the local row in 13 bits whatever the height R (<= 8192)

__device__ __forceinline__ int row_of_entry(const int *__restrict__ rowptr, int rows, int k) {
    int lo = 0, hi = rows;  // rowptr[lo] <= k < rowptr[hi]
    while (hi - lo > 1) {
        const int mid = lo + ((hi - lo) >> 1);
        if (rowptr[mid] <= k) lo = mid;
        else hi = mid;
    }
    return lo;
}

__global__ void __launch_bounds__(kThreads) k_make_keys(long nnz, int rows, int tile_bits, int R, int T, const int *__restrict__ rowptr,
                                                       const int *__restrict__ col, unsigned long long *__restrict__ key,
                                                       int *__restrict__ idx) {
    const long k = static_cast<long>(blockIdx.x) * kThreads + threadIdx.x;
    if (k >= nnz) return;
    const int r = row_of_entry(rowptr, rows, static_cast<int>(k));
    const unsigned long long sb = static_cast<unsigned long long>(r / R), rl = static_cast<unsigned long long>(r % R);
    const unsigned long long tl = static_cast<unsigned long long>(col[k] / T);
    key[k] = (sb << (tile_bits + kRowBits)) | (tl << kRowBits) | rl;
    idx[k] = static_cast<int>(k);
}

__global__ void __launch_bounds__(kThreads) k_run_heads(long nnz, const unsigned long long *__restrict__ skey, int *__restrict__ head) {
    const long p = static_cast<long>(blockIdx.x) * kThreads + threadIdx.x;
    if (p >= nnz) return;
    head[p] = (p == 0 || (skey[p] >> kRowBits) != (skey[p - 1] >> kRowBits)) ? 1 : 0;
}

__global__ void __launch_bounds__(kThreads) k_run_starts(long nnz, const int *__restrict__ head, const int *__restrict__ run_incl,
                                                        int *__restrict__ run_start) {
    const long p = static_cast<long>(blockIdx.x) * kThreads + threadIdx.x;
    if (p >= nnz) return;
    if (head[p]) run_start[run_incl[p] - 1] = static_cast<int>(p);
}

// One walk of a run = the host builder's loop over one bucket (tiled.cpp).  PASS 1 sizes the padded list and flags
// what goes to the remainder; PASS 2 writes the packed entries.
//
// A lane walks its run entry by entry, and every decision needs the entry's key: read from memory where it is needed that is one
// trip per entry per lane (60 000 runs of 3 300 entries on config 5: 18 ms for pass 2, 3 ms for pass 1).  Instead the wave keeps
// a window of kWalkWin entries per lane in LDS -- keys, and in pass 2 the entries' CSR positions and columns -- refilled for all
// lanes at once (kWalkWin independent loads per lane in flight) whenever some lane's window runs short.
constexpr int kWalkThreads = 64, kWalkWin = 32;
struct WalkWindow {
    unsigned long long *key;  // [kWalkWin][kWalkThreads], this lane's column
    int *perm, *col;          // pass 2 only
};

// PASS 1: count the pieces per layer (tiled.h: kTileLayers) -> *nl_out layers kept, counts[q] their piece counts, returns the
//         padded length of the run's list, *steps_out its steps, *dense / *pad its entries and padding.
// PASS 3: flag what goes to the remainder (pieces beyond the kept layers), given nl.
// PASS 2: write the packed entries, given nl and counts.
template <int PASS>
__device__ __forceinline__ int walk_run(int begin, int end, const unsigned long long *__restrict__ skey,
                                        const int *__restrict__ sperm, const int *__restrict__ col, int col0,
                                        char *__restrict__ flag_sorted, uint32_t *__restrict__ tidx, int *__restrict__ tperm,
                                        int out0, int *dense, int *pad, const WalkWindow &w, int4 *counts, int nl, int *nl_out, int *steps_out) {
    constexpr int W = kWalkWin, NTH = kWalkThreads, NL = kTileLayers;
    constexpr int RM = (1 << kRowBits) - 1;
    int cnt[NL][5];   // PASS 1: pieces of length 1..4 per layer; PASS 2: rank of the next piece of each length and layer
#pragma unroll
    for (int q = 0; q < NL; ++q)
#pragma unroll
        for (int l = 0; l < 5; ++l) cnt[q][l] = 0;
    PackLayout lay[NL];
    int layer_at[NL + 1];
    layer_at[0] = out0;
    if (PASS == 2) {
#pragma unroll
        for (int q = 0; q < NL; ++q) {
            if (q < nl) lay[q].set(counts[q].x, counts[q].y, counts[q].z, counts[q].w);
            layer_at[q + 1] = layer_at[q] + (q < nl ? lay[q].entries() : 0);
        }
    }
    int i = begin;
    int wb = begin - W;  // window base: the window holds entries [wb, wb + W) of this lane's run
    auto key_at = [&](int idx) { return idx < wb + W ? w.key[(idx - wb) * NTH] : skey[idx]; };  // (beyond the window: a long segment's scan)
    while (i < end) {
        if (__any(i + K + 2 > wb + W)) {  // (uniform over the lanes still walking)
            wb = i;
            unsigned long long kr[W];
            int pr[W], cr[W];
#pragma unroll
            for (int e = 0; e < W; ++e) {
                const int idx = min(i + e, end - 1);
                kr[e] = skey[idx];
                if (PASS == 2) pr[e] = sperm[idx];
            }
            if (PASS == 2) {
#pragma unroll
                for (int e = 0; e < W; ++e) cr[e] = col[pr[e]];
            }
#pragma unroll
            for (int e = 0; e < W; ++e) {
                w.key[e * NTH] = kr[e];
                if (PASS == 2) {
                    w.perm[e * NTH] = pr[e];
                    w.col[e * NTH] = cr[e];
                }
            }
        }
        const unsigned long long rkey = key_at(i);
        int j = i + 1;
        while (j < end && key_at(j) == rkey) ++j;
        const int len = j - i, row = static_cast<int>(rkey & RM);
        if (PASS == 1) {
            for (int q = 0; q < NL && q * K < len; ++q) {
                const int pl = min(K, len - q * K);
#pragma unroll
                for (int qq = 0; qq < NL; ++qq)   // (static indexing: the counters stay in registers)
#pragma unroll
                    for (int l = 1; l <= K; ++l)
                        if (qq == q && l == pl) ++cnt[qq][l];
            }
        } else if (PASS == 3) {
            for (int e = i + nl * K; e < j; ++e) flag_sorted[e] = 1;   // pieces beyond the kept layers
        } else {
            for (int q = 0; q < nl && q * K < len; ++q) {
                const int pl = min(K, len - q * K), p0 = i + q * K;
                int rank = 0, at = 0, np = 0;
#pragma unroll
                for (int qq = 0; qq < NL; ++qq)
#pragma unroll
                    for (int l = 1; l <= K; ++l)
                        if (qq == q && l == pl) {
                            rank = cnt[qq][l]++;
                            at = layer_at[qq] + lay[qq].pos(l, rank);
                            np = lay[qq].pads_after(l, rank);
                        }
                for (int e = 0; e < pl; ++e) {
                    const int idx = p0 + e;
                    int pm, cc;
                    if (idx < wb + W) {
                        pm = w.perm[(idx - wb) * NTH];
                        cc = w.col[(idx - wb) * NTH];
                    } else {  // (a segment longer than the window's lookahead)
                        pm = sperm[idx];
                        cc = col[pm];
                    }
                    tperm[at + e] = pm;
                    tidx[at + e] = (static_cast<uint32_t>(cc - col0) << kRowBits) | static_cast<uint32_t>(row);
                }
                for (int e = 0; e < np; ++e) {
                    tidx[at + pl + e] = static_cast<uint32_t>(row);
                    tperm[at + pl + e] = -1;
                }
            }
        }
        i = j;
    }
    if (PASS == 1) {
        auto layer_entries = [&](int q) { return cnt[q][1] + 2 * cnt[q][2] + 3 * cnt[q][3] + 4 * cnt[q][4]; };
        int keep = 1;
#pragma unroll
        for (int q = 1; q < NL; ++q)
            if (keep == q && layer_entries(q) >= kTileLayerMin) keep = q + 1;
        int total = 0, nst = 0;
#pragma unroll
        for (int q = 0; q < NL; ++q) {
            counts[q] = make_int4(cnt[q][1], cnt[q][2], cnt[q][3], cnt[q][4]);
            if (q < keep) {
                lay[q].set(cnt[q][1], cnt[q][2], cnt[q][3], cnt[q][4]);
                const int le = lay[q].entries();
                *dense += layer_entries(q);
                *pad += le - layer_entries(q);
                total += le;
                nst += (le + kTileStepCap - 1) / kTileStepCap;
            }
        }
        *nl_out = keep;
        *steps_out = nst;
        return total;
    }
    return 0;
}

__global__ void __launch_bounds__(kWalkThreads) k_run_pass1(int nruns, bool all_rem, const int *__restrict__ run_start,
                                                           const unsigned long long *__restrict__ skey, char *__restrict__ flag_sorted,
                                                           int *__restrict__ padded_len, int *__restrict__ nsteps,
                                                           unsigned long long *__restrict__ totals, int4 *__restrict__ seg_counts,
                                                           int *__restrict__ run_layers) {
    __shared__ unsigned long long wkey[kWalkWin * kWalkThreads];
    const int r = blockIdx.x * kWalkThreads + threadIdx.x;
    if (r >= nruns) return;
    const int begin = run_start[r], end = run_start[r + 1];
    if (all_rem || end - begin < kTileDenseMin) {  // (all_rem: the all-remainder form stages no tile at all, tiled.h)
        for (int q = begin; q < end; ++q) flag_sorted[q] = 1;
        padded_len[r] = 0;
        nsteps[r] = 0;
        run_layers[r] = 0;
        return;
    }
    int dense = 0, pad = 0, nl = 1, nst = 0;
    const WalkWindow w{wkey + threadIdx.x, nullptr, nullptr};
    int4 c4[kTileLayers];
    const int len = walk_run<1>(begin, end, skey, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0, &dense, &pad, w, c4, 0, &nl, &nst);
    for (int q = 0; q < kTileLayers; ++q) seg_counts[static_cast<size_t>(r) * kTileLayers + q] = c4[q];
    padded_len[r] = len;
    nsteps[r] = nst;
    run_layers[r] = nl;
    atomicAdd(&totals[0], static_cast<unsigned long long>(dense));
    atomicAdd(&totals[1], static_cast<unsigned long long>(pad));
}

// the remainder flags of the staged runs: what their kept layers do not hold (a walk over the keys alone)
__global__ void __launch_bounds__(kWalkThreads) k_run_flags(int nruns, const int *__restrict__ run_start, const unsigned long long *__restrict__ skey,
                                                           const int *__restrict__ run_layers, char *__restrict__ flag_sorted) {
    __shared__ unsigned long long wkey[kWalkWin * kWalkThreads];
    const int r = blockIdx.x * kWalkThreads + threadIdx.x;
    if (r >= nruns) return;
    const int nl = run_layers[r];
    if (nl == 0) return;  // (flagged whole by pass 1)
    const WalkWindow w{wkey + threadIdx.x, nullptr, nullptr};
    walk_run<3>(run_start[r], run_start[r + 1], skey, nullptr, nullptr, 0, flag_sorted, nullptr, nullptr, 0, nullptr, nullptr, w, nullptr, nl, nullptr, nullptr);
}

// first run whose super-block is >= sb, for sb = 0..nsb  (runs are sorted by (super-block, tile))
__global__ void __launch_bounds__(kThreads) k_first_run_of_sb(int nsb, int nruns, int tile_bits, const int *__restrict__ run_start,
                                                             const unsigned long long *__restrict__ skey, int *__restrict__ first_run) {
    const int sb = blockIdx.x * kThreads + threadIdx.x;
    if (sb > nsb) return;
    int lo = 0, hi = nruns;
    while (lo < hi) {
        const int mid = lo + ((hi - lo) >> 1);
        const int rsb = static_cast<int>(skey[run_start[mid]] >> (tile_bits + kRowBits));
        if (rsb < sb) lo = mid + 1;
        else hi = mid;
    }
    first_run[sb] = lo;
}

__global__ void __launch_bounds__(kWalkThreads) k_run_pass2(int nruns, int tile_bits, int T, const int *__restrict__ run_start,
                                                       const unsigned long long *__restrict__ skey, const int *__restrict__ sperm,
                                                       const int *__restrict__ padded_len, const int *__restrict__ run_off,
                                                       const int *__restrict__ run_step_off, const int *__restrict__ first_run,
                                                       const int *__restrict__ sb_ptr, const int *__restrict__ col,
                                                       uint32_t *__restrict__ tidx, int *__restrict__ tperm, TileStep *__restrict__ steps,
                                                       const int4 *__restrict__ seg_counts, const int *__restrict__ run_layers) {
    __shared__ unsigned long long wkey[kWalkWin * kWalkThreads];
    __shared__ int wperm[kWalkWin * kWalkThreads], wcol[kWalkWin * kWalkThreads];
    const int r = blockIdx.x * kWalkThreads + threadIdx.x;
    if (r >= nruns) return;
    const int len = padded_len[r];
    if (len == 0) return;
    const int begin = run_start[r], end = run_start[r + 1], out0 = run_off[r];
    const unsigned long long k0 = skey[begin];
    const int tl = static_cast<int>((k0 >> kRowBits) & ((1ULL << tile_bits) - 1));
    const WalkWindow w{wkey + threadIdx.x, wperm + threadIdx.x, wcol + threadIdx.x};
    const int nl = run_layers[r];
    int4 c4[kTileLayers];
    for (int q = 0; q < kTileLayers; ++q) c4[q] = seg_counts[static_cast<size_t>(r) * kTileLayers + q];
    walk_run<2>(begin, end, skey, sperm, col, tl * T, nullptr, tidx, tperm, out0, nullptr, nullptr, w, c4, nl, nullptr, nullptr);
    const int sb = static_cast<int>(k0 >> (tile_bits + kRowBits));
    int sj = sb_ptr[sb] + (run_step_off[r] - run_step_off[first_run[sb]]);
    int at = out0;
    for (int q = 0; q < nl; ++q) {   // every layer has its own steps
        PackLayout lay;
        lay.set(c4[q].x, c4[q].y, c4[q].z, c4[q].w);
        const int le = lay.entries();
        for (int p = 0; p < le; p += kTileStepCap) steps[sj++] = TileStep{tl * T, at + p, at + min(le, p + kTileStepCap), 0};
        at += le;
    }
}

__global__ void __launch_bounds__(kThreads) k_gather_int(int n, const int *__restrict__ idx, const int *__restrict__ src, int *__restrict__ dst) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

__global__ void __launch_bounds__(kThreads) k_scatter_flags(long nnz, const char *__restrict__ flag_sorted, const int *__restrict__ sperm,
                                                           int *__restrict__ flag_orig) {
    const long p = static_cast<long>(blockIdx.x) * kThreads + threadIdx.x;
    if (p >= nnz) return;
    flag_orig[sperm[p]] = flag_sorted[p];
}

__global__ void __launch_bounds__(kThreads) k_rem_before(int nsb, int rows, int R, const int *__restrict__ rowptr, const int *__restrict__ rem_prefix,
                                                        int *__restrict__ rem_before) {
    const int sb = blockIdx.x * kThreads + threadIdx.x;
    if (sb > nsb) return;
    const long r = static_cast<long>(sb) * R;
    rem_before[sb] = rem_prefix[rowptr[r < rows ? r : rows]];
}

__global__ void __launch_bounds__(kThreads) k_fill_remainder(int n_rem, int rows, int R, const int *__restrict__ rem_k, const int *__restrict__ rowptr,
                                                            const int *__restrict__ col, int *__restrict__ rperm, int *__restrict__ rcol,
                                                            uint16_t *__restrict__ rrow) {
    const int e = blockIdx.x * kThreads + threadIdx.x;
    if (e >= n_rem) return;
    const int k = rem_k[e];
    rperm[e] = k;
    rcol[e] = col[k];
    rrow[e] = static_cast<uint16_t>(row_of_entry(rowptr, rows, k) % R);
}

__global__ void __launch_bounds__(kThreads) k_rem_steps(int nsb, int rem_cap, const int *__restrict__ sb_mid, const int *__restrict__ rem_before,
                                                       TileStep *__restrict__ steps) {
    const int sb = blockIdx.x * kThreads + threadIdx.x;
    if (sb >= nsb) return;
    const int b = rem_before[sb], e = rem_before[sb + 1];
    for (int p = b, j = 0; p < e; p += rem_cap, ++j) steps[sb_mid[sb] + j] = TileStep{0, p, min(e, p + rem_cap), 0};
}

inline unsigned grid_for(long n) { return static_cast<unsigned>((n + kThreads - 1) / kThreads); }

__global__ void __launch_bounds__(kThreads) k_pack_codes(long nchunk, const uint32_t *__restrict__ code, uint32_t *__restrict__ packed) {
    const long c = static_cast<long>(blockIdx.x) * kThreads + threadIdx.x;
    if (c >= nchunk) return;
    const uint32_t e0 = code[4 * c], e1 = code[4 * c + 1], e2 = code[4 * c + 2], e3 = code[4 * c + 3];
    packed[3 * c] = e0 | (e1 << 24);
    packed[3 * c + 1] = (e1 >> 8) | (e2 << 16);
    packed[3 * c + 2] = (e2 >> 16) | (e3 << 8);
}

}  // namespace

void DeviceTiled::pack_indices(hipStream_t s) {
    const long nchunk = n_tile / K;
    tidx3.alloc_zero(static_cast<size_t>(nchunk) * 3 + 8);
    if (nchunk > 0) hipLaunchKernelGGL(k_pack_codes, dim3(grid_for(nchunk)), dim3(kThreads), 0, s, nchunk, tidx.p, tidx3.p);
    HIP_CHECK(hipStreamSynchronize(s));
    tidx.release();
}

// ------------------------------------------------------------------------------------------------
// Launch schedule.  The kernel runs min(per, resident) persistent workgroups per XCD; workgroup `slot` of an XCD
// takes that XCD's super-blocks slot, slot + slots, ...: the workgroups of one XCD work on `slots` consecutive
// super-blocks at any time (a cohort).  Neighbouring super-blocks read almost the same vector tiles, but a sweep
// that starts at each window's own first tile reads them 4 steps apart, and the matrix stream has flushed the
// 4 MiB L2 by then.  Rotation: every super-block starts its sweep at the first tile whose index is a multiple of
// rot_period (wrapping around to its window's first tiles at the end), so that the workgroups of a cohort read
// the SAME tile in the same step and one of them pays the miss.  rot_period = mean window width in tiles.
// The rotation is a fixed property of the matrix: results are reproducible, the per-row summation order is
// "tiles from the rotation point upwards, then the tiles below it, then the remainder entries".
// ------------------------------------------------------------------------------------------------
namespace {

__global__ void __launch_bounds__(kThreads) k_window_widths(int nsb, int T, const int *__restrict__ sb_ptr, const int *__restrict__ sb_mid,
                                                           const TileStep *__restrict__ steps, unsigned long long *__restrict__ out) {
    const int sb = blockIdx.x * kThreads + threadIdx.x;
    if (sb >= nsb) return;
    const int s0 = sb_ptr[sb], smid = sb_mid[sb];
    if (s0 >= smid) return;
    const int w = (steps[smid - 1].col0 - steps[s0].col0) / T + 1;
    atomicAdd(out, static_cast<unsigned long long>(w));  // integer sums: order-independent
    atomicAdd(out + 1, 1ull);
    int rep = 0;  // tiles with more than one step (dense tiles: more than kTileStepCap entries)
    for (int s = s0 + 1; s < smid; ++s) rep += steps[s].col0 == steps[s - 1].col0;
    if (rep) atomicAdd(out + 2, static_cast<unsigned long long>(rep));
}

__global__ void __launch_bounds__(kThreads) k_rotation(int nsb, int period, int T, const int *__restrict__ sb_ptr, const int *__restrict__ sb_mid,
                                                      TileStep *__restrict__ steps) {
    const int sb = blockIdx.x * kThreads + threadIdx.x;
    if (sb >= nsb) return;
    const int s0 = sb_ptr[sb], smid = sb_mid[sb];
    if (s0 >= smid) return;
    const int nst = smid - s0;
    const int w0 = steps[s0].col0 / T;
    const long target = static_cast<long>((w0 + period - 1) / period) * period * T;
    int lo = 0, hi = nst;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (steps[s0 + mid].col0 < target) lo = mid + 1; else hi = mid;
    }
    steps[s0].rot = lo < nst ? lo : 0;
}

}  // namespace

void DeviceTiled::finish_schedule(hipStream_t s) {
    const int nsb = view.nsb;
    int dev = 0, cus = 256;
    HIP_CHECK(hipGetDevice(&dev));
    HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int resident = std::max(1, cus / 8) * kTileResidentPerCu;
    view.per = (nsb + 7) / 8;
    view.grid = 8 * std::min(view.per, resident);
    // Fewer super-blocks than resident workgroup slots (row shards of a multi-GPU run, mid-size LPs): one workgroup per
    // super-block leaves CUs idle and every workgroup with a full-length sweep.  The piece form cuts the tile steps of
    // all super-blocks, laid end to end, into as many equal pieces as there are slots (tiled.h).  HPRLP_TILE_PIECES
    // forces a piece count (tests), 0 disables.
    view.n_pieces = 0;
    int want = (nsb > 0 && nsb <= resident * 8) ? resident * 8 : 0;
    // a lowered super-block height (tiled.h) was chosen so that the slots get whole super-blocks: pieces only below one per CU
    if (view.R < kTileRows && nsb >= cus) want = 0;
    if (view.rem_cap != kTileRemCap) want = 0;  // (k_tiled_part adds remainder steps of kTileRemCap entries; the all-remainder form has its own fused kernel)
    if (const char *e = env_get("HPRLP_TILE_PIECES")) want = std::max(0, std::atoi(e));
    if (want > 0 && nsb > 0) {
        std::vector<int> h_ptr(static_cast<size_t>(nsb) + 1), h_mid(static_cast<size_t>(nsb));
        sb_ptr.download(h_ptr.data(), h_ptr.size());
        sb_mid.download(h_mid.data(), h_mid.size());
        // positions: a super-block's tile steps, then its remainder steps (which stay with its last segment but count as
        // work, so the pieces that carry a remainder get fewer tile steps)
        long total = 0;
        for (int sb = 0; sb < nsb; ++sb) total += h_ptr[sb + 1] - h_ptr[sb];
        const int np = static_cast<int>(std::max<long>(1, std::min<long>(want, std::max<long>(total, 1))));
        std::vector<int4> h_segs;
        std::vector<int> h_piece(static_cast<size_t>(np) + 1, 0), h_slot(static_cast<size_t>(nsb) + 1, 0);
        // walk the super-blocks; global step position g; piece of a position = g * np / total
        long g = 0;
        std::vector<std::vector<int4>> by_piece(static_cast<size_t>(np));
        for (int sb = 0; sb < nsb; ++sb) {
            const int nst = h_mid[sb] - h_ptr[sb];
            h_slot[sb] = 0;  // filled below
            if (nst == 0) {  // remainder only (or an empty super-block): one segment, attached to the piece of position g
                const int pc = static_cast<int>(total > 0 ? std::min<long>(np - 1, g * np / total) : 0);
                by_piece[pc].push_back(make_int4(sb, 0, 0, 1));
                g += h_ptr[sb + 1] - h_mid[sb];
                continue;
            }
            int done = 0;
            while (done < nst) {
                const int pc = static_cast<int>(std::min<long>(np - 1, (g + done) * np / total));
                // first position of the next piece: the smallest q with q * np / total >= pc + 1
                const long nxt = pc + 1 >= np ? total : ((static_cast<long>(pc) + 1) * total + np - 1) / np;
                const int cnt = static_cast<int>(std::min<long>(nst - done, std::max<long>(1, nxt - (g + done))));
                by_piece[pc].push_back(make_int4(sb, done, cnt, done + cnt == nst ? 1 : 0));
                done += cnt;
            }
            g += h_ptr[sb + 1] - h_ptr[sb];
        }
        // segments in piece order; a super-block's segments are consecutive in that order (pieces are ranges of the global
        // step sequence), so its slots are a contiguous range in summation order
        for (int pc = 0; pc < np; ++pc) {
            h_piece[pc] = static_cast<int>(h_segs.size());
            for (const int4 &sg : by_piece[pc]) h_segs.push_back(sg);
        }
        h_piece[np] = static_cast<int>(h_segs.size());
        {
            size_t i = 0;
            for (int sb = 0; sb < nsb; ++sb) {
                h_slot[sb] = static_cast<int>(i);
                while (i < h_segs.size() && h_segs[i].x == sb) ++i;
            }
            h_slot[nsb] = static_cast<int>(i);
            if (i != h_segs.size()) throw std::runtime_error("piece schedule: segments out of super-block order");
        }
        piece_ptr.alloc(h_piece.size()); piece_ptr.upload(h_piece.data(), h_piece.size());
        slot_ptr.alloc(h_slot.size()); slot_ptr.upload(h_slot.data(), h_slot.size());
        segs.alloc(h_segs.size()); segs.upload(h_segs.data(), h_segs.size());
        parts.alloc_zero(h_segs.size() * static_cast<size_t>(view.R));
        view.n_pieces = np;
        view.piece_ptr = piece_ptr.p;
        view.slot_ptr = slot_ptr.p;
        view.segs = segs.p;
        view.parts = parts.p;
        if (const char *e = env_get("HPRLP_TILE_STAMPS"); e && e[0] == '1' && !view.repeats) {
            stamps.alloc_zero(static_cast<size_t>(np) * 16);
            view.stamps = stamps.p;
        }
    }
    if (const char *e = env_get("HPRLP_WG_TIMES"); e && (e[0] == '1' || e[0] == '2') && view.grid > 0) {
        wgtimes.alloc_zero(static_cast<size_t>(view.grid) * 8);
        view.wgtimes = wgtimes.p;
        view.wg_filter = e[0] == '2';
    }
    rot_period = 0;
    if (nsb <= 0 || n_steps <= 0) return;
    {
        DBuf<unsigned long long> acc(3);
        HIP_CHECK(hipMemsetAsync(acc.p, 0, 3 * sizeof(unsigned long long), s));
        hipLaunchKernelGGL(k_window_widths, dim3(grid_for(nsb)), dim3(kThreads), 0, s, nsb, view.T, sb_ptr.p, sb_mid.p, steps.p, acc.p);
        unsigned long long h[3] = {0, 0, 0};
        HIP_CHECK(hipMemcpyAsync(h, acc.p, sizeof(h), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        if (h[1] > 0) rot_period = static_cast<int>((h[0] + h[1] - 1) / h[1]);
        view.repeats = h[2] > 0;
    }
    if (const char *e = env_get("HPRLP_TILE_ROT")) rot_period = std::atoi(e);
    if (rot_period > 0) hipLaunchKernelGGL(k_rotation, dim3(grid_for(nsb)), dim3(kThreads), 0, s, nsb, rot_period, view.T, sb_ptr.p, sb_mid.p, steps.p);
    HIP_CHECK(hipStreamSynchronize(s));
}

bool DeviceTiled::build_on_device(int rows, int cols, long nnz, const int *rowptr, const int *col, int min_rows,
                                  double min_dense_fraction, hipStream_t s, int R, int T, int rem_cap) {
    if (rem_cap != kTileRemCap && rem_cap != kPbRemCap) throw std::runtime_error("tiled build: unsupported remainder step size");
    if (rem_cap == kPbRemCap && R > kPbRowsMax) throw std::runtime_error("tiled build: the all-remainder form takes super-blocks of at most 4096 rows");
    if (rows < min_rows || rows <= 0 || cols <= 0 || nnz <= 0 || nnz >= 2000000000L) return false;
    if (R < 64 || R > kTileRows || R % 64 != 0) throw std::runtime_error("tiled build: unsupported super-block height");
    if (T != kTileCols && T != kTileColsNarrow) throw std::runtime_error("tiled build: unsupported tile width");
    const int nsb = (rows + R - 1) / R;
    const int ntile = (cols + T - 1) / T;
    int tile_bits = 1;
    while ((1 << tile_bits) < ntile) ++tile_bits;
    int sb_bits = 1;
    while ((1 << sb_bits) < nsb) ++sb_bits;
    const int key_bits = sb_bits + tile_bits + kRowBits;

    DBuf<unsigned long long> key_in(static_cast<size_t>(nnz)), skey(static_cast<size_t>(nnz));
    DBuf<int> idx_in(static_cast<size_t>(nnz)), sperm(static_cast<size_t>(nnz));
    hipLaunchKernelGGL(k_make_keys, dim3(grid_for(nnz)), dim3(kThreads), 0, s, nnz, rows, tile_bits, R, T, rowptr, col, key_in.p, idx_in.p);
    size_t tmp_bytes = 0;
    // Sorted on the (super-block, tile) bits only: the sort is stable and the input is in CSR order -- rows ascending -- so inside a
    // (super-block, tile) run the entries come out by row, then CSR position, exactly as a sort on the whole key leaves them; 24
    // instead of 37 key bits on config 5 = three radix passes instead of five (7.8 -> 4.7 ms per matrix).
    HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, key_in.p, skey.p, idx_in.p, sperm.p, static_cast<int>(nnz), kRowBits,
                                                 key_bits, s));
    {
        DBuf<char> tmp(tmp_bytes + 16);
        HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, key_in.p, skey.p, idx_in.p, sperm.p, static_cast<int>(nnz), kRowBits,
                                                     key_bits, s));
        HIP_CHECK(hipStreamSynchronize(s));
    }
    key_in.release();
    idx_in.release();

    // runs of equal (super-block, tile)
    DBuf<int> head(static_cast<size_t>(nnz)), run_incl(static_cast<size_t>(nnz));
    hipLaunchKernelGGL(k_run_heads, dim3(grid_for(nnz)), dim3(kThreads), 0, s, nnz, skey.p, head.p);
    HIP_CHECK(hipcub::DeviceScan::InclusiveSum(nullptr, tmp_bytes, head.p, run_incl.p, static_cast<int>(nnz), s));
    {
        DBuf<char> tmp(tmp_bytes + 16);
        HIP_CHECK(hipcub::DeviceScan::InclusiveSum(tmp.p, tmp_bytes, head.p, run_incl.p, static_cast<int>(nnz), s));
        HIP_CHECK(hipStreamSynchronize(s));
    }
    int nruns = 0;
    HIP_CHECK(hipMemcpy(&nruns, run_incl.p + (nnz - 1), sizeof(int), hipMemcpyDeviceToHost));
    DBuf<int> run_start(static_cast<size_t>(nruns) + 1);
    hipLaunchKernelGGL(k_run_starts, dim3(grid_for(nnz)), dim3(kThreads), 0, s, nnz, head.p, run_incl.p, run_start.p);
    {
        const int end = static_cast<int>(nnz);
        HIP_CHECK(hipMemcpyAsync(run_start.p + nruns, &end, sizeof(int), hipMemcpyHostToDevice, s));
        HIP_CHECK(hipStreamSynchronize(s));
    }
    head.release();
    run_incl.release();

    // pass 1: sizes and remainder flags
    DBuf<char> flag_sorted;
    flag_sorted.alloc_zero(static_cast<size_t>(nnz));
    DBuf<int> padded_len(static_cast<size_t>(nruns) + 1), nsteps(static_cast<size_t>(nruns) + 1);
    DBuf<int4> seg_counts((static_cast<size_t>(nruns) + 1) * kTileLayers);  // per run and layer: its pieces of length 1..4 (tiled.h: PackLayout, kTileLayers)
    DBuf<int> run_layers(static_cast<size_t>(nruns) + 1);
    DBuf<unsigned long long> totals;
    totals.alloc_zero(2);
    HIP_CHECK(hipMemsetAsync(padded_len.p, 0, sizeof(int) * (static_cast<size_t>(nruns) + 1), s));
    HIP_CHECK(hipMemsetAsync(nsteps.p, 0, sizeof(int) * (static_cast<size_t>(nruns) + 1), s));
    hipLaunchKernelGGL(k_run_pass1, dim3((nruns + kWalkThreads - 1) / kWalkThreads), dim3(kWalkThreads), 0, s, nruns, rem_cap == kPbRemCap, run_start.p, skey.p, flag_sorted.p, padded_len.p,
                       nsteps.p, totals.p, seg_counts.p, run_layers.p);
    hipLaunchKernelGGL(k_run_flags, dim3((nruns + kWalkThreads - 1) / kWalkThreads), dim3(kWalkThreads), 0, s, nruns, run_start.p, skey.p, run_layers.p, flag_sorted.p);
    unsigned long long tot[2] = {0, 0};
    HIP_CHECK(hipMemcpyAsync(tot, totals.p, sizeof(tot), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    if (static_cast<double>(tot[0]) < min_dense_fraction * static_cast<double>(nnz) || tot[0] + tot[1] > 2000000000ULL) return false;
    n_tile = static_cast<long>(tot[0] + tot[1]);

    // offsets of the runs' entries and steps (exclusive scans over nruns + 1 elements: the last one is the total)
    DBuf<int> run_off(static_cast<size_t>(nruns) + 1), run_step_off(static_cast<size_t>(nruns) + 1);
    auto exclusive_scan = [&](const int *in, int *out, int count) {
        size_t bytes = 0;
        HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, in, out, count, s));
        DBuf<char> tmp(bytes + 16);
        HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(tmp.p, bytes, in, out, count, s));
        HIP_CHECK(hipStreamSynchronize(s));
    };
    exclusive_scan(padded_len.p, run_off.p, nruns + 1);
    exclusive_scan(nsteps.p, run_step_off.p, nruns + 1);

    // remainder: flags back in original order, prefix sums, per-super-block boundaries
    DBuf<int> flag_orig(static_cast<size_t>(nnz) + 1), rem_prefix(static_cast<size_t>(nnz) + 1);
    HIP_CHECK(hipMemsetAsync(flag_orig.p + nnz, 0, sizeof(int), s));
    hipLaunchKernelGGL(k_scatter_flags, dim3(grid_for(nnz)), dim3(kThreads), 0, s, nnz, flag_sorted.p, sperm.p, flag_orig.p);
    exclusive_scan(flag_orig.p, rem_prefix.p, static_cast<int>(nnz) + 1);
    int n_rem_i = 0;
    HIP_CHECK(hipMemcpy(&n_rem_i, rem_prefix.p + nnz, sizeof(int), hipMemcpyDeviceToHost));
    n_rem = n_rem_i;
    flag_sorted.release();

    // per-super-block tables: finished on the host (nsb + 1 entries)
    DBuf<int> first_run(static_cast<size_t>(nsb) + 1), rem_before(static_cast<size_t>(nsb) + 1);
    hipLaunchKernelGGL(k_first_run_of_sb, dim3(grid_for(nsb + 1)), dim3(kThreads), 0, s, nsb, nruns, tile_bits, run_start.p, skey.p,
                       first_run.p);
    hipLaunchKernelGGL(k_rem_before, dim3(grid_for(nsb + 1)), dim3(kThreads), 0, s, nsb, rows, R, rowptr, rem_prefix.p, rem_before.p);
    HIP_CHECK(hipStreamSynchronize(s));
    DBuf<int> dsteps(static_cast<size_t>(nsb) + 1);  // tile steps before each super-block
    hipLaunchKernelGGL(k_gather_int, dim3(grid_for(nsb + 1)), dim3(kThreads), 0, s, nsb + 1, first_run.p, run_step_off.p, dsteps.p);
    HIP_CHECK(hipStreamSynchronize(s));
    std::vector<int> h_rem(static_cast<size_t>(nsb) + 1), h_dsteps(static_cast<size_t>(nsb) + 1);
    rem_before.download(h_rem.data(), h_rem.size());
    dsteps.download(h_dsteps.data(), h_dsteps.size());
    std::vector<int> h_sb_ptr(static_cast<size_t>(nsb) + 1), h_sb_mid(static_cast<size_t>(nsb));
    int rsteps_before = 0;
    for (int sb = 0; sb <= nsb; ++sb) {
        h_sb_ptr[sb] = h_dsteps[sb] + rsteps_before;
        if (sb < nsb) {
            h_sb_mid[sb] = h_sb_ptr[sb] + (h_dsteps[sb + 1] - h_dsteps[sb]);
            rsteps_before += (h_rem[sb + 1] - h_rem[sb] + rem_cap - 1) / rem_cap;
        }
    }
    const int total_steps = h_sb_ptr[nsb];
    sb_ptr.alloc(h_sb_ptr.size());
    sb_ptr.upload(h_sb_ptr.data(), h_sb_ptr.size());
    sb_mid.alloc(h_sb_mid.size());
    sb_mid.upload(h_sb_mid.data(), h_sb_mid.size());
    steps.alloc(static_cast<size_t>(total_steps));

    // pass 2: the packed entries and the tile steps; remainder arrays and steps
    tidx.alloc_zero(static_cast<size_t>(n_tile) + 8);
    tperm.alloc(static_cast<size_t>(n_tile) + 8);
    tval.alloc_zero(static_cast<size_t>(n_tile) + 8);
    hipLaunchKernelGGL(k_run_pass2, dim3((nruns + kWalkThreads - 1) / kWalkThreads), dim3(kWalkThreads), 0, s, nruns, tile_bits, T, run_start.p, skey.p, sperm.p, padded_len.p,
                       run_off.p, run_step_off.p, first_run.p, sb_ptr.p, col, tidx.p, tperm.p, steps.p, seg_counts.p, run_layers.p);
    rcol.alloc_zero(static_cast<size_t>(n_rem) + 8);
    rperm.alloc(static_cast<size_t>(n_rem) + 8);
    rrow.alloc_zero(static_cast<size_t>(n_rem) + 8);
    if (n_rem > 0) {
        DBuf<int> rem_k(static_cast<size_t>(n_rem)), nsel(1);
        hipcub::CountingInputIterator<int> iota(0);
        size_t bytes = 0;
        HIP_CHECK(hipcub::DeviceSelect::Flagged(nullptr, bytes, iota, flag_orig.p, rem_k.p, nsel.p, static_cast<int>(nnz), s));
        DBuf<char> tmp(bytes + 16);
        HIP_CHECK(hipcub::DeviceSelect::Flagged(tmp.p, bytes, iota, flag_orig.p, rem_k.p, nsel.p, static_cast<int>(nnz), s));
        hipLaunchKernelGGL(k_fill_remainder, dim3(grid_for(n_rem)), dim3(kThreads), 0, s, static_cast<int>(n_rem), rows, R, rem_k.p, rowptr, col,
                           rperm.p, rcol.p, rrow.p);
        HIP_CHECK(hipStreamSynchronize(s));
    }
    hipLaunchKernelGGL(k_rem_steps, dim3(grid_for(nsb)), dim3(kThreads), 0, s, nsb, rem_cap, sb_mid.p, rem_before.p, steps.p);
    HIP_CHECK(hipStreamSynchronize(s));

    dense_entries = static_cast<long>(tot[0]);
    padding = static_cast<long>(tot[1]);
    n_steps = total_steps;
    view = TiledDev();
    view.valid = true;
    view.R = R;
    view.T = T;
    view.rem_cap = rem_cap;
    view.nsb = nsb;
    view.sb_ptr = sb_ptr.p;
    view.sb_mid = sb_mid.p;
    view.steps = steps.p;
    view.tval = tval.p;
    pack_indices(s);
    view.tidx3 = tidx3.p;
    finish_schedule(s);
    return true;
}


// ------------------------------------------------------------------------------------------------
// Propagation-blocking lists of the remainder (tiled.h).  Input: the remainder entries in (super-block, row, CSR)
// order (rcol, rrow, rperm) and the remainder steps (ranges of at most kTileRemCap entries that never cross a
// super-block).  Three stable radix sorts:
//   P order  = (super-block, source group, e): where the pre-pass writes the product of entry e -- the run of one
//              (source group, super-block) pair is contiguous, a super-block's slice of P is what the tiled kernel streams;
//   rq       = per step range of P: its entries in e order (= row, CSR order), as slot-in-step << 16 | local row;
//   f order  = (source group, P position): what one pre-pass workgroup reads, ascending in its writes.
// ------------------------------------------------------------------------------------------------
namespace {

__global__ void __launch_bounds__(kThreads) k_far_step_of(int nsb, const int *__restrict__ sb_ptr, const int *__restrict__ sb_mid,
                                                         TileStep *__restrict__ steps, const uint16_t *__restrict__ rrow,
                                                         int *__restrict__ step_of, int *__restrict__ sb_of, bool mark) {
    // one workgroup per super-block: every remainder entry learns its step and its super-block; a step in which some row
    // holds more than kTileRemRun consecutive entries is marked (col0 = 1, unused by remainder steps otherwise; read by the
    // HPRLP_TIMING statistics only since round 4: kernels.hip, remainder_steps adds rows of any length the same way)
    const int sb = blockIdx.x;
    if (sb >= nsb) return;
    for (int s = sb_mid[sb]; s < sb_ptr[sb + 1]; ++s) {
        const TileStep st = steps[s];
        if (threadIdx.x == 0) steps[s].col0 = 0;
        __syncthreads();
        for (int e = st.e_begin + threadIdx.x; e < st.e_end; e += kThreads) {
            step_of[e] = s;
            sb_of[e] = sb;
            if (mark && e + kTileRemRun < st.e_end && rrow[e] == rrow[e + kTileRemRun]) steps[s].col0 = 1;  // (entries are in row order)
        }
    }
}

__global__ void __launch_bounds__(kThreads) k_far_key_p(int n, int G, const int *__restrict__ sb_of, const int *__restrict__ rcol,
                                                       unsigned long long *__restrict__ key, int *__restrict__ val) {
    const int e = blockIdx.x * kThreads + threadIdx.x;
    if (e >= n) return;
    key[e] = (static_cast<unsigned long long>(sb_of[e]) << 32) | static_cast<unsigned long long>(rcol[e] / G);
    val[e] = e;
}

// p -> key (step of p, entry), value p.  The step ranges are ranges of the P index space as well: a super-block's
// entries occupy the same index range in either order.
__global__ void __launch_bounds__(kThreads) k_far_key_q(int n, const int *__restrict__ step_of, const int *__restrict__ e_of_p,
                                                       unsigned long long *__restrict__ key, int *__restrict__ val) {
    const int p = blockIdx.x * kThreads + threadIdx.x;
    if (p >= n) return;
    key[p] = (static_cast<unsigned long long>(step_of[p]) << 32) | static_cast<unsigned long long>(e_of_p[p]);
    val[p] = p;
}

__global__ void __launch_bounds__(kThreads) k_far_fill_q(int n, const unsigned long long *__restrict__ skey, const int *__restrict__ p_sorted,
                                                        const TileStep *__restrict__ steps, const uint16_t *__restrict__ rrow,
                                                        uint32_t *__restrict__ rq) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const int step = static_cast<int>(skey[i] >> 32), e = static_cast<int>(skey[i] & 0xffffffffu);
    const int slot = p_sorted[i] - steps[step].e_begin;
    rq[i] = (static_cast<uint32_t>(slot) << 16) | static_cast<uint32_t>(rrow[e]);
}

__global__ void __launch_bounds__(kThreads) k_far_key_f(int n, int G, const int *__restrict__ e_of_p, const int *__restrict__ rcol,
                                                       unsigned long long *__restrict__ key, int *__restrict__ val) {
    const int p = blockIdx.x * kThreads + threadIdx.x;
    if (p >= n) return;
    key[p] = (static_cast<unsigned long long>(rcol[e_of_p[p]] / G) << 32) | static_cast<unsigned long long>(p);
    val[p] = e_of_p[p];
}

__global__ void __launch_bounds__(kThreads) k_far_fill_f(int n, int G, const unsigned long long *__restrict__ skey, const int *__restrict__ e_sorted,
                                                        const int *__restrict__ rcol, const int *__restrict__ rperm, int *__restrict__ f_pos,
                                                        uint16_t *__restrict__ f_lcol, int *__restrict__ f_perm) {
    const int f = blockIdx.x * kThreads + threadIdx.x;
    if (f >= n) return;
    const int g = static_cast<int>(skey[f] >> 32), e = e_sorted[f];
    f_pos[f] = static_cast<int>(skey[f] & 0xffffffffu);
    f_lcol[f] = static_cast<uint16_t>(rcol[e] - g * G);
    f_perm[f] = rperm[e];
}

// gptr[g] = first f whose group is >= g
__global__ void __launch_bounds__(kThreads) k_far_gptr(int ngroups, int n, const unsigned long long *__restrict__ skey, int *__restrict__ gptr) {
    const int g = blockIdx.x * kThreads + threadIdx.x;
    if (g > ngroups) return;
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = lo + ((hi - lo) >> 1);
        if (static_cast<int>(skey[mid] >> 32) < g) lo = mid + 1;
        else hi = mid;
    }
    gptr[g] = lo;
}

// run tables of the source side (tiled.h: f_rk / f_rp / f_rptr): entry f starts a run if it is the first of its group or its
// P position does not follow its predecessor's
__global__ void __launch_bounds__(kThreads) k_far_run_flags(int n, const unsigned long long *__restrict__ skey, const int *__restrict__ f_pos,
                                                           int *__restrict__ flag) {
    const int f = blockIdx.x * kThreads + threadIdx.x;
    if (f >= n) return;
    flag[f] = (f == 0 || (skey[f] >> 32) != (skey[f - 1] >> 32) || f_pos[f] != f_pos[f - 1] + 1) ? 1 : 0;
}

__global__ void __launch_bounds__(kThreads) k_far_run_fill(int n, const int *__restrict__ flag, const int *__restrict__ run_excl,
                                                          const int *__restrict__ f_pos, int *__restrict__ rk, int *__restrict__ rp) {
    const int f = blockIdx.x * kThreads + threadIdx.x;
    if (f >= n || !flag[f]) return;
    rk[run_excl[f]] = f;
    rp[run_excl[f]] = f_pos[f];
}

// rptr[g] = runs before group g's first entry (a group's first entry starts a run); rptr[ngroups] = all runs
__global__ void __launch_bounds__(kThreads) k_far_rptr(int ngroups, int n, int nruns, const int *__restrict__ gptr, const int *__restrict__ run_excl,
                                                      int *__restrict__ rptr, int *__restrict__ maxruns) {
    const int g = blockIdx.x * kThreads + threadIdx.x;
    if (g > ngroups) return;
    const int f = gptr[g];
    rptr[g] = f < n ? run_excl[f] : nruns;
    if (g < ngroups) {
        const int f1 = gptr[g + 1];
        const int r1 = f1 < n ? run_excl[f1] : nruns;
        atomicMax(maxruns, r1 - rptr[g]);
    }
}

void sort_pairs(DBuf<unsigned long long> &kin, DBuf<unsigned long long> &kout, DBuf<int> &vin, DBuf<int> &vout, int n, int end_bit, hipStream_t s) {
    size_t bytes = 0;
    HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, kin.p, kout.p, vin.p, vout.p, n, 0, end_bit, s));
    DBuf<char> tmp(bytes + 16);
    HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(tmp.p, bytes, kin.p, kout.p, vin.p, vout.p, n, 0, end_bit, s));
    HIP_CHECK(hipStreamSynchronize(s));
}

int bits_for(long v) {
    int b = 1;
    while ((1L << b) <= v) ++b;
    return b;
}

}  // namespace

void DeviceTiled::build_far(int cols, hipStream_t s, int G) {
    static_assert(kFarGroup <= 65536 && kTileRemCap <= 65536 && kTileRows <= 65536, "16-bit local columns, slots and rows");
    if (G < 64 || G > kFarGroup) throw std::runtime_error("remainder lists: unsupported source-group size");
    view.G = G;
    view.P = nullptr;
    view.rq = nullptr;
    view.n_groups = 0;
    if (!view.valid || n_rem <= 0) {
        rcol.release(); rrow.release(); rperm.release();
        return;
    }
    const int n = static_cast<int>(n_rem), nsb = view.nsb;
    const int ngroups = (cols + G - 1) / G;
    DBuf<int> step_of(static_cast<size_t>(n)), sb_of(static_cast<size_t>(n));
    const char *no2 = env_get("HPRLP_NO_REM2");  // diagnostic: long runs of the remainder added by one lane, as before
    hipLaunchKernelGGL(k_far_step_of, dim3(nsb), dim3(kThreads), 0, s, nsb, sb_ptr.p, sb_mid.p, steps.p, rrow.p, step_of.p, sb_of.p, !(no2 && no2[0] == '1'));
    if (env_get("HPRLP_TIMING")) {
        std::vector<TileStep> hs(static_cast<size_t>(n_steps));
        std::vector<int> hp(static_cast<size_t>(nsb) + 1), hm(static_cast<size_t>(nsb));
        HIP_CHECK(hipStreamSynchronize(s));
        steps.download(hs.data(), hs.size());
        sb_ptr.download(hp.data(), hp.size());
        sb_mid.download(hm.data(), hm.size());
        long rem_steps = 0, marked = 0, most = 0;
        for (int sb = 0; sb < nsb; ++sb) {
            most = std::max<long>(most, hp[sb + 1] - hm[sb]);
            for (int q = hm[sb]; q < hp[sb + 1]; ++q) {
                ++rem_steps;
                marked += hs[q].col0 != 0;
            }
        }
        std::fprintf(stderr, "[timing]   remainder: %ld steps (most in one super-block: %ld), %ld with a long run of one row\n", rem_steps, most, marked);
    }
    DBuf<unsigned long long> kin(static_cast<size_t>(n)), kout(static_cast<size_t>(n));
    DBuf<int> vin(static_cast<size_t>(n)), e_of_p(static_cast<size_t>(n));
    // P order
    hipLaunchKernelGGL(k_far_key_p, dim3(grid_for(n)), dim3(kThreads), 0, s, n, G, sb_of.p, rcol.p, kin.p, vin.p);
    sort_pairs(kin, kout, vin, e_of_p, n, 32 + bits_for(nsb), s);
    sb_of.release();
    // rq: the entries of every step range of P in e order
    DBuf<int> p_sorted(static_cast<size_t>(n));
    hipLaunchKernelGGL(k_far_key_q, dim3(grid_for(n)), dim3(kThreads), 0, s, n, step_of.p, e_of_p.p, kin.p, vin.p);
    sort_pairs(kin, kout, vin, p_sorted, n, 32 + bits_for(n_steps), s);
    rq.alloc_zero(static_cast<size_t>(n) + 8);
    hipLaunchKernelGGL(k_far_fill_q, dim3(grid_for(n)), dim3(kThreads), 0, s, n, kout.p, p_sorted.p, steps.p, rrow.p, rq.p);
    HIP_CHECK(hipStreamSynchronize(s));
    step_of.release();
    // source side
    hipLaunchKernelGGL(k_far_key_f, dim3(grid_for(n)), dim3(kThreads), 0, s, n, G, e_of_p.p, rcol.p, kin.p, vin.p);
    sort_pairs(kin, kout, vin, p_sorted, n, 32 + bits_for(ngroups), s);  // p_sorted now holds e in f order
    f_pos.alloc_zero(static_cast<size_t>(n) + 8);
    f_lcol.alloc_zero(static_cast<size_t>(n) + 8);
    f_perm.alloc(static_cast<size_t>(n) + 8);
    f_val.alloc_zero(static_cast<size_t>(n) + 8);
    f_gptr.alloc(static_cast<size_t>(ngroups) + 1);
    hipLaunchKernelGGL(k_far_fill_f, dim3(grid_for(n)), dim3(kThreads), 0, s, n, G, kout.p, p_sorted.p, rcol.p, rperm.p, f_pos.p, f_lcol.p, f_perm.p);
    hipLaunchKernelGGL(k_far_gptr, dim3(grid_for(ngroups + 1)), dim3(kThreads), 0, s, ngroups, n, kout.p, f_gptr.p);
    P.alloc_zero(static_cast<size_t>(n) + 8);
    HIP_CHECK(hipStreamSynchronize(s));
    {   // pre-pass work list: only where a source group is several times the mean (tiled.h: f_work)
        std::vector<int> gp(static_cast<size_t>(ngroups) + 1);
        f_gptr.download(gp.data(), gp.size());
        const long cap = std::max<long>(kFarWorkMin, kFarWorkOverMean * static_cast<long>(n) / std::max(ngroups, 1));
        int heaviest = 0;
        for (int g = 0; g < ngroups; ++g) heaviest = std::max(heaviest, gp[g + 1] - gp[g]);
        {   // how concentrated the remainder is (DeviceMatrix::build_tiled_copy): the heaviest groups covering 2 MB of the vector
            std::vector<int> sz(static_cast<size_t>(ngroups));
            for (int g = 0; g < ngroups; ++g) sz[g] = gp[g + 1] - gp[g];
            const int k = std::max(1, std::min(ngroups, static_cast<int>((2L << 20) / (8L * G))));
            std::nth_element(sz.begin(), sz.begin() + (k - 1), sz.end(), std::greater<int>());
            long top = 0;
            for (int g = 0; g < k; ++g) top += sz[g];
            rem_top_share = static_cast<double>(top) / static_cast<double>(n);
        }
        view.f_work = nullptr;
        view.n_work = 0;
        f_work.release();
        if (heaviest > cap && !env_get("HPRLP_NO_FAR_WORK")) {
            std::vector<int4> wl;
            wl.reserve(static_cast<size_t>(ngroups) + 64);
            for (int g = 0; g < ngroups; ++g) {
                const long b = gp[g], e = gp[g + 1];
                if (e <= b) continue;
                const long parts = (e - b + cap - 1) / cap, per = (e - b + parts - 1) / parts;
                for (long q = b; q < e; q += per) wl.push_back(make_int4(g, static_cast<int>(q), static_cast<int>(std::min(e, q + per)), 0));
            }
            f_work.alloc(wl.size());
            f_work.upload(wl.data(), wl.size());
            view.f_work = f_work.p;
            view.n_work = static_cast<int>(wl.size());
            if (env_get("HPRLP_TIMING"))
                std::fprintf(stderr, "[timing]   pre-pass work list: %d workgroups for %d source groups (heaviest group %d entries, chunks of at most %ld)\n",
                             view.n_work, ngroups, heaviest, cap);
        }
    }
    view.f_rptr = view.f_rk = view.f_rp = nullptr;
    view.f_maxruns = 0;
    if (view.rem_cap == kPbRemCap) {
        // all-remainder form: run tables, so that the producers need not read f_pos (4 of the 22 bytes per entry they move)
        DBuf<int> flag(static_cast<size_t>(n)), excl(static_cast<size_t>(n));
        hipLaunchKernelGGL(k_far_run_flags, dim3(grid_for(n)), dim3(kThreads), 0, s, n, kout.p, f_pos.p, flag.p);
        size_t bytes = 0;
        HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, flag.p, excl.p, n, s));
        {
            DBuf<char> tmp(bytes + 16);
            HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(tmp.p, bytes, flag.p, excl.p, n, s));
            HIP_CHECK(hipStreamSynchronize(s));
        }
        int last_excl = 0, last_flag = 0;
        HIP_CHECK(hipMemcpy(&last_excl, excl.p + (n - 1), sizeof(int), hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(&last_flag, flag.p + (n - 1), sizeof(int), hipMemcpyDeviceToHost));
        const int nruns = last_excl + last_flag;
        f_rk.alloc(static_cast<size_t>(nruns) + 8);
        f_rp.alloc(static_cast<size_t>(nruns) + 8);
        f_rptr.alloc(static_cast<size_t>(ngroups) + 1);
        DBuf<int> mx;
        mx.alloc_zero(1);
        hipLaunchKernelGGL(k_far_run_fill, dim3(grid_for(n)), dim3(kThreads), 0, s, n, flag.p, excl.p, f_pos.p, f_rk.p, f_rp.p);
        hipLaunchKernelGGL(k_far_rptr, dim3(grid_for(ngroups + 1)), dim3(kThreads), 0, s, ngroups, n, nruns, f_gptr.p, excl.p, f_rptr.p, mx.p);
        int h_mx = 0;
        HIP_CHECK(hipMemcpyAsync(&h_mx, mx.p, sizeof(int), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        // (only where runs are long: with 14 entries per run -- the config-3 recipe x 30 -- neighbouring lanes sit in different
        // runs, the lookups stop being broadcasts and the table costs more than the 4 bytes it saves: measured +6-8 %)
        if (h_mx > 0 && h_mx <= kPbRunTabCap && static_cast<long>(n) >= 48L * nruns) {
            view.f_rptr = f_rptr.p;
            view.f_rk = f_rk.p;
            view.f_rp = f_rp.p;
            view.f_maxruns = h_mx;
        }
        if (env_get("HPRLP_TIMING"))
            std::fprintf(stderr, "[timing]   source-side run tables: %d runs of %d entries (%.1f per run), most in one group %d%s\n", nruns, n,
                         static_cast<double>(n) / std::max(nruns, 1), h_mx, view.f_rk ? "" : " -- not used (too many for the producers' LDS table)");
    }
    rcol.release(); rrow.release(); rperm.release();
    view.P = P.p;
    view.rq = rq.p;
    view.n_groups = ngroups;
    view.f_gptr = f_gptr.p;
    view.f_val = f_val.p;
    view.f_pos = f_pos.p;
    view.f_lcol = f_lcol.p;
}

namespace {

// compact CSR without the rows of zero compact length: one thread per ORIGINAL entry
__global__ void __launch_bounds__(kThreads) k_compact_cols(long nnz, int rows, const int *__restrict__ rp, const int *__restrict__ rp_c,
                                                          const int *__restrict__ col, int *__restrict__ col_c, int *__restrict__ map_c) {
    const long k = static_cast<long>(blockIdx.x) * kThreads + threadIdx.x;
    if (k >= nnz) return;
    int lo = 0, hi = rows;  // rp[lo] <= k < rp[hi]
    while (hi - lo > 1) {
        const int mid = lo + ((hi - lo) >> 1);
        if (rp[mid] <= k) lo = mid;
        else hi = mid;
    }
    if (rp_c[lo + 1] == rp_c[lo]) return;  // a row left out (or empty)
    const int q = rp_c[lo] + static_cast<int>(k - rp[lo]);
    col_c[q] = col[k];
    map_c[q] = static_cast<int>(k);
}

__global__ void __launch_bounds__(kThreads) k_compose_perm(long n, int *__restrict__ perm, const int *__restrict__ map) {
    const long i = static_cast<long>(blockIdx.x) * kThreads + threadIdx.x;
    if (i < n && perm[i] >= 0) perm[i] = map[perm[i]];
}

}  // namespace

void compact_without_rows(long nnz, int rows, const int *rp_dev, const int *rp_c_dev, const int *col_dev, int *col_c, int *map_c, hipStream_t s) {
    if (nnz > 0) hipLaunchKernelGGL(k_compact_cols, dim3(grid_for(nnz)), dim3(kThreads), 0, s, nnz, rows, rp_dev, rp_c_dev, col_dev, col_c, map_c);
    HIP_CHECK(hipStreamSynchronize(s));
}

void DeviceTiled::compose_perms(const int *map_dev, hipStream_t s) {
    if (n_tile > 0) hipLaunchKernelGGL(k_compose_perm, dim3(grid_for(n_tile)), dim3(kThreads), 0, s, n_tile, tperm.p, map_dev);
    if (n_rem > 0 && f_perm.p) hipLaunchKernelGGL(k_compose_perm, dim3(grid_for(n_rem)), dim3(kThreads), 0, s, n_rem, f_perm.p, map_dev);
    HIP_CHECK(hipStreamSynchronize(s));
}

void DeviceTiled::set_side(int rows, const int *rp, const std::vector<int> &long_rows) {
    // block list of the stream kernel over the long rows only: {slot, 1, first nonzero, count} (vector mode), rows over
    // kSplitRow as chunks {chunk slot, 0, first, count} + an entry {slot, first chunk slot, one past the last, 0}
    std::vector<int4> blk, lng;
    int chunk_slots = 0;
    for (size_t q = 0; q < long_rows.size(); ++q) {
        const int i = long_rows[q], len = rp[i + 1] - rp[i];
        if (len > kSplitRow) {
            const int first = chunk_slots;
            for (int k = rp[i]; k < rp[i + 1]; k += kSplitRow) blk.push_back(make_int4(chunk_slots++, 0, k, std::min(kSplitRow, rp[i + 1] - k)));
            lng.push_back(make_int4(static_cast<int>(q), first, chunk_slots, 0));
        } else {
            blk.push_back(make_int4(static_cast<int>(q), 1, rp[i], len));
        }
    }
    side_blk.alloc(blk.size());
    side_blk.upload(blk.data(), blk.size());
    side_rows.alloc(long_rows.size());
    side_rows.upload(long_rows.data(), long_rows.size());
    base.alloc_zero(static_cast<size_t>(rows));
    view.side_blk = side_blk.p;
    view.side_nblk = static_cast<int>(blk.size());
    view.side_rows = side_rows.p;
    view.base = base.p;
    view.side_long = nullptr;
    view.side_nlong = 0;
    view.side_partial = nullptr;
    if (!lng.empty()) {
        side_long.alloc(lng.size());
        side_long.upload(lng.data(), lng.size());
        side_partial.alloc_zero(static_cast<size_t>(chunk_slots) * 2);
        view.side_long = side_long.p;
        view.side_nlong = static_cast<int>(lng.size());
        view.side_partial = side_partial.p;
    }
}

void DeviceTiled::dump_wgtimes() const {
    if (!wgtimes.p || view.grid <= 0) return;
    std::vector<unsigned long long> h(static_cast<size_t>(view.grid) * 8);
    wgtimes.download(h.data(), h.size());
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int w = 0; w < view.grid; ++w) {
        if (h[w * 8] == 0) continue;
        t0 = std::min(t0, h[w * 8]);
        t1 = std::max(t1, h[w * 8 + 7]);
    }
    if (env_get("HPRLP_PB_STAMPS")) {  // developer build -DHPRLP_PB_PHASE_STAMPS=1: the slots hold phase durations of k_pb_fused (100 MHz ticks)
        double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, mx[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int cnt = 0;
        for (int w = 0; w < view.grid; ++w) {
            if (h[w * 8 + 7] == 0) continue;
            ++cnt;
            for (int i = 0; i < 8; ++i) {
                sum[i] += h[w * 8 + i] / 100.0;
                mx[i] = std::max(mx[i], h[w * 8 + i] / 100.0);
            }
        }
        if (cnt)
            std::fprintf(stderr, "[pb stamps] %d workgroups, mean (max) us: wait+barrier %.1f (%.1f) | stage %.1f (%.1f) | flags+scan %.1f (%.1f) | level 1 %.1f (%.1f) | "
                                 "level 2 %.1f (%.1f) | epilogue %.1f (%.1f) | push %.1f (%.1f) | total %.1f (%.1f)\n",
                         cnt, sum[0] / cnt, mx[0], sum[1] / cnt, mx[1], sum[2] / cnt, mx[2], sum[3] / cnt, mx[3], sum[4] / cnt, mx[4], sum[5] / cnt, mx[5],
                         sum[6] / cnt, mx[6], sum[7] / cnt, mx[7]);
        return;
    }
    if (t1 == 0) return;
    if (const char *f = env_get("HPRLP_WG_TIMES_DUMP")) {
        // raw table for tools/wgtimes_analyze.py: workgroup, XCC, CU key (SE/SH/CU bits of HW_ID), start, up to 5 super-block ends, end [us]
        if (FILE *fp = std::fopen(f, "a")) {
            std::fprintf(fp, "# launch grid %d\n", view.grid);
            for (int w = 0; w < view.grid; ++w) {
                if (h[w * 8] == 0) continue;
                const unsigned long long id = h[w * 8 + 6];
                std::fprintf(fp, "%d %llu %llu %.2f", w, id >> 32, (id >> 8) & 0xffull, (h[w * 8] - t0) / 100.0);
                for (int q = 1; q <= 5; ++q) std::fprintf(fp, " %.2f", h[w * 8 + q] ? (h[w * 8 + q] - t0) / 100.0 : -1.0);
                std::fprintf(fp, " %.2f\n", (h[w * 8 + 7] - t0) / 100.0);
            }
            std::fclose(fp);
        }
    }
    // wall clock: 100 MHz
    std::fprintf(stderr, "[wg times] %d workgroups, kernel span %.1f us (first start to last end)\n", view.grid, (t1 - t0) / 100.0);
    for (int x = 0; x < 8; ++x) {
        double s_min = 1e30, s_max = 0, e_min = 1e30, e_max = 0, e_sum = 0;
        int cnt = 0, rounds_max = 0;
        for (int w = x; w < view.grid; w += 8) {
            if (h[w * 8] == 0) continue;
            const double st = (h[w * 8] - t0) / 100.0, en = (h[w * 8 + 7] - t0) / 100.0;
            s_min = std::min(s_min, st); s_max = std::max(s_max, st);
            e_min = std::min(e_min, en); e_max = std::max(e_max, en);
            e_sum += en;
            ++cnt;
            int r = 0;
            for (int q = 1; q <= 5; ++q) r += h[w * 8 + q] != 0;
            rounds_max = std::max(rounds_max, r);
        }
        if (cnt) std::fprintf(stderr, "[wg times]   XCD %d: %d workgroups, start %.1f..%.1f us, end %.1f..%.1f us (mean %.1f), up to %d super-blocks each\n", x, cnt,
                              s_min, s_max, e_min, e_max, e_sum / cnt, rounds_max);
    }
    // distribution of per-super-block durations by round
    for (int q = 1; q <= 4; ++q) {  // (slot 6 holds the hardware id)
        double mn = 1e30, mx = 0, sum = 0;
        int cnt = 0;
        for (int w = 0; w < view.grid; ++w) {
            if (h[w * 8 + q] == 0) continue;
            const double d = (h[w * 8 + q] - h[w * 8 + q - 1]) / 100.0;
            mn = std::min(mn, d); mx = std::max(mx, d); sum += d;
            ++cnt;
        }
        if (cnt) std::fprintf(stderr, "[wg times]   super-block %d of a workgroup: %d workgroups, %.1f..%.1f us (mean %.1f)\n", q, cnt, mn, mx, sum / cnt);
    }
}

void DeviceTiled::dump_stamps() const {
    if (!stamps.p || view.n_pieces <= 0) return;
    std::vector<unsigned long long> h(static_cast<size_t>(view.n_pieces) * 16);
    stamps.download(h.data(), h.size());
    double sum[16] = {0};
    for (int p = 0; p < view.n_pieces; ++p)
        for (int k = 0; k < 16; ++k) sum[k] += static_cast<double>(h[static_cast<size_t>(p) * 16 + k]);
    const double steps = std::max(sum[6], 1.0);
    static const char *name[6] = {"barrier A", "tile wait + LDS store", "tile load issue", "barrier B", "entries wait + LDS accumulate", "entry load issue"};
    std::fprintf(stderr, "[tile stamps] %d pieces, %.0f steps; shader cycles per step (wave 0):\n", view.n_pieces, sum[6]);
    for (int k = 0; k < 6; ++k) std::fprintf(stderr, "[tile stamps]   %-32s %9.1f\n", name[k], sum[k] / steps);
    std::fprintf(stderr, "[tile stamps]   whole sweep per step %9.1f; per piece: remainder %.0f, store issue %.0f cycles\n", sum[8] / steps,
                 sum[9] / view.n_pieces, sum[10] / view.n_pieces);
}

// warm-up (abi.cpp: hprlp_warmup): an attribute query makes the runtime load this translation unit's code object now instead
// of at the first launch of one of its kernels
void warm_tiled_build_tu() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&k_run_heads));
}

}  // namespace hprlp
