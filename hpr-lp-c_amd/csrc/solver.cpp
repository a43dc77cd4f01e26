// solver.cpp -- device-resident HPR-LP solve on one GPU or one rank of a row-partitioned job.
//
// Host orchestration only: every vector operation is a kernel from kernels.hip on `stream`; the host
// synchronises exactly where the reference does (one scalar fetch per residual evaluation / sigma
// update, reference src/utils.cu:65-69).  Normal iterations between two residual evaluations are
// replayed from captured hipGraphs of 2 launches per iteration.
#include "solver.h"
#include "env.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <iomanip>
#include <iostream>
#include <limits>
#include <thread>

#include "dist.h"
#include "reorder.h"

namespace hprlp {

constexpr int kMaxGraphIters = 64;
constexpr int kDeviceStartRows = 1000000;  // power iteration: start vectors longer than this are generated on the device
constexpr int kPowerBlockKey = -10;  // Solver::graphs: the power iteration's block of ten iterations (positive keys: normal iterations)

// ------------------------------------------------------------------------------------------------
// cuts (optional): for rows that are to be split at given entries -- row index -> ascending chunk starts relative to the row's
// first entry (the first one is 0) and the XCD share (0..7) each chunk should run on, -1 for "anywhere".  See slab_cuts().
struct RowCuts {
    std::vector<int> rows;                 // ascending
    std::vector<std::vector<int>> starts;  // per row
    std::vector<std::vector<int>> share;   // per chunk
    int find(int r) const {
        const auto it = std::lower_bound(rows.begin(), rows.end(), r);
        return it != rows.end() && *it == r ? static_cast<int>(it - rows.begin()) : -1;
    }
};

static std::vector<int4> build_row_blocks_cut(int rows, const int *rowptr, std::vector<int4> *longrows, const RowCuts *cuts) {
    std::vector<int4> blk;
    std::vector<int> blk_share;  // wanted XCD share of a block, -1: none
    blk.reserve(static_cast<size_t>(rows) / 8 + 16);
    // block shape: at most cap_rows rows and cap_nnz nonzeros per wave (tuning knobs; the kernel needs
    // rows <= kStreamRows and nonzeros <= kStreamW)
    // measured (tools/latency_probe.py, config 3): launch-latency-bound matrices run 9 % faster with twice
    // as many, half as long blocks (256: 15.9 us/iteration, 512: 17.4, 128: 16.0); large ones stream best at 512
    int cap_rows = kStreamRows, cap_nnz = (rows > 0 && rowptr[rows] < (1 << 22)) ? kStreamW / 2 : kStreamW;
    if (const char *e = env_get("HPRLP_STREAM_ROWS")) cap_rows = std::min(kStreamRows, std::max(1, std::atoi(e)));
    if (const char *e = env_get("HPRLP_STREAM_NNZ")) cap_nnz = std::min(kStreamW, std::max(16, std::atoi(e)));
    int r = 0, slots = 0;
    while (r < rows) {
        const int len = rowptr[r + 1] - rowptr[r];
        const int cq = (cuts && longrows) ? cuts->find(r) : -1;
        if (cq >= 0) {
            // a dense row cut where its columns cross into another XCD's eighth of the gathered vector (slab_cuts)
            const std::vector<int> &st = cuts->starts[cq];
            const int first = slots;
            for (size_t q = 0; q < st.size(); ++q) {
                const int b0 = st[q], e0 = q + 1 < st.size() ? st[q + 1] : len;
                blk.push_back(make_int4(slots++, 0, rowptr[r] + b0, e0 - b0));
                blk_share.push_back(cuts->share[cq][q]);
            }
            longrows->push_back(make_int4(r, first, slots, 0));
            ++r;
            continue;
        }
        if (len > kSplitRow && longrows) {
            // a dense row/column (LPs have them): kSplitRow-sized chunks on separate waves, descriptor
            // {chunk slot, 0, first nonzero, count}; the row itself is finished by k_long_finish
            const int first = slots;
            for (int k = rowptr[r]; k < rowptr[r + 1]; k += kSplitRow) {
                blk.push_back(make_int4(slots++, 0, k, std::min(kSplitRow, rowptr[r + 1] - k)));
                blk_share.push_back(-1);
            }
            longrows->push_back(make_int4(r, first, slots, 0));
            ++r;
            continue;
        }
        if (len > kLongRow) {
            blk.push_back(make_int4(r, 1, rowptr[r], len));
            blk_share.push_back(-1);
            ++r;
            continue;
        }
        const int start = r;
        int nz = 0;
        while (r < rows && r - start < cap_rows) {
            const int l2 = rowptr[r + 1] - rowptr[r];
            if (l2 > kLongRow || (nz + l2 > cap_nnz && r > start)) break;
            nz += l2;
            ++r;
        }
        blk.push_back(make_int4(start, r - start, rowptr[start], nz));
        blk_share.push_back(-1);
    }
    // The stream kernel gives every XCD a contiguous eighth of the block list (kernels.hip: k_spmv_fused), and a vector-mode
    // block -- a long row or a chunk of a split row -- is up to eight times the work of a stream block.  LPs carry such rows
    // in bunches (linking constraints at the end of a block-angular model: 400 chunk blocks in the last XCD's range made that
    // XCD's share 1.6 x the others').  Chunks with a wanted share go to the front of that share (their columns lie in the
    // eighth of the vector that share's rows mostly read: one L2 holds it); the other heavy blocks are spread evenly; light
    // blocks keep their order (neighbouring rows stay neighbours); split-row chunk slots keep ascending inside a share.
    const size_t N = blk.size();
    if (N >= 256) {
        std::vector<int4> light, heavy, pinned[8];
        for (size_t i = 0; i < N; ++i) {
            const int4 &d = blk[i];
            if (blk_share[i] >= 0) pinned[blk_share[i] & 7].push_back(d);
            else if (d.y == 0 || (d.y == 1 && d.w > kLongRow)) heavy.push_back(d);
            else light.push_back(d);
        }
        size_t npinned = 0;
        for (int x = 0; x < 8; ++x) npinned += pinned[x].size();
        if ((!heavy.empty() || npinned > 0) && !light.empty()) {
            // the shares' block ranges, as k_spmv_fused maps them: grid = ceil(N / 4) workgroups of 4 blocks
            const size_t grid = (N + kWavesPerBlock - 1) / kWavesPerBlock, per_lo = grid >> 3, rem = grid & 7;
            size_t ih = 0, il = 0, pos = 0;
            const size_t rest = N - npinned, H = heavy.size();
            size_t placed_rest = 0;  // light + evenly spread heavy blocks placed so far
            for (int x = 0; x < 8; ++x) {
                const size_t wg0 = x * per_lo + std::min<size_t>(x, rem), wg1 = wg0 + per_lo + (static_cast<size_t>(x) < rem ? 1 : 0);
                const size_t end = std::min(N, wg1 * kWavesPerBlock);
                size_t ip = 0;
                while (pos < end) {
                    if (ip < pinned[x].size()) {
                        blk[pos++] = pinned[x][ip++];
                        continue;
                    }
                    // heavy block ih belongs at position ih * rest / H of the unpinned sequence
                    if (ih < H && (il >= light.size() || ih * rest / H <= placed_rest)) blk[pos++] = heavy[ih++];
                    else if (il < light.size()) blk[pos++] = light[il++];
                    else break;
                    ++placed_rest;
                }
                // pinned chunks that did not fit their share (more of them than the share holds) spill into the next one
                if (ip < pinned[x].size()) {
                    std::vector<int4> &nx = pinned[std::min(x + 1, 7)];
                    if (x < 7) nx.insert(nx.begin(), pinned[x].begin() + static_cast<long>(ip), pinned[x].end());
                    else throw std::runtime_error("row blocks: pinned chunks exceed the block list");
                }
            }
            if (pos != N || ih != H || il != light.size()) throw std::runtime_error("row blocks: arrangement lost a block");
        }
    }
    return blk;
}

std::vector<int4> build_row_blocks(int rows, const int *rowptr, std::vector<int4> *longrows) {
    return build_row_blocks_cut(rows, rowptr, longrows, nullptr);
}

// Dense rows of a large matrix, cut at the columns where the gathered vector's XCD eighths meet.  A row of thousands of entries
// gathers from all over the vector: every gather pulls a 128-byte line into the L2 of whichever XCD runs the chunk, and 200
// such rows (linking constraints) were a third of the y-half's memory traffic on the block-angular ladder point.  Chunks by
// column eighth, each run by the XCD whose own rows read that eighth anyway: the lines are fetched once.  cols_of(row, out):
// the row's column indices (host copy or a download).
constexpr int kSlabSplitMin = 2048;   // rows at least this long are cut by column eighths (shorter ones: one wave, anywhere)
constexpr int kSlabChunkMin = 96;     // a chunk shorter than this joins its predecessor
template <class ColsOf>
static RowCuts slab_cuts(int rows, int cols, const int *rp, ColsOf cols_of) {
    RowCuts rc;
    const int W = (cols + 7) / 8;
    std::vector<int> ci;
    for (int r = 0; r < rows; ++r) {
        const int len = rp[r + 1] - rp[r];
        if (len < kSlabSplitMin) continue;
        ci.resize(static_cast<size_t>(len));
        cols_of(r, ci.data());
        std::vector<int> st{0}, sh{std::min(ci[0] / W, 7)};
        for (int k = 1; k < len; ++k) {
            const int slab = std::min(ci[k] / W, 7);
            const bool too_long = k - st.back() >= kSplitRow;
            if ((slab != sh.back() && k - st.back() >= kSlabChunkMin) || too_long) {
                st.push_back(k);
                sh.push_back(slab);
            }
        }
        // a short tail joins its predecessor -- unless the joined chunk would pass the split-row limit (a 4100-entry row in one
        // eighth: [0,4096) + [4096,4100) stays two chunks)
        if (st.size() > 1 && len - st.back() < kSlabChunkMin && len - st[st.size() - 2] <= kSplitRow) {
            st.pop_back();
            sh.pop_back();
        }
        rc.rows.push_back(r);
        rc.starts.push_back(std::move(st));
        rc.share.push_back(std::move(sh));
    }
    return rc;
}

// The kernels trust the block list: every chunk slot exactly once, chunk and block sizes within what a wave handles.
// Returns the number of chunk slots.
static int check_row_blocks(const std::vector<int4> &b) {
    int nslots = 0;
    std::vector<char> seen;
    for (const int4 &d : b) {
        const bool vec = d.y == 0 || (d.y == 1 && d.w > kLongRow);
        if (d.y == 0) {
            // (chunk slots: each exactly once -- the arrangement by XCD shares moves chunks, k_long_finish adds a row's slots in order)
            if (d.w < 1 || d.w > kSplitRow || d.x < 0) throw std::runtime_error("bad split-row block");
            if (static_cast<size_t>(d.x) >= seen.size()) seen.resize(static_cast<size_t>(d.x) + 1, 0);
            if (seen[d.x]++) throw std::runtime_error("bad split-row block: slot used twice");
            ++nslots;
        } else if (!vec && (d.w > kStreamW || d.y > kStreamRows || d.y < 1)) {
            throw std::runtime_error("bad row block");
        }
    }
    if (static_cast<size_t>(nslots) != seen.size()) throw std::runtime_error("bad split-row blocks: slots not contiguous");
    return nslots;
}

// Host only (no device call): the stream kernel's block list of a CSR pattern as describe_when() would build it, checked.
// out: {blocks, split rows, chunk slots, rows cut by column eighths, longest chunk, entries covered}.
void row_block_plan_host(int rows, int cols, const int *rp, const int *ci, bool with_cuts, long out[6]) {
    RowCuts cuts;
    if (with_cuts) cuts = slab_cuts(rows, cols, rp, [&](int r, int *o) { std::memcpy(o, ci + rp[r], static_cast<size_t>(rp[r + 1] - rp[r]) * sizeof(int)); });
    std::vector<int4> lr;
    const std::vector<int4> b = build_row_blocks_cut(rows, rp, &lr, cuts.rows.empty() ? nullptr : &cuts);
    const int nslots = check_row_blocks(b);
    long longest = 0, covered = 0;
    for (const int4 &d : b) {
        if (d.y == 0) longest = std::max<long>(longest, d.w);
        covered += d.w;
    }
    out[0] = static_cast<long>(b.size());
    out[1] = static_cast<long>(lr.size());
    out[2] = nslots;
    out[3] = static_cast<long>(cuts.rows.size());
    out[4] = longest;
    out[5] = covered;
}

static int workgroup_slots() {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return cus * kTileResidentPerCu;
}

constexpr int kSkewRow = 256;              // build_tiled_copy: rows longer than this count as long ...
constexpr double kMaxLongRowShare = 0.2;   // ... and a matrix with more than this share of its entries in them keeps the stream kernel
constexpr double kMaxBlockLoad = 4.0;      // build_tiled_copy: heaviest block of sb_rows rows / mean, above which the fused tiled forms are declined
constexpr double kPiecesMinDense = 0.75;   // build_tiled_copy: share of the entries in staged tiles below which the PIECE form is declined
constexpr double kStreamLineDensity = 0.25;  // build_tiled_copy: at most this many 64-byte lines gathered per entry -> stream kernel
constexpr double kStreamL2LineDensity = 0.6;  // ... and only while neighbouring rows still share lines: with a line per entry the L2 holds the window but every gather
                                              // misses the L1 (1.5M x 1.5M, band 75 000, 0.93 lines per entry: stream 0.177 ms per half-step, pieces 0.128; the
                                              // multicommodity-flow LP the rule was made for: 0.35 / 0.15)
constexpr int kTiledMinCols = 7 << 16;        // build_tiled_copy: fewest columns of a matrix that is tried in a tiled form (458 752: 3.5 MiB of gathered vector)
constexpr double kStreamLineDensityLong = 0.12;  // ... lines per entry up to which rows of ANY length keep the coalesced-rows preference
constexpr double kCoalescedMaxRowEntries = 32.0;   // build_tiled_copy: the coalesced-rows preference for the stream kernel holds up to this many entries per row
constexpr double kStreamL2LineDensityFused = 0.5;  // ... the same against a FUSED tiled form that needs its longest rows kept aside (build_tiled_copy)
constexpr double kFewRowsTileShare = 1.8;    // build_tiled_copy: most tile bytes per entry byte at which a matrix of few rows is still tried in the piece form
                                             // (threshold sweep, tools/form_regret.py --corpus boundaries: piece form ahead of the all-remainder form by 12-41 % at
                                             // 0.8 / 1.2 / 1.5, level at 1.3, behind by 8-17 % at 2.2 / 2.8 and 2-3 x from 3.2 on)
constexpr double kPopularFarShare = 0.8;     // build_tiled_copy: share of the remainder that 2 MB of the gathered vector serve, from which ...
constexpr double kPopularFarMinRem = 0.1;    // ... a copy with at least this share of its entries in the remainder is dropped for the stream kernel (one-L2 window)
constexpr double kPiecesThinRows = 10.0;     // build_tiled_copy: below this many entries per row a PIECE-form copy is dropped for the stream kernel
constexpr double kPiecesMinRowEntries = 16.0;  // ... or while rows are thin (build_tiled_copy)
constexpr double kStreamL2Bytes = 3.0e6;  // build_tiled_copy: an XCD's share of the gathered vector that one 4 MiB L2 keeps beside the matrix stream

// HPRLP_TIMING=1: wall time of the set-up phases on stderr
struct PhaseTimer {
    bool on;
    clock_type::time_point t0;
    PhaseTimer() : on(env_get("HPRLP_TIMING") != nullptr), t0(time_now()) {}
    void tick(const char *what) {
        if (on) std::cerr << "[timing] " << what << ": " << time_since(t0) << " s" << std::endl;
        t0 = time_now();
    }
};

// A long copy on its own stream (driven by a helper thread) as a sequence of pieces with a short sleep after each.  While a
// pageable-memory copy is inside the HIP runtime, other threads' runtime calls wait for it: beside one 1.6 GB copy the device build
// of the tiled copy stood still for 20 ms at its first stream wait, and back-to-back pieces starved it just the same.  With the
// sleep the waiting thread gets its turn (measured on config 5, warm process: set-up of A 89 -> 67 ms; 16 / 32 / 64 MB pieces
// with 30 / 100 us equal within 3 ms).  HPRLP_COPY_PIECE_MB / HPRLP_COPY_PAUSE_US: for measurements.
static hipError_t copy_in_pieces(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t cs) {
    static const size_t kPiece = size_t(env_get("HPRLP_COPY_PIECE_MB") ? std::max(1, std::atoi(env_get("HPRLP_COPY_PIECE_MB"))) : 64) << 20;
    static const int pause_us = env_get("HPRLP_COPY_PAUSE_US") ? std::atoi(env_get("HPRLP_COPY_PAUSE_US")) : 100;
    for (size_t off = 0; off < bytes; off += kPiece) {
        const size_t len = std::min(kPiece, bytes - off);
        hipError_t e = hipMemcpyAsync(static_cast<char *>(dst) + off, static_cast<const char *>(src) + off, len, kind, cs);
        if (e == hipSuccess) e = hipStreamSynchronize(cs);
        if (e != hipSuccess) return e;
        if (pause_us > 0) std::this_thread::sleep_for(std::chrono::microseconds(pause_us));
    }
    return hipSuccess;
}

void DeviceMatrix::upload(int rows, int cols, const int *rp, const int *ci, const double *v, std::shared_ptr<void> keep) {
    PhaseTimer pt;
    const int nnz = rp[rows];
    // the kernels index without bounds checks: refuse anything that could fault on the device
    for (int i = 0; i < rows; ++i)
        if (rp[i + 1] < rp[i]) throw std::runtime_error("row pointer array is not monotone");
    const bool check_on_device = nnz > 4000000;  // (a pass over the uploaded copy: 0.3 ms instead of 10 ms of 8 host threads at 2e8)
    if (!check_on_device) {
        for (long k = 0; k < nnz; ++k)
            if (ci[k] < 0 || ci[k] >= cols) throw std::runtime_error("column index out of range");
    }
    pt.tick("  validate indices");
    rowptr.alloc(static_cast<size_t>(rows) + 1);
    rowptr.upload(rp, static_cast<size_t>(rows) + 1);
    col.alloc(nnz);
    col.upload(ci, nnz);
    if (check_on_device) {
        DBuf<int> bad;
        bad.alloc_zero(1);
        launch_check_columns(nnz, cols, col.p, bad.p, nullptr);
        int b = 0;
        bad.download(&b, 1);
        if (b) throw std::runtime_error("column index out of range");
    }
    val.alloc(nnz);
    // The values are not needed before the tiled copy is filled (or, without one, before describe() returns): a large array
    // travels on a copy stream of its own, driven by a helper thread, beside the host-side row blocks and the device build
    // of the tiled copy, which work on the index arrays (config 5: 27 ms of 1.6 GB hidden).
    std::future<void> val_job;
    if (nnz > 4000000 && env_get("HPRLP_NO_SETUP_OVERLAP") == nullptr) {
        int dev = 0;
        HIP_CHECK(hipGetDevice(&dev));
        double *dst = val.p;
        val_job = std::async(std::launch::async, [dev, dst, v, nnz]() {
            HIP_CHECK(hipSetDevice(dev));
            hipStream_t cs = nullptr;
            HIP_CHECK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
            const hipError_t e = copy_in_pieces(dst, v, static_cast<size_t>(nnz) * sizeof(double), hipMemcpyHostToDevice, cs);
            (void)hipStreamDestroy(cs);
            HIP_CHECK(e);
        });
    } else {
        val.upload(v, nnz);
    }
    pt.tick("  upload CSR (values may still be in flight)");
    std::promise<const int *> have_rp;
    have_rp.set_value(rp);
    describe_when(rows, cols, nnz, have_rp.get_future().share(), ci, std::move(keep), -1.0, &val_job);
}

void DeviceMatrix::describe(int rows, int cols, const int *rp, const int *ci, std::shared_ptr<void> keep, double min_dense_override) {
    std::promise<const int *> have_rp;
    have_rp.set_value(rp);
    describe_when(rows, cols, rp[rows], have_rp.get_future().share(), ci, std::move(keep), min_dense_override, nullptr);
}

// Row blocks, kernel views and the tiled copy, for a matrix whose CSR arrays are already in rowptr / col / val on the device.
// rp_ready delivers the same row pointers on the host (for A^T after a device transpose: when their download, running in a thread
// of the caller, is done); ci: the host column indices or null.  The row blocks of the stream kernel are a host-side walk over the
// row pointers (13-18 ms at 1e7 rows): a job of their own beside the device build of the tiled copy, which reads only the device
// arrays.  values_ready (optional): the upload of val still in flight -- joined before the first use of the values.
void DeviceMatrix::describe_when(int rows, int cols, long nnz_l, std::shared_future<const int *> rp_ready, const int *ci, std::shared_ptr<void> keep,
                                 double min_dense_override, std::future<void> *values_ready) {
    PhaseTimer pt;
    view.longrows = nullptr;  // (describe() may run again on a permuted copy of the matrix: start from a clean view)
    view.nlong = 0;
    view.long_partial = nullptr;
    view.tiled = TiledDev();
    const int nnz = static_cast<int>(nnz_l);
    auto host_rp = [&rp_ready]() { return rp_ready.get(); };
    auto join_values = [values_ready]() {
        if (values_ready && values_ready->valid()) values_ready->get();
    };
    struct Blocks {
        std::vector<int4> b, lr;
    };
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    const int *dcol = col.p;
    // dense rows of a matrix whose gathered vector is beyond one L2: cut by column eighths (slab_cuts); the column indices of
    // those rows come from the host copy or, for a matrix built on the device, from a download of just those rows
    auto make_blocks = [rows, cols, nnz, rp_ready, ci, dev, dcol]() {
        HIP_CHECK(hipSetDevice(dev));
        const int *rp = rp_ready.get();
        RowCuts cuts;
        const char *noslab = env_get("HPRLP_NO_SLAB_CUTS");
        if (cols >= (1 << 19) && nnz >= (1 << 22) && !(noslab && noslab[0] == '1')) {
            cuts = slab_cuts(rows, cols, rp, [&](int r, int *out) {
                const size_t len = static_cast<size_t>(rp[r + 1] - rp[r]);
                if (ci) std::memcpy(out, ci + rp[r], len * sizeof(int));
                else HIP_CHECK(hipMemcpy(out, dcol + rp[r], len * sizeof(int), hipMemcpyDeviceToHost));
            });
        }
        Blocks out;
        out.b = build_row_blocks_cut(rows, rp, &out.lr, cuts.rows.empty() ? nullptr : &cuts);
        return out;
    };
    const bool overlap = rows > 100000 && env_get("HPRLP_NO_SETUP_OVERLAP") == nullptr;
    std::future<Blocks> blocks_job = std::async(overlap ? std::launch::async : std::launch::deferred, make_blocks);
    view.rows = rows;
    view.cols = cols;
    view.nnz = nnz;
    view.rowptr = rowptr.p;
    view.col = col.p;
    view.val = val.p;
    // nontemporal matrix loads only when the matrix cannot stay in the eight 4 MiB L2s anyway
    view.nt = static_cast<size_t>(nnz) * 12 > (static_cast<size_t>(16) << 20);
    if (const char *e = env_get("HPRLP_NT")) view.nt = std::atoi(e) != 0;
    try {
        longest_row = 0;
        long_row_share = 0.0;
        if (rows > 100000) {  // (from the device copy: the host row pointers of A^T may still be on their way)
            long in_long = 0;
            longest_row = launch_longest_row(rowptr.p, rows, nullptr, kSkewRow, &in_long);
            long_row_share = nnz > 0 ? static_cast<double>(in_long) / nnz : 0.0;
        } else {
            const int *rp = host_rp();
            for (int i = 0; i < rows; ++i) longest_row = std::max(longest_row, rp[i + 1] - rp[i]);
        }
        build_tiled_copy(rows, cols, nnz, host_rp, ci, std::move(keep), min_dense_override, join_values, pt);
        join_values();
    } catch (...) {
        if (blocks_job.valid()) blocks_job.wait();
        throw;
    }
    Blocks B = blocks_job.get();
    std::vector<int4> &b = B.b, &lr = B.lr;
    const int nslots = check_row_blocks(b);
    blk.alloc(b.size());
    blk.upload(b.data(), b.size());
    pt.tick("  row blocks");
    if (!lr.empty()) {
        longrows.alloc(lr.size());
        longrows.upload(lr.data(), lr.size());
        long_partial.alloc_zero(static_cast<size_t>(nslots) * 2);
        view.longrows = longrows.p;
        view.nlong = static_cast<int>(lr.size());
        view.long_partial = long_partial.p;
    }
    view.blk = blk.p;
    view.nblk = static_cast<int>(b.size());
}

// The tiled copy of describe_when(): decision, device build (or the host builder's background job), values filled in.
void DeviceMatrix::build_tiled_copy(int rows, int cols, int nnz, const std::function<const int *()> &host_rp, const int *ci, std::shared_ptr<void> keep,
                                    double min_dense_override, const std::function<void()> &join_values, PhaseTimer &pt) {
    // column-tiled copy: only for matrices with at least one 8192-row super-block per CU
    // and with enough column locality (HPRLP_NO_TILED=1 disables; thresholds overridable for tests)
    const char *no = env_get("HPRLP_NO_TILED");
    // (a stream of its own for the build was measured: the stall beside the value upload is the runtime's lock, not a stream wait)
    const hipStream_t bs = nullptr;
    if (!(no && no[0] == '1')) {
        const char *mr = env_get("HPRLP_TILED_MIN_ROWS");
        const char *md = env_get("HPRLP_TILED_MIN_DENSE");
        // measured on shard-shaped matrices of the banded benchmark: 305 super-blocks 0.31 ms tiled vs
        // 0.38 ms stream, 153 super-blocks 0.21 ms both -> one super-block per CU is the break-even
        // (round 2: matrices with fewer super-blocks than CUs run the split form -- several workgroups per super-block)
        const int rb = sb_rows, gb = far_group;  // (heights, not bit counts)
        const bool short_form = rb < kTileRows;
        // (lowered height: chosen by Solver::choose_sb_rows so that there is a super-block per workgroup slot)
        // (a copy asked for WITHOUT a dense-tile requirement -- the all-remainder form, Solver::pb_fallback_wanted -- stages no tile:
        // the row count that makes staging pay does not apply to it)
        int min_rows = mr ? std::atoi(mr) : min_dense_override >= 0.0 ? 1 : (short_form ? 256 * rb : 32 * kTileRows);
        const double min_dense = min_dense_override >= 0.0 ? min_dense_override : (md ? std::atof(md) : 0.5);
        declined_sparse = false;
        declined_thin = declined_popular = false;
        declined_few_rows = false;
        // Two more conditions on the shape (measured late in round 2, tools/longrow_ab.py):
        //  * the gathered vector must be big enough for staging it to pay: a 300k x 100k matrix passes the dense-tile test but
        //    its 0.8 MB vector lives in every L2 anyway -- stream kernel 11.5 us, tiled 30.7 us per launch.  Tiled from 2^20
        //    columns on (8 MB: beyond an XCD's 4 MiB L2).  An explicit HPRLP_TILED_MIN_ROWS (tests) lifts the default.
        //  * no long rows: a row's entries beyond four per tile go to the remainder list, where ONE lane adds a row's
        //    consecutive products (two dependent LDS reads each): five rows of 3000 entries took that launch from 31 to 203
        //    us.  Such matrices keep the stream kernel, which spreads a long row over a wave or several.
        const char *mc = env_get("HPRLP_TILED_MIN_COLS");
        // (2^19 columns = 4 MiB = one L2.  Until round 5 the full-height form waited for 800 k columns: a 600k x 600k band of 40 000
        // columns, 40 per row, kept the stream kernel at 0.25 of 8 TB/s where the piece form runs 0.33 and the lowered fused form 0.36)
        // (7 * 2^16 since the threshold sweep of round 5: 2 % band, 20 per row: 400 k columns stream 0.070 / lowered tiled 0.078 ms,
        // 500 k columns 0.098 / 0.082 -- the vector shares its L2 with the matrix stream)
        const int min_cols = mc ? std::atoi(mc) : (mr ? 0 : kTiledMinCols);
        const int longest = longest_row;  // (describe_when)
        // (an all-remainder copy -- min_dense_override >= 0 -- takes rows of any length: its steps add a row's products by a segmented
        // reduction over the lanes, Solver::pb_fallback_wanted)
        declined_shape = cols < min_cols || (longest > kTileMaxRow && min_dense_override < 0.0);
        declined_long_rows = false;
        // Round 4, late.  A matrix of fewer full-height super-blocks than workgroup slots whose height could not be lowered (its
        // rows' column windows are too wide for short super-blocks) would run the piece form: partial sums through memory and a
        // finish launch.  When the stream kernel's gathers stay inside one L2 anyway -- every XCD runs a contiguous eighth of the
        // rows, whose columns (median row span + the eighth's own drift along the diagonal, Solver::choose_sb_rows) cover less
        // than kStreamL2Bytes of the vector -- the stream kernel is the faster form: multicommodity-flow LP, 535 k x 2.03 M,
        // 40 diagonal blocks: y-half 64.9 us (512 pieces of 66 super-blocks) against 22.4 us, 10.1 k -> 18.0 k iterations/s.
        // Config 5's quarter shard (window 4.0 MB: pieces 0.31 ms, stream 0.38) keeps the pieces.
        declined_l2 = false;
        // bytes of the vector tiles a FULL-height super-block stages against the bytes of its entries (the row span back out of
        // Solver::choose_sb_rows' estimate): above 1 the piece form moves more tile bytes than matrix bytes (staircase LP of 12 stages,
        // 8 entries per row, span 2.4e5 columns: pieces 0.086 ms per half-step, stream kernel 0.068)
        const double span_est = xcd_gather_bytes > 0.0 ? std::max(0.0, xcd_gather_bytes / 8.0 - cols / 8.0) : 0.0;
        const double tile_share_full = rows > 0 && nnz > 0 ? (span_est + static_cast<double>(kTileRows) * cols / rows) * 8.0 / (static_cast<double>(nnz) / rows * kTileRows * 11.0) : 0.0;
        // Second held-out set, round 5: FEW rows (under a super-block per CU) whose full-height tiles would still be dense -- the vector
        // bytes a super-block stages stay under kFewRowsTileShare times its entries' bytes -- go through the tiled build after all and run the
        // piece form: 50k x 2M with 400 random entries per row (the transpose of a 10-per-row matrix): 7 super-blocks in 512 pieces
        // 0.101 ms per half-step, all-remainder form 0.162, stream kernel 0.291.  (100k x 5M with 150 per row: share 3.2 -- all-remainder
        // form, Solver::pb_fallback_wanted.)
        if (!mr && min_dense_override < 0.0 && rb == kTileRows && rows < min_rows && rows >= 4 * kTileRows && nnz >= 4000000 && xcd_gather_bytes > 0.0 &&
            tile_share_full <= kFewRowsTileShare)
            min_rows = rows;
        // ... and thin rows: a piece's cost goes with the tiles it stages, the stream kernel's with the entries (1M x 1M band of 16 000
        // columns, 6 per row: pieces 0.060 ms per half-step, stream 0.041; 12 per row + dense borders: 0.102 / 0.088; 20 per row: 0.128 / 0.177)
        const double entries_per_row = rows > 0 ? static_cast<double>(nnz) / rows : 0.0;
        line_density = (rows >= 8192 && nnz > 1000000) ? launch_line_density(rowptr.p, col.p, rows, nullptr) : 1.0;
        if (pt.on) std::cerr << "[timing]   gathered 64-byte lines per entry (sampled 64-row windows): " << line_density << std::endl;
        // Rows whose neighbours gather from the same 64-byte lines (stencil rows, incidence matrices, bands a few hundred columns
        // wide) are what the stream kernel is good at: its gathers coalesce and hit the L1 / L2, and there is nothing for staged
        // tiles to save.  Measured (tools/ab_forms.sh, 1M x 1M, 20 per row): band 500 (0.1 lines per entry) stream 0.071 ms per
        // half-step, band 2000 (0.2) 0.107 -- the tiled build declines such bands (more than four entries of a row per tile) and the
        // all-remainder form that used to follow took 0.138 / 0.149; grid PDE-control LP (0.11 / 0.20): stream 0.0245 against
        // 0.0361 ms in the lowered fused form.  From 0.37 lines per entry on (band 4000) the fused tiled form wins (0.089 / 0.124).
        // (these two hold for a matrix whose longest rows would be kept aside as well: evaluated whatever the longest row is -- with the
        // layered tile lists of round 5 the copy of a block-angular LP WITHOUT its 400 linking rows passes the dense-tile test, and ran
        // 0.48 / 0.42 of 8 TB/s where the stream kernel, whose rows share their lines, runs 0.58 / 0.47)
        declined_coalesced = false;
        const bool long_only = declined_shape && cols >= min_cols;  // declined so far for its longest row alone
        // (rows of more than kCoalescedMaxRowEntries entries excepted: the stream kernel packs 512 entries per wave, so 60-entry rows leave
        // it 8 busy lanes in its row-sum phase -- 600k x 600k, 60 per row in 6 000 columns, 0.2 lines per entry: stream 0.24 / 0.35 of
        // 8 TB/s, lowered tiled form with three layers per tile 0.42 / 0.50)
        // (... unless the rows share their lines almost completely: 1M x 1M, 40 / 48 per row inside 1 500 columns, 0.08 lines per entry:
        // stream 0.243 / 0.298 ms per iteration, lowered tiled form 0.274 / 0.359 -- threshold sweep, round 5)
        if ((!declined_shape || long_only) && line_density <= kStreamLineDensity && (entries_per_row <= kCoalescedMaxRowEntries || line_density <= kStreamLineDensityLong) &&
            !mr && min_dense_override < 0.0 &&
            env_get("HPRLP_TILED_ANYWAY") == nullptr)
            declined_shape = declined_coalesced = true;
        {
            const bool pieces_expected = rb == kTileRows && (rows + rb - 1) / rb < workgroup_slots();
            const bool in_one_l2 = xcd_gather_bytes > 0.0 && xcd_gather_bytes <= kStreamL2Bytes;
            // pieces: round 4's rule with round 5's conditions.  A FUSED tiled form only where the copy would need its longest rows kept
            // aside (two more launches per half-step for them) and the stream kernel's rows share their lines inside one L2:
            // block-angular LP without its 400 linking rows (0.39 lines per entry, 2 MB per XCD): fused 1984-row form + side 0.48 / 0.42
            // of 8 TB/s, stream kernel 0.58 / 0.47.  (A band of 4 000 columns has the same line density and window and no long rows:
            // fused form 0.51, stream kernel 0.36.)
            const bool stream_wins = pieces_expected ? (line_density <= kStreamL2LineDensity || tile_share_full > 1.0 || entries_per_row < kPiecesMinRowEntries)
                                                     : (long_only && line_density <= kStreamL2LineDensityFused);
            if ((!declined_shape || (long_only && !declined_coalesced)) && in_one_l2 && stream_wins && env_get("HPRLP_PIECES_ANYWAY") == nullptr &&
                env_get("HPRLP_TILED_ANYWAY") == nullptr && !mr && min_dense_override < 0.0)
                declined_shape = declined_l2 = true;
        }
        const char *ht = env_get("HPRLP_HOST_TILING");
        const bool host_tiling = ht && ht[0] == '1';
        // Round 5, from the form-regret corpus (tools/form_regret.py, profiles/r05_form_regret.txt) -- two properties of the ROW
        // LENGTHS that the tiled forms do not survive, whatever the columns look like:
        //  * skew: a matrix with a fifth of its entries in rows of more than kSkewRow entries (R-MAT / Kronecker graphs: 40 %).  A
        //    long row's entries beyond four per tile all go through the remainder steps of ONE super-block; the stream kernel
        //    gives such a row a wave of its own.  Kronecker 2^20 x 2^20, 7.5e6 entries, y-half: piece form 0.67 ms, lowered fused
        //    0.54-0.58, all-remainder 0.71, stream kernel 0.076.
        //  * imbalance: the heaviest block of sb_rows consecutive rows holds more than kMaxBlockLoad times the mean (a few hundred
        //    coupling rows at the end of a block-diagonal model).  A fused launch ends when its heaviest super-block does:
        //    block-diagonal 1M x 1.2M with 300 rows of 900 entries behind it, y-half 0.295 ms (1984-row super-blocks) against
        //    0.066 with the stream kernel.  (The piece form cuts its work evenly and is exempt.)
        declined_skew = declined_imbalance = false;
        if (!mr && min_dense_override < 0.0 && env_get("HPRLP_TILED_ANYWAY") == nullptr) {
            if (long_row_share > kMaxLongRowShare) {   // (also for a matrix whose longest rows would be kept aside, below)
                declined_shape = declined_skew = true;
            } else if (!declined_shape && rows > 100000) {
                const int nsb = (rows + rb - 1) / rb;
                const bool pieces_expected = rb == kTileRows && nsb <= workgroup_slots();
                const int heaviest = pieces_expected ? 0 : launch_heaviest_block(rowptr.p, rows, rb, nullptr);
                if (static_cast<double>(heaviest) > kMaxBlockLoad * static_cast<double>(nnz) / nsb) declined_shape = declined_imbalance = true;
            }
            if (pt.on && (declined_skew || declined_imbalance))
                std::cerr << "[timing]   tiled forms declined for the row lengths: " << (declined_skew ? "skew" : "imbalance") << " (share of the entries in rows over "
                          << kSkewRow << ": " << long_row_share << ")" << std::endl;
        }
        // Thin rows (rule below, after the build: a PIECE-form copy of a matrix with under kPiecesThinRows entries per row is dropped
        // for the stream kernel) decided BEFORE the build where the cheap tiling test (a sort of the entries' tile keys, under a
        // millisecond; the locality ordering's acceptance test) already says the copy would pass: the build and its drop were
        // 20-65 ms per matrix of a 0.5 s solve (two-stage LP: 0.126 s of 0.57).
        {
            const bool pieces_expected = rb == kTileRows && (rows + rb - 1) / rb <= workgroup_slots();
            if ((!declined_shape || long_only) && pieces_expected && rows >= min_rows && nnz > 0 && entries_per_row < kPiecesThinRows && !host_tiling && !mr && !md &&
                min_dense_override < 0.0 && env_get("HPRLP_PIECES_ANYWAY") == nullptr && env_get("HPRLP_TILED_ANYWAY") == nullptr &&
                env_get("HPRLP_TILING_CHECK") == nullptr) {
                const double share = device_tiling_dense_fraction(rows, cols, nnz, rowptr.p, col.p, nullptr, nullptr, bs);
                if (pt.on) std::cerr << "[timing]   thin rows (" << entries_per_row << " per row), tiling test: " << share << " of the entries in dense tiles" << std::endl;
                if (share >= kPiecesMinDense) {
                    declined_thin = true;
                    pt.tick("  tiling test (thin rows: the stream kernel without a build)");
                    return;
                }
            }
        }
        // A FEW long rows (dense LP columns / rows) do not have to cost the matrix the tiled kernel: they are left out of the
        // tiled copy and summed by the stream kernel's vector / split-row mode into a base vector that every tiled launch
        // adds (tiled.h: TiledDev::side_*).  At most 0.1 % of the rows (and 64) and a fifth of the nonzeros.
        const char *nside = env_get("HPRLP_NO_LONG_SIDE");
        if (cols >= min_cols && longest > kTileMaxRow && rows >= min_rows && nnz > 0 && !host_tiling && !(nside && nside[0] == '1') && !declined_skew &&
            !declined_coalesced && !declined_l2) {
            const int *rp = host_rp();
            std::vector<int> long_rows;
            long long_nnz = 0;
            for (int i = 0; i < rows; ++i)
                if (rp[i + 1] - rp[i] > kTileMaxRow) {
                    long_rows.push_back(i);
                    long_nnz += rp[i + 1] - rp[i];
                }
            if (static_cast<long>(long_rows.size()) <= std::max<long>(64, rows / 1000) && long_nnz * 5 <= nnz) {
                std::vector<int> rp_c(static_cast<size_t>(rows) + 1, 0);
                for (int i = 0; i < rows; ++i) {
                    const int len = rp[i + 1] - rp[i];
                    rp_c[i + 1] = rp_c[i] + (len > kTileMaxRow ? 0 : len);
                }
                const long nnz_c = rp_c[rows];
                DBuf<int> d_rp_c(rp_c.size()), col_c(static_cast<size_t>(std::max<long>(nnz_c, 1))), map_c(static_cast<size_t>(std::max<long>(nnz_c, 1)));
                d_rp_c.upload(rp_c.data(), rp_c.size());
                compact_without_rows(nnz, rows, rowptr.p, d_rp_c.p, col.p, col_c.p, map_c.p, bs);
                const bool ok = tiled.build_on_device(rows, cols, nnz_c, d_rp_c.p, col_c.p, min_rows, min_dense, bs, rb, tile_cols, rem_cap);
                if (pt.on)
                    std::cerr << "[timing]   tiled copy without " << long_rows.size() << " long rows (" << long_nnz << " entries, longest " << longest
                              << "): " << (ok ? "" : "declined; ") << tiled.view.nsb << " super-blocks, " << tiled.n_steps << " steps" << std::endl;
                bool ok_kept = ok;
                if (ok && tiled.view.n_pieces > 0 && !mr && !md && env_get("HPRLP_PIECES_ANYWAY") == nullptr) {   // (as below: kPiecesMinDense)
                    const double staged = static_cast<double>(tiled.dense_entries) / std::max(1.0, static_cast<double>(tiled.dense_entries) + static_cast<double>(tiled.n_rem));
                    if (staged < kPiecesMinDense) {
                        tiled = DeviceTiled();
                        ok_kept = false;
                    } else if (entries_per_row < kPiecesThinRows) {   // (as below: thin rows)
                        tiled = DeviceTiled();
                        ok_kept = false;
                        declined_thin = true;
                    }
                }
                if (ok_kept) {
                    tiled.build_far(cols, bs, gb);
                    tiled.compose_perms(map_c.p, bs);
                    tiled.set_side(rows, rp, long_rows);
                    view.tiled = tiled.view;
                    join_values();
                    launch_tiled_refresh(tiled, val.p, bs);
                    HIP_CHECK(hipDeviceSynchronize());
                    declined_shape = false;
                }
                pt.tick("  build tiled copy (device, long rows aside)");
                if (ok_kept) return;
            }
        }
        if (declined_shape) {
            // declined for its row lengths alone (longest row, or too many entries in long rows) -- not because its rows share lines or
            // gather from one L2's window: a candidate for the all-remainder form where the COLUMNS are not popular (pb_fallback_wanted)
            declined_long_rows = cols >= min_cols && longest > kTileMaxRow && !declined_coalesced && !declined_l2 && !declined_imbalance && rows > 0 && nnz > 0;
            if (pt.on) std::cerr << "[timing]   tiled copy not attempted: " << cols << " columns, longest row " << longest << std::endl;
        } else if (rows < min_rows) {
            declined_few_rows = rows > 0 && nnz > 0;  // (Solver::pb_fallback_wanted)
            if (pt.on) std::cerr << "[timing]   tiled copy not attempted: " << rows << " rows (staged tiles from " << min_rows << " on)" << std::endl;
        } else if (rows >= min_rows && rows > 0 && nnz > 0 && !host_tiling) {
            // built on the device from the device CSR arrays (tiled_build.hip); HPRLP_TILING_CHECK=1 also runs the
            // host builder and compares every array
            const bool ok = tiled.build_on_device(rows, cols, nnz, rowptr.p, col.p, min_rows, min_dense, bs, rb, tile_cols, rem_cap);
            declined_sparse = !ok;  // rows >= min_rows here: what was missing is dense tiles
            if (pt.on)
                std::cerr << "[timing]   tiled copy (device build): " << (ok ? "" : "declined; ") << tiled.view.nsb << " super-blocks, "
                          << tiled.n_steps << " steps, " << tiled.dense_entries << " entries in tiles + " << tiled.padding
                          << " padding, " << tiled.n_rem << " in the remainder list" << std::endl;
            const char *chk = env_get("HPRLP_TILING_CHECK");
            if (chk && chk[0] == '1' && ci) {
                TiledHost th;
                const bool hok = build_tiled(rows, cols, host_rp(), ci, &th, min_rows, min_dense, rb, tile_cols, rem_cap);
                if (hok != ok) throw std::runtime_error("tiling check: host and device builders disagree on acceptance");
                if (ok) tiled.compare_with(th);
            }
            // Round 5 (form-regret corpus): the PIECE form of a copy that stages only about half of its entries is the worst of both
            // worlds -- every super-block's remainder steps stay with one piece, the partial sums go through memory.  Uniform random
            // 1.2M x 1.2M, 16 per row (51 % in tiles): 0.55 ms per iteration against 0.27 in the all-remainder form; band + 30 % far
            // entries (53 %): 0.39 against 0.28.  Such a copy is handed back as "too few entries in dense tiles": the
            // all-remainder form follows where the matrix is large enough for it (Solver::pb_fallback_wanted), else the stream kernel.
            declined_thin = false;
            if (ok && tiled.view.n_pieces > 0 && !mr && !md && min_dense_override < 0.0 && env_get("HPRLP_PIECES_ANYWAY") == nullptr) {
                const double staged = static_cast<double>(tiled.dense_entries) / std::max(1.0, static_cast<double>(tiled.dense_entries) + static_cast<double>(tiled.n_rem));
                if (staged < kPiecesMinDense) {
                    if (pt.on) std::cerr << "[timing]   piece form with " << staged << " of the entries in staged tiles: declined" << std::endl;
                    tiled = DeviceTiled();
                    declined_sparse = true;
                } else if (entries_per_row < kPiecesThinRows) {
                    // Held-out corpus, round 5: a copy that passes the dense-tile test has its rows' columns close together -- and
                    // with fewer than ten entries per row the stream kernel then beats the PIECE form whether or not an XCD's window
                    // fits its L2 (a piece's cost goes with the tiles it stages): node-arc incidence 1M x 4M after the locality
                    // ordering, 8 / 2 per row: 0.070 / 0.093 ms per half-step in pieces, 0.055 / 0.085 on the stream kernel; 5-, 7-
                    // and 9-point stencils in random order (after the ordering) 5-13 % per iteration; 3M x 3M band of 300 000 columns,
                    // 8 per row: 0.357 -> 0.327 ms (12 per row: pieces stay ahead, 0.275 against 0.293).
                    if (pt.on) std::cerr << "[timing]   piece form with " << entries_per_row << " entries per row: the stream kernel instead" << std::endl;
                    tiled = DeviceTiled();
                    declined_thin = true;
                }
            }
            bool kept = ok && tiled.view.valid;
            if (kept) {
                tiled.build_far(cols, bs, gb);  // consumes the remainder lists the check above compares
                // Second held-out set, round 5: what the tiles could not hold gathers from a FEW popular columns (the first-stage columns of
                // a two-stage stochastic LP: 20 % of the entries, 160 KB of the vector) and the rest of a row from a window that an XCD's L2
                // holds anyway: the stream kernel finds ALL of it in its L2, the tiled form sends the popular fifth through the remainder at
                // 30 bytes per entry.  1M x 1.42M, 8 per row, 2 000 scenario blocks: lowered fused form 0.080 ms per half-step (28 % in
                // the remainder), stream kernel 0.034.  (A band with 30 % uniformly far entries has the same share in the remainder and NO
                // such concentration: the tiled form stays ahead, 0.237 against 0.288 ms per iteration.)
                const bool in_one_l2 = xcd_gather_bytes > 0.0 && xcd_gather_bytes <= kStreamL2Bytes;
                if (in_one_l2 && !mr && !md && min_dense_override < 0.0 && env_get("HPRLP_TILED_ANYWAY") == nullptr && env_get("HPRLP_PIECES_ANYWAY") == nullptr &&
                    static_cast<double>(tiled.n_rem) >= kPopularFarMinRem * static_cast<double>(nnz) && tiled.rem_top_share >= kPopularFarShare) {
                    if (pt.on) std::cerr << "[timing]   " << tiled.rem_top_share << " of the remainder on 2 MB of popular columns, window in one L2: the stream kernel instead" << std::endl;
                    tiled = DeviceTiled();
                    declined_popular = true;
                    kept = false;
                }
            }
            if (kept) {
                view.tiled = tiled.view;
                join_values();
                launch_tiled_refresh(tiled, val.p, bs);
                HIP_CHECK(hipDeviceSynchronize());
            }
            pt.tick("  build tiled copy (device)");
        } else if (rows >= min_rows && rows > 0 && nnz > 0 && ci) {  // the host builder needs the host column indices
            planned_grid = std::max(((rows + rb - 1) / rb + 7) / 8 * 8, (rows + kThreads - 1) / kThreads);  // fused grid or the split form's finish grid
            const int tc = tile_cols, rc = rem_cap;
            const int *rp = host_rp();
            tiling = std::async(std::launch::async, [=]() -> std::shared_ptr<TiledHost> {
                (void)keep;  // keeps the host arrays alive for the duration of the build
                auto th = std::make_shared<TiledHost>();
                return build_tiled(rows, cols, rp, ci, th.get(), min_rows, min_dense, rb, tc, rc) ? th : nullptr;
            });
        }
    }
}

void DeviceMatrix::finish_tiling(hipStream_t s) {
    if (!tiling.valid()) return;
    PhaseTimer pt;
    std::shared_ptr<TiledHost> th = tiling.get();  // rethrows what the job threw
    pt.tick("  wait for the tiled build");
    if (!th) {
        planned_grid = 0;
        return;
    }
    if (pt.on)
        std::cerr << "[timing]   tiled copy: " << th->sb_mid.size() << " super-blocks, " << th->steps.size() << " steps, "
                  << th->dense_entries << " entries in tiles + " << th->padding << " padding, " << th->n_rem
                  << " in the remainder list" << std::endl;
    tiled.upload(*th, sb_rows, tile_cols, rem_cap);
    tiled.build_far(view.cols, s, far_group);
    view.tiled = tiled.view;
    launch_tiled_refresh(tiled, val.p, s);
    HIP_CHECK(hipStreamSynchronize(s));
    pt.tick("  upload tiled copy");
}

void DeviceMatrix::refresh_tiled(hipStream_t s) {
    if (view.tiled.valid) launch_tiled_refresh(tiled, val.p, s);
}

// ------------------------------------------------------------------------------------------------
Solver::~Solver() {
    A.tiled.dump_stamps();
    AT.tiled.dump_stamps();
    A.tiled.dump_wgtimes();
    AT.tiled.dump_wgtimes();
    for (auto &kv : graphs) (void)hipGraphExecDestroy(kv.second);
    if (ev_ready) (void)hipEventDestroy(ev_ready);
    if (ev_done_x) (void)hipEventDestroy(ev_done_x);
    if (ev_done_y) (void)hipEventDestroy(ev_done_y);
    if (comm_stream) (void)hipStreamDestroy(comm_stream);
    if (stream) (void)hipStreamDestroy(stream);
}

void Solver::alloc_work() {
    x.alloc_zero(n_loc); last_x.alloc_zero(n_loc); z_bar.alloc_zero(n_loc);
    last_y.alloc_zero(m_loc); y_obj.alloc_zero(m_loc); y_temp.alloc_zero(m_loc);
    gy.alloc_zero(m_pad); gyb.alloc_zero(m_pad); gsm.alloc_zero(m_pad);
    gxh.alloc_zero(n_pad); gxb.alloc_zero(n_pad); gxt.alloc_zero(n_pad); gsn.alloc_zero(n_pad);
    y = gy.p + row_off; y_bar = gyb.p + row_off;
    x_hat = gxh.p + col_off; x_bar = gxb.p + col_off; x_temp = gxt.p + col_off;
    sm1.alloc_zero(m_loc); sn1.alloc_zero(n_loc);
    row_norm.alloc(m_loc); col_norm.alloc(n_loc);
    ctrl.alloc_zero(1);
    scal.alloc_zero(kNumScalars);
    scal_h.alloc(kNumScalars);
    stride_x = std::max(std::max(AT.view.grid(), AT.planned_grid), 1);  // either kernel's partials fit
    stride_y = std::max(std::max(A.view.grid(), A.planned_grid), 1);
    part_x.alloc_zero(static_cast<size_t>(3) * stride_x);
    part_y.alloc_zero(static_cast<size_t>(2) * stride_y);
    part_r.alloc_zero(static_cast<size_t>(2) * std::max(stride_x, stride_y));
    part_v.alloc_zero(static_cast<size_t>(2) * kReduceBlocks);
    const char *ng = env_get("HPRLP_NO_GRAPH");
    use_graph = !(ng && ng[0] == '1') && comm == nullptr;
    const char *no = env_get("HPRLP_NO_OVERLAP");
    overlap_enabled = comm != nullptr && comm->size > 1 && !(no && no[0] == '1');
    // HPRLP_OVERLAP_COMM_FIRST=1 (tests): take RCCL's launch order with the in-process group too
    overlap_spmv_first = comm != nullptr && comm->host_blocking() && !env_get("HPRLP_OVERLAP_COMM_FIRST");
}

void Solver::setup(const LP_info_cpu *model, const HPRLP_parameters *param) {
    const auto t0 = time_now();
    prm = *param;
    env_at_setup = env_in_effect(&env_ignored_at_setup);
    read_hooks();
    HIP_CHECK(hipSetDevice(prm.device_number));
    HIP_CHECK(hipStreamCreate(&stream));
    m = m_loc = model->m;
    n = n_loc = model->n;
    m_pad = m; n_pad = n;
    row_off = col_off = 0;
    obj_constant = model->obj_constant;
    const sparseMatrix *As = model->A;
    if (!As || As->row != m || As->col != n) throw std::runtime_error("model matrix dimensions inconsistent");
    const long nnz = As->numElements;
    PhaseTimer pt;
    // choose_sb_rows() reads rowPtr / colIndex on the host before upload() has validated them: refuse a malformed row
    // pointer array here (upload() repeats the monotonicity test and checks the column range)
    if (!As->rowPtr || !As->colIndex || !As->value) throw std::runtime_error("model matrix arrays missing");
    if (As->rowPtr[0] != 0 || As->rowPtr[m] != nnz) throw std::runtime_error("row pointer array does not span the nonzeros");
    for (int i = 0; i < m; ++i)
        if (As->rowPtr[i + 1] < As->rowPtr[i]) throw std::runtime_error("row pointer array is not monotone");
    choose_sb_rows(model);
    A.upload(m, n, As->rowPtr, As->colIndex, As->value);
    pt.tick("A upload total");
    // the five model vectors (in the solver's numbering): uploaded beside the device build of A^T's tiled copy when the matrix is
    // large (own copy stream, helper thread), otherwise in line
    std::future<void> vec_job;
    auto upload_vectors = [this, model](bool own_stream) {
        int dev = prm.device_number;
        HIP_CHECK(hipSetDevice(dev));
        hipStream_t cs = nullptr;
        if (own_stream) HIP_CHECK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
        auto upload_vec = [cs, own_stream](DBuf<double> &d, const double *src, int len, const std::vector<int> &perm) {
            std::vector<double> tmp;
            if (!perm.empty()) {
                tmp.resize(static_cast<size_t>(len));
                for (int i = 0; i < len; ++i) tmp[i] = src[perm[i]];
                src = tmp.data();
            }
            if (own_stream) HIP_CHECK(copy_in_pieces(d.p, src, static_cast<size_t>(len) * sizeof(double), hipMemcpyHostToDevice, cs));
            else d.upload(src, len);
        };
        try {
            upload_vec(AL, model->AL, m, perm_r);
            upload_vec(AU, model->AU, m, perm_r);
            upload_vec(l, model->l, n, perm_c);
            upload_vec(u, model->u, n, perm_c);
            upload_vec(c, model->c, n, perm_c);
        } catch (...) {
            if (cs) (void)hipStreamDestroy(cs);
            throw;
        }
        if (cs) (void)hipStreamDestroy(cs);
    };
    AL.alloc(m);
    AU.alloc(m);
    l.alloc(n);
    u.alloc(n);
    c.alloc(n);
    {   // explicit A^T built on the host, stable in row order (reference src/preprocess.cu:78-82); its index
        // arrays are handed to the background job that builds the tiled copy of A^T
        struct HostT {
            std::vector<int> trp, tci;
        };
        auto ht = std::make_shared<HostT>();
        std::vector<int> &trp = ht->trp, &tci = ht->tci;
        const char *hostt = env_get("HPRLP_HOST_TRANSPOSE");
        const char *dmin = env_get("HPRLP_DEVICE_TRANSPOSE_MIN");  // nonzero threshold (tests lower it)
        if (nnz > (dmin ? std::atol(dmin) : 4000000L) && !(hostt && hostt[0] == '1')) {
            // large matrix: transpose on the device (transpose.hip), bring the index arrays back for the host-side
            // consumers (tiled build, row statistics)
            AT.rowptr.alloc(static_cast<size_t>(n) + 1);
            AT.col.alloc(static_cast<size_t>(nnz));
            AT.val.alloc(static_cast<size_t>(nnz));
            device_transpose(m, n, nnz, A.rowptr.p, A.col.p, A.val.p, AT.rowptr.p, AT.col.p, AT.val.p, stream);
            pt.tick("device transpose");
            const bool reordered = try_reorder(model);
            if (reordered) {  // A now holds P A Q: transpose that
                device_transpose(m, n, nnz, A.rowptr.p, A.col.p, A.val.p, AT.rowptr.p, AT.col.p, AT.val.p, stream);
                pt.tick("locality ordering + device transpose of the permuted matrix");
            }
            // (also behind an ordering whose copy the build then declined -- the cheap tiling test had accepted it: covering pattern
            // 300k x 3M, four per column: stream kernel 0.188 ms on the y-half where the remainder path takes 0.133; end of round 5)
            if (!A.view.tiled.valid && pb_fallback_wanted(A, AT.rowptr.p, n)) {
                choose_pb_rows(A, AT, m, n);
                // no column locality to be had (or the ordering is disabled): every random 8-byte gather of the stream kernel
                // would cost a 128-byte line from the Infinity Cache / HBM.  Accept the tiled form with NO dense-tile
                // requirement: whatever is not in dense tiles -- here almost everything -- goes through the propagation-
                // blocking remainder (pre-pass in source order, streamed products in destination order): about 30 bytes per
                // nonzero, all sequential.  Unstructured 2e8-nnz matrix: 3.85 -> 1.87 ms per half-step.
                if (reordered) {
                    std::vector<int> hrp(static_cast<size_t>(m) + 1);
                    A.rowptr.download(hrp.data(), hrp.size());
                    A.describe(m, n, hrp.data(), nullptr, nullptr, 0.0);
                } else {
                    A.describe(m, n, As->rowPtr, nullptr, nullptr, 0.0);
                }
                pt.tick("tiled copy of A without a dense-tile requirement (propagation blocking for all entries)");
            }
            // the host copy of A^T's row pointers (row blocks, statistics) comes back on a copy stream of its own, driven by a
            // thread, while the tiled copy of A^T is built from the device arrays (describe_when)
            int dev = 0;
            HIP_CHECK(hipGetDevice(&dev));
            const bool overlap_setup = env_get("HPRLP_NO_SETUP_OVERLAP") == nullptr;
            const int *d_trp = AT.rowptr.p;
            const int n_rows_t = n;
            std::shared_future<const int *> trp_ready =
                std::async(overlap_setup ? std::launch::async : std::launch::deferred, [ht, dev, d_trp, n_rows_t]() -> const int * {
                    HIP_CHECK(hipSetDevice(dev));
                    ht->trp.resize(static_cast<size_t>(n_rows_t) + 1);
                    hipStream_t cs = nullptr;
                    HIP_CHECK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
                    const hipError_t e = copy_in_pieces(ht->trp.data(), d_trp, ht->trp.size() * sizeof(int), hipMemcpyDeviceToHost, cs);
                    (void)hipStreamDestroy(cs);
                    HIP_CHECK(e);
                    return ht->trp.data();
                }).share();
            // the column indices of A^T are only needed on the host by the host tiled builder (or its check)
            const char *htile = env_get("HPRLP_HOST_TILING"), *chk = env_get("HPRLP_TILING_CHECK");
            const bool need_tci = (htile && htile[0] == '1') || (chk && chk[0] == '1');
            if (need_tci) {
                tci.resize(static_cast<size_t>(nnz));
                AT.col.download(tci.data(), tci.size());
            }
            if (overlap_setup) vec_job = std::async(std::launch::async, upload_vectors, true);  // (the numbering is final here)
            AT.describe_when(n, m, nnz, trp_ready, need_tci ? tci.data() : nullptr, ht, -1.0, nullptr);
            trp_ready.wait();
            if (pb_fallback_wanted(AT, A.rowptr.p, m)) {
                choose_pb_rows(AT, A, n, m);
                AT.describe(n, m, trp.data(), nullptr, nullptr, 0.0);
            }
        } else {
            std::vector<double> tv;
            csr_transpose_host(m, n, nnz, As->rowPtr, As->colIndex, As->value, trp, tci, tv);
            pt.tick("host transpose");
            AT.upload(n, m, trp.data(), tci.data(), tv.data(), ht);
        }
        pt.tick("A^T upload total");
        max_row_A = std::max(max_row_A, A.longest_row);  // (describe_when; a renumbering does not change the longest row)
        max_row_AT = std::max(max_row_AT, AT.longest_row);
        const char *ns = env_get("HPRLP_NO_SMALL");
        use_small = !(ns && ns[0] == '1') && small_path_fits(m, n, nnz, max_row_A, max_row_AT);
        if (use_small) {
            auto by_length = [](int rows, const int *rp, DBuf<int> &out) {
                std::vector<int> ord(rows);
                for (int i = 0; i < rows; ++i) ord[i] = i;
                std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return rp[a + 1] - rp[a] > rp[b + 1] - rp[b]; });
                out.alloc(rows);
                out.upload(ord.data(), rows);
            };
            by_length(n, trp.data(), small_order_x);
            by_length(m, As->rowPtr, small_order_y);
            // the transpose is stable in row order: the k-th entry of A (row i, column j) is the next free
            // slot of row j of A^T
            std::vector<int> next(trp.begin(), trp.end() - 1), ij(static_cast<size_t>(nnz)), posA(static_cast<size_t>(nnz));
            for (int i = 0; i < m; ++i)
                for (int k = As->rowPtr[i]; k < As->rowPtr[i + 1]; ++k) {
                    const int j = As->colIndex[k], q = next[j]++;
                    if (tci[q] != i) throw std::runtime_error("small path: transpose is not stable");
                    ij[q] = i | (j << 16);
                    posA[q] = k;
                }
            small_ij.alloc(ij.size());
            small_ij.upload(ij.data(), ij.size());
            small_posA.alloc(posA.size());
            small_posA.upload(posA.data(), posA.size());
        }
    }
    if (vec_job.valid()) vec_job.get();
    else upload_vectors(false);
    alloc_work();
    // the tiled copy of A was built from the model's own arrays, which the caller may free once we return; the
    // job of A^T owns its arrays and keeps running under scale()
    A.finish_tiling(stream);
    pt.tick("vectors, work space, tiled copy of A");
    HIP_CHECK(hipDeviceSynchronize());
    setup_time = time_since(t0);
    if (pt.on) {
        const AllocStats &as = AllocStats::get();
        std::cerr << "[timing] allocator so far (process): " << as.mallocs << " hipMalloc " << as.malloc_s << " s, " << as.frees << " hipFree "
                  << as.free_s << " s" << std::endl;
    }
}

// Gather vector too long for the L2s (>= 4 M entries = 32 MB) and the tiled build declined for lack of dense tiles: the
// stream kernel would pay a fabric line per gathered element (HPRLP_NO_PB_FALLBACK=1 keeps it anyway; one GPU only).
// gathered vector from which the all-remainder tiled form beats the stream kernel on a pattern without locality (measured,
// tools/unstructured_ab.py, uniformly random 10 per row: 1M columns 0.154 vs 0.125 ms per half-step, 2M 0.218 vs 0.318, 3M 0.316
// vs 0.508, 4.2M 0.46 vs 0.75, 6M 0.59 vs 1.13)
constexpr double kMaxTileShare = 0.6;  // choose_sb_rows: most tile bytes per entry byte a lowered super-block may stage (one round)
constexpr double kMaxTileShareRounds = 0.9;  // ... when the height only trims a partial last round of a larger matrix
constexpr long kPbMinCols = 800000;  // (round 4, tools/unstructured_ab.py with k_pb_fused, 10 per row: 0.5M columns 0.065 vs 0.041 ms stream, 1.0M 0.080 vs 0.123, 1.5M 0.119 vs 0.214: from where the vector outgrows a 4 MiB L2)
constexpr int kPbFewRowsMin = 32768;   // pb_fallback_wanted: fewest rows of a matrix that takes the all-remainder form without having been through the tiled build
constexpr long kPopularLines = 32768;      // pb_fallback_wanted: 2 MB of the gathered vector ...
constexpr double kPopularShareMax = 0.3;   // ... that may not take more than this share of a few-row matrix' gathers
constexpr double kPbHeaviestBlockShare = 48.0;  // pb_fallback_wanted: a matrix with long rows takes the all-remainder form only if its heaviest 4096-row block holds at most 1 / 48 of the entries
constexpr int kPbFewRowsLow = 80000;    // ... half that height below this many rows
constexpr int kPbFewRowsHeight = 512;  // choose_pb_rows: super-block height for such a matrix (below 32 full-height super-blocks' worth of rows)
constexpr double kNarrowTilesFrom = 1.2;  // choose_sb_rows: entries of a row per 2048-column tile from which the copy gets 1024-column tiles

// Share of a matrix' entries whose gathers go to the `lines` most popular 64-byte lines of the gathered vector (columns 8 l .. 8 l + 7;
// `other` is the transpose: its row lengths are the column counts).  What the stream kernel's gathers find in an L2 however far apart
// the rows reach.
static double popular_lines_share(const int *d_rowptr, int cols, long nnz, long lines) {
    if (cols <= 0 || nnz <= 0 || !d_rowptr) return 0.0;
    std::vector<int> rp(static_cast<size_t>(cols) + 1);
    HIP_CHECK(hipMemcpy(rp.data(), d_rowptr, rp.size() * sizeof(int), hipMemcpyDeviceToHost));
    const long nl = (static_cast<long>(cols) + 7) / 8;
    if (nl <= lines) return 1.0;
    std::vector<int> deg(static_cast<size_t>(nl));
    for (long l = 0; l < nl; ++l) deg[l] = rp[std::min<long>(8 * (l + 1), cols)] - rp[8 * l];
    std::nth_element(deg.begin(), deg.begin() + lines, deg.end(), std::greater<int>());
    long top = 0;
    for (long l = 0; l < lines; ++l) top += deg[l];
    return static_cast<double>(top) / static_cast<double>(nnz);
}

bool Solver::pb_fallback_wanted(const DeviceMatrix &M, const int *other_rowptr, int other_rows) const {
    const char *no = env_get("HPRLP_NO_PB_FALLBACK");
    if (no && no[0] == '1') return false;
    const char *nt = env_get("HPRLP_NO_TILED");
    if (nt && nt[0] == '1') return false;
    const long min_cols = env_get("HPRLP_PB_MIN_COLS") ? std::atol(env_get("HPRLP_PB_MIN_COLS")) : kPbMinCols;
    const long min_nnz = env_get("HPRLP_PB_MIN_NNZ") ? std::atol(env_get("HPRLP_PB_MIN_NNZ")) : 4000000L;  // (tests lower it)
    // a pattern whose rows stay near a diagonal keeps the stream kernel: each XCD's eighth of the rows gathers from a window of the
    // vector that its L2 holds (Solver::choose_sb_rows: xcd_gather_bytes; 0 = not estimated).  1M x 1M, band 2000, 20 per row (the
    // tiled build declines it: too many entries of a row per tile): stream 0.107 ms per half-step, all-remainder form 0.149.
    const bool in_l2 = M.xcd_gather_bytes > 0.0 && M.xcd_gather_bytes <= kStreamL2Bytes && env_get("HPRLP_PB_MIN_COLS") == nullptr;
    // Round 5, held-out corpus (tools/form_regret.py --corpus held_out): a matrix with FEWER rows than the staged forms ask for
    // (a super-block per CU) never reached the tiled build, so it never got here either -- and kept the stream kernel at 0.11 of
    // 8 TB/s where its rows gather at random from millions of columns: 200k x 5M with 75 per row (the transpose of a 3-per-row
    // matrix), x-half 0.262 ms against 0.125 here (pre-pass 0.073 + k_pb_fused 0.049, super-blocks of 512 rows); 100k x 5M with
    // 150 per row: 0.262 against 0.140.  Taken where the rows do NOT share their lines (launch_line_density).
    // ... and where a ROW's own column window is beyond an L2 (xcd_gather_bytes less the drift of the eighth along the diagonal =
    // the median row span): 150k x 3M with 60 per row inside a window of 150 000 columns has every entry on a line of its own and
    // still gathers out of 1.2 MB -- stream kernel 0.063 ms, all-remainder form 0.081.
    const double row_window_bytes = M.xcd_gather_bytes > 0.0 ? M.xcd_gather_bytes - static_cast<double>(M.view.cols) : 0.0;
    bool few_rows = M.declined_few_rows && M.view.rows >= kPbFewRowsMin && M.line_density >= kStreamL2LineDensity &&
                    (M.xcd_gather_bytes <= 0.0 || row_window_bytes > kStreamL2Bytes);
    const bool size_ok = !comm && !M.view.tiled.valid && !in_l2 && M.view.cols >= min_cols && M.view.nnz >= min_nnz;
    // Validation set, end of round 5: a matrix kept off the tiled forms for its LONG rows (hubs of a b-matching LP: rows of up to 77 000
    // entries, a quarter of the entries in rows over 1 024) whose columns are NOT popular gathers at random like any unstructured
    // matrix -- stream kernel 0.12 of 8 TB/s.  k_pb_fused adds rows of any length; what it cannot take is a super-block far heavier
    // than the chip's share (the launch ends with it).  (A Kronecker graph has popular columns: it keeps the stream kernel, rule 1.)
    bool long_rows = false;
    if (M.declined_long_rows && size_ok && !M.declined_sparse && !few_rows && M.line_density >= kStreamL2LineDensity && env_get("HPRLP_NO_PB_LONG_ROWS") == nullptr) {
        const double share = popular_lines_share(other_rowptr, other_rows, M.view.nnz, kPopularLines);
        const int heaviest = launch_heaviest_block(M.rowptr.p, M.view.rows, kPbRowsMax, stream);
        if (env_get("HPRLP_TIMING"))
            std::cerr << "[timing] long rows: share of the entries on the " << kPopularLines << " most popular lines " << share << ", heaviest block of " << kPbRowsMax
                      << " rows " << heaviest << " entries of " << M.view.nnz << std::endl;
        long_rows = share <= kPopularShareMax && static_cast<double>(heaviest) * kPbHeaviestBlockShare <= static_cast<double>(M.view.nnz);
    }
    if (long_rows) return true;
    if (few_rows && size_ok && !M.declined_sparse) {
        // ... and where no small set of popular columns takes a large share of the gathers (they stay in the L2s whatever the rows'
        // reach): set-covering pattern 200k x 2M, 50 per row, column popularity ~ c^-0.6 -- 44 % of the entries on the 32 768 most
        // popular lines (2 MB): stream kernel 0.123 ms, all-remainder form 0.144 (uniform columns: 5 %).
        const double share = popular_lines_share(other_rowptr, other_rows, M.view.nnz, kPopularLines);
        if (env_get("HPRLP_TIMING")) std::cerr << "[timing] few rows: share of the entries on the " << kPopularLines << " most popular lines " << share << std::endl;
        if (share > kPopularShareMax) few_rows = false;
    }
    return size_ok && (M.declined_sparse || few_rows);
}

// Super-block heights of this LP's tiled copies (tiled.h).  A matrix with fewer than 512 full-height super-blocks cannot give
// every workgroup slot of the chip a whole super-block: it ran the piece form (three launches, partial sums through memory) or,
// below 2^20 columns, the stream kernel.  With R = rows / 512 every slot gets exactly one, the epilogue stays fused, a half-step
// is one launch and there is no tail.  What a lower super-block costs is tile traffic -- a staged tile serves R rows -- so the
// column window of a super-block must stay narrow against its entries: estimated from the column span of the middle three quarters of the
// entries of 2048 sampled rows (the far entries of a band matrix do not count: they go through the remainder lists).  A
// source group of one matrix' remainder lists is a super-block of the other (hand-off, kernels.h FarPush): far_group of A is
// sb_rows of A^T and vice versa.  Same-box A/B (profiles/r03_ab_rows*.txt): 1M x 1M, band 1e4: 3658 it/s stream kernel, 3393
// pieces, 5287 with 2048-row super-blocks; the 1.25M x 10M shard of config 5 (window of 2e5 columns): a loss, declined here.
// The height that fills exactly k rounds of the chip's workgroup slots, k = the rounds the FULL height needs (k = 1: one
// super-block per slot); the full height where its rounds are nearly full already.
static int whole_rounds_height(int rows, int slots) {
    const int nsb_full = (rows + kTileRows - 1) / kTileRows;
    const int k = std::max(1, (nsb_full + slots - 1) / slots);
    if (k > 1 && static_cast<double>(nsb_full) / (static_cast<double>(k) * slots) >= 0.8) return kTileRows;  // rounds nearly full already
    const int per = (rows + k * slots - 1) / (k * slots);
    return std::min(kTileRows, (per + 63) / 64 * 64);
}

// Heights for a matrix that runs the tiled form WITHOUT staged tiles (pb_fallback_wanted: every entry through the
// propagation-blocking remainder).  No tile is staged, so a lower super-block costs nothing in tile traffic: take the height
// that gives every workgroup slot whole super-blocks -- the half-step is then ONE fused launch whose epilogue hands the
// products over to the other half (round 3 ran such matrices at full height: 245 super-blocks of a 2M x 2M matrix = the piece
// form, three launches per half-step, partial sums through memory, no hand-off).  HPRLP_TILE_ROWS still overrides.
void Solver::choose_pb_rows(DeviceMatrix &M, DeviceMatrix &other, int rows, int other_rows) {
    M.rem_cap = kTileRemCap;
    if (comm) return;
    if (!env_get("HPRLP_TILE_ROWS")) {
        const int slots = workgroup_slots();
        // at most kPbRowsMax rows (the all-remainder kernel's accumulators, kernels.hip: k_pb_fused): larger matrices take more rounds
        auto height = [&](int nrows) {
            // (few rows: 512-row super-blocks measured best -- 200k rows: 256 / 384 / 512 / 1024 rows 0.166 / 0.148 / 0.125 / 0.142 ms,
            // 100k rows: 0.160 / 0.153 / 0.140 / 0.193)
            // (50k rows: 256 / 512 rows 0.170 / 0.182; 33k rows: 0.155 / 0.190)
            if (nrows < 32 * kTileRows) return nrows < kPbFewRowsLow ? kPbFewRowsHeight / 2 : kPbFewRowsHeight;
            int r = std::max(kTileRowsMin, whole_rounds_height(nrows, slots));
            for (int k = 2; r > kPbRowsMax; ++k) r = std::max(kTileRowsMin, ((nrows + k * slots - 1) / (k * slots) + 63) / 64 * 64);
            return r;
        };
        M.sb_rows = other.far_group = height(rows);
        // the other matrix has the same graph: if it is not described yet, expect it to take the same form (its source groups are
        // this matrix' hand-off unit, kernels.h FarPush; a wrong guess only costs the hand-off)
        if (!other.view.tiled.valid) other.sb_rows = M.far_group = height(other_rows);
    }
    // (HPRLP_NO_PB_KERNEL, A/B runs: the all-remainder copy through k_tiled_fused's remainder steps, as in round 3)
    if (M.sb_rows <= kPbRowsMax && !env_get("HPRLP_NO_PB_KERNEL")) M.rem_cap = kPbRemCap;
    if (env_get("HPRLP_TIMING"))
        std::cerr << "[timing] no column locality: all-remainder form with super-blocks of " << M.sb_rows << " rows, remainder steps of " << M.rem_cap
                  << " entries" << std::endl;
}

void Solver::choose_sb_rows(const LP_info_cpu *model) {
    A.sb_rows = A.far_group = AT.sb_rows = AT.far_group = kTileRows;
    A.tile_cols = AT.tile_cols = kTileCols;
    A.xcd_gather_bytes = AT.xcd_gather_bytes = 0.0;
    if (const char *force = env_get("HPRLP_TILE_COLS")) {  // tests / A/B runs: one tile width for both matrices
        A.tile_cols = AT.tile_cols = std::atoi(force) <= kTileColsNarrow ? kTileColsNarrow : kTileCols;
    }
    if (const char *force = env_get("HPRLP_TILE_ROWS")) {  // tests / A/B runs: one height for both matrices
        const int R = std::max(64, std::min(kTileRows, std::atoi(force) / 64 * 64));
        A.sb_rows = A.far_group = AT.sb_rows = AT.far_group = R;
        return;
    }
    if (comm) return;  // row shards: all columns of the LP against 1 / P of the rows -- full height
    const sparseMatrix *As = model->A;
    const long nnz = As->numElements;
    if (nnz < 4000000) return;
    const int slots = workgroup_slots();
    // median column span of a row without its outermost entries
    std::vector<long> span;
    const int samples = 2048;
    for (int q = 0; q < samples; ++q) {
        const int i = static_cast<int>(static_cast<long>(q) * m / samples);
        const int b = As->rowPtr[i], e = As->rowPtr[i + 1], len = e - b;
        if (len < 4) continue;
        // an eighth of the entries off either end (at least one: a 16-entry row of config 5's kind carries one far entry),
        // the span of the rest scaled back to the whole row
        // (rows need not be sorted by column: the span is |difference|, an unsorted row only makes the estimate coarser,
        // never negative -- a negative span used to pass the `ratio <= most` test below)
        const int cut = std::max(1, len / 8);
        const long ia = std::min<long>(std::max<long>(static_cast<long>(e) - 1 - cut, b), nnz - 1), ib = std::min<long>(static_cast<long>(b) + cut, nnz - 1);
        const long inner = std::labs(static_cast<long>(As->colIndex[ia]) - As->colIndex[ib]);
        span.push_back(inner * (len - 1) / std::max(1, len - 1 - 2 * cut));
    }
    if (span.size() < 16) return;
    std::nth_element(span.begin(), span.begin() + span.size() / 2, span.end());
    const double w_a = static_cast<double>(std::max<long>(span[span.size() / 2], 1));
    const double slope = static_cast<double>(n) / m;  // columns per row along the "diagonal"
    // what one XCD's eighth of the rows gathers from (stream kernel; DeviceMatrix::build_tiled_copy weighs it against the piece form)
    A.xcd_gather_bytes = (w_a + m / 8.0 * slope) * 8.0;
    AT.xcd_gather_bytes = (w_a / slope + n / 8.0 / slope) * 8.0;
    // Tile width (round 4).  A row segment of more than kTileChunk entries in one tile goes to the remainder lists WHOLE (34
    // bytes of traffic per entry against 11 in a tile).  With d entries per row spread over a window of w columns a tile of T
    // columns holds d T / w of them on average; from about 1.2 on, segments of five and more are common (1M x 1M, band 1e4,
    // d = 19: 1.95 per 2048-column tile, 10.5 % of the entries in such segments; 1024 columns: 1.2 %).  Narrow tiles halve
    // the staged bytes per step and leave the number of steps about the same (the wide tiles of such a matrix take two).
    if (!env_get("HPRLP_TILE_COLS")) {
        const double per_tile_a = static_cast<double>(nnz) / m * kTileCols / w_a;
        const double per_tile_at = static_cast<double>(nnz) / n * kTileCols / std::max(w_a / slope, 1.0);
        if (per_tile_a > kNarrowTilesFrom) A.tile_cols = kTileColsNarrow;
        if (per_tile_at > kNarrowTilesFrom) AT.tile_cols = kTileColsNarrow;
        if (env_get("HPRLP_TIMING"))
            std::cerr << "[timing] entries of a row per 2048-column tile: " << per_tile_a << " (A), " << per_tile_at << " (A^T) -> tiles of "
                      << A.tile_cols << " / " << AT.tile_cols << " columns" << std::endl;
    }
    // Heights considered for a matrix of `rows` rows: with k = the rounds the FULL height needs (ceil of its super-blocks over the
    // slots), the height that fills exactly k rounds.  k = 1: one super-block per slot (mid-size matrices).  k >= 2: the same
    // number of rounds as now without the partial last one -- only when the full height wastes more than a fifth of its rounds
    // (6M x 6M, band 6e4: 733 super-blocks = 1.43 rounds run as 2; 1020 of 5888 rows: 1121 -> 1148 it/s; 5M x 5M: 1264 -> 1321;
    // profiles/r03_ab_rows7.txt).  Nothing to gain below kTileRowsMin (launch-bound matrices: the stream kernel).
    const int ra = whole_rounds_height(m, slots), rat = whole_rounds_height(n, slots);
    if ((ra >= kTileRows && rat >= kTileRows) || ra < kTileRowsMin || rat < kTileRowsMin) return;
    // bytes of the vector tiles a super-block stages against the bytes of its entries; a height that only trims a partial round
    // may stage a little more (it saves a fifth of the rounds or more)
    const double ratio_a = (w_a + ra * slope) * 8.0 / (static_cast<double>(nnz) / m * ra * 11.0);
    const double ratio_at = (w_a / slope + rat / slope) * 8.0 / (static_cast<double>(nnz) / n * rat * 11.0);
    const bool multi = static_cast<long>(ra) * slots < m || static_cast<long>(rat) * slots < n;  // more than one round
    const double most = multi ? kMaxTileShareRounds : kMaxTileShare;
    const bool ok = ratio_a <= most && ratio_at <= most;
    if (ok) {
        A.sb_rows = AT.far_group = ra;
        AT.sb_rows = A.far_group = rat;
    }
    if (env_get("HPRLP_TIMING"))
        std::cerr << "[timing] super-block heights for whole rounds of " << slots << " slots: " << ra << " (A), " << rat << " (A^T); median row span " << w_a
                  << " columns, tile bytes / entry bytes " << ratio_a << ", " << ratio_at << " -> " << (ok ? "lowered" : "full height (8192)") << std::endl;
}

// Large matrix whose given order failed the tiling test: look for a locality ordering (reorder.cpp).  On entry A (device
// CSR of the model) and the arrays of AT (device transpose, not yet described) are in place.  On success A's device
// arrays hold P A Q, A is re-described (row blocks, tiled copy) and perm_r / perm_c are set.
bool Solver::try_reorder(const LP_info_cpu *model) {
    if (!allow_reorder) return false;
    const char *no = env_get("HPRLP_NO_REORDER");
    if (no && no[0] == '1') return false;
    const char *nt = env_get("HPRLP_NO_TILED");
    if (nt && nt[0] == '1') return false;
    const char *mr = env_get("HPRLP_TILED_MIN_ROWS");
    const int min_rows = mr ? std::atoi(mr) : 32 * kTileRows;
    if (comm || A.view.tiled.valid || A.declined_shape || m < min_rows || n < min_rows) return false;
    const auto t0 = time_now();
    const sparseMatrix *As = model->A;
    const long nnz = As->numElements;
    ReorderStats st;
    // everything but the spectral ordering of the (small) cluster graph runs on the device, on the patterns of A and A^T
    // that are resident already; HPRLP_REORDER_HOST=1 keeps the clustering on the host (reorder.cpp, the reference form)
    const char *hostc = env_get("HPRLP_REORDER_HOST");
    const bool host_clusters = hostc && hostc[0] == '1';
    const bool timing = env_get("HPRLP_TIMING") != nullptr;
    auto tphase = time_now();
    auto tick = [&](const char *what) {
        if (timing) std::cerr << "[timing]   reorder " << what << " " << time_since(tphase) << " s" << std::endl;
        tphase = time_now();
    };
    st.fraction_before = device_tiling_dense_fraction(m, n, nnz, A.rowptr.p, A.col.p, nullptr, nullptr, stream);
    reorder_before = st.fraction_before;
    tick("tiling test (given order)");
    if (st.fraction_before >= 0.5) return false;  // the tiled build declined for another reason
    DBuf<double> dpr(static_cast<size_t>(m)), dpc(static_cast<size_t>(n));
    if (host_clusters) {
        std::vector<int> trp(static_cast<size_t>(n) + 1), tci(static_cast<size_t>(nnz));
        AT.rowptr.download(trp.data(), trp.size());
        AT.col.download(tci.data(), tci.size());
        std::vector<double> pr, pc;
        cluster_positions(m, n, As->rowPtr, As->colIndex, trp.data(), tci.data(), &pr, &pc, &st);
        dpr.upload(pr.data(), pr.size());
        dpc.upload(pc.data(), pc.size());
    } else {
        device_cluster_positions(m, n, nnz, A.rowptr.p, A.col.p, AT.rowptr.p, AT.col.p, dpr.p, dpc.p, &st, stream);
    }
    tick("clusters and their order");
    DBuf<int> d_r(static_cast<size_t>(m)), d_c(static_cast<size_t>(n));
    device_refine_order(m, n, A.rowptr.p, A.col.p, AT.rowptr.p, AT.col.p, dpr.p, dpc.p, kReorderSweeps, d_r.p, d_c.p, stream);
    tick("median sweeps");
    st.fraction_after = device_tiling_dense_fraction(m, n, nnz, A.rowptr.p, A.col.p, d_r.p, d_c.p, stream);
    reorder_after = st.fraction_after;
    tick("tiling test (permuted)");
    if (verbose || env_get("HPRLP_TIMING"))
        std::cerr << "[reorder] tiled share of the entries " << st.fraction_before << " -> " << st.fraction_after << " (" << st.clusters
                  << " clusters, " << st.components << " components, " << st.bfs_levels << " BFS levels, " << time_since(t0) << " s)" << std::endl;
    if (st.fraction_after < 0.5) {
        reorder_time = time_since(t0);
        return false;
    }
    std::vector<int> hr(static_cast<size_t>(m)), hc(static_cast<size_t>(n));
    d_r.download(hr.data(), hr.size());
    d_c.download(hc.data(), hc.size());
    // P A Q on the device, swapped into A
    DBuf<int> nrp(static_cast<size_t>(m) + 1), nci(static_cast<size_t>(nnz));
    DBuf<double> nval(static_cast<size_t>(nnz));
    device_permute_csr(m, n, nnz, A.rowptr.p, A.col.p, A.val.p, d_r.p, d_c.p, nrp.p, nci.p, nval.p, stream);
    A.rowptr = std::move(nrp);
    A.col = std::move(nci);
    A.val = std::move(nval);
    std::vector<int> hrp(static_cast<size_t>(m) + 1);
    A.rowptr.download(hrp.data(), hrp.size());
    A.describe(m, n, hrp.data(), nullptr, nullptr);
    perm_r = std::move(hr);
    perm_c = std::move(hc);
    reorder_time = time_since(t0);
    return true;
}

void Solver::finish_tiling() {
    if (A.tiling.valid() || AT.tiling.valid()) {
        invalidate_far();
        // captured graphs (normal iterations, the power iteration's block of ten) bake in the kernel form of the views:
        // a view that is about to change from the stream kernel to its tiled copy makes them stale (slower, not wrong)
        for (auto &kv : graphs) (void)hipGraphExecDestroy(kv.second);
        graphs.clear();
    }
    if (A.tiling.valid()) A.finish_tiling(stream);
    if (AT.tiling.valid()) AT.finish_tiling(stream);
}

void Solver::setup_shard(int m_glob, int n_glob, int row_off_, int m_loc_, int col_off_, int n_loc_, const int *Arp,
                         const int *Aci, const double *Av, const int *ATrp, const int *ATci, const double *ATv,
                         const double *AL_, const double *AU_, const double *l_, const double *u_, const double *c_,
                         double obj_constant_, const HPRLP_parameters *param, Comm *comm_) {
    const auto t0 = time_now();
    prm = *param;
    comm = comm_;
    env_at_setup = env_in_effect(&env_ignored_at_setup);
    read_hooks();
    HIP_CHECK(hipSetDevice(prm.device_number));
    HIP_CHECK(hipStreamCreate(&stream));
    m = m_glob; n = n_glob;
    m_loc = m_loc_; n_loc = n_loc_;
    row_off = row_off_; col_off = col_off_;
    const int P = comm ? comm->size : 1;
    const int chunk_m = (m + P - 1) / P, chunk_n = (n + P - 1) / P;
    m_pad = chunk_m * P; n_pad = chunk_n * P;
    if (comm) {
        if (row_off != comm->rank * chunk_m || col_off != comm->rank * chunk_n ||
            m_loc != std::max(0, std::min(m, row_off + chunk_m) - row_off) ||
            n_loc != std::max(0, std::min(n, col_off + chunk_n) - col_off))
            throw std::runtime_error("shard bounds do not match the block partition of hprlp_partition()");
    }
    obj_constant = obj_constant_;
    A.upload(m_loc, n, Arp, Aci, Av);
    AT.upload(n_loc, m, ATrp, ATci, ATv);
    AL.alloc(m_loc); AL.upload(AL_, m_loc);
    AU.alloc(m_loc); AU.upload(AU_, m_loc);
    l.alloc(n_loc); l.upload(l_, n_loc);
    u.alloc(n_loc); u.upload(u_, n_loc);
    c.alloc(n_loc); c.upload(c_, n_loc);
    alloc_work();
    if (comm) {
        // length-m vectors are read through the columns of the A^T shard, length-n ones through those of A
        halo_m.build(comm, ATci, ATrp[n_loc], m, chunk_m, stream);
        halo_n.build(comm, Aci, Arp[m_loc], n, chunk_n, stream);
        verify_exchange();
    }
    finish_tiling();  // the shard arrays belong to the caller
    HIP_CHECK(hipDeviceSynchronize());
    setup_time = time_since(t0);
}

// ------------------------------------------------------------------------------------------------
// collectives (no-ops on one GPU)
// ------------------------------------------------------------------------------------------------
void HaloPlan::build(Comm *comm, const int *cols, long nnz, int total, int chunk, hipStream_t s) {
    const int P = comm->size, r = comm->rank;
    sparse = false;
    if (P <= 1) return;
    // entries of the gathered vector that this shard's column indices name, by owning rank
    std::vector<char> need(static_cast<size_t>(total), 0);
    for (long k = 0; k < nnz; ++k) need[cols[k]] = 1;
    std::vector<int> want;            // concatenated over peers, ascending inside a peer
    std::vector<long> want_off(P + 1, 0);
    for (int p = 0; p < P; ++p) {
        if (p != r) {
            const int lo = std::min(total, p * chunk), hi = std::min(total, (p + 1) * chunk);
            for (int j = lo; j < hi; ++j)
                if (need[j]) want.push_back(j);
        }
        want_off[p + 1] = static_cast<long>(want.size());
    }
    // request counts of everybody: cnt[a*P + b] = entries rank a wants from rank b
    std::vector<double> cnt(static_cast<size_t>(P) * P, 0.0);
    for (int p = 0; p < P; ++p) cnt[static_cast<size_t>(r) * P + p] = static_cast<double>(want_off[p + 1] - want_off[p]);
    DBuf<double> dcnt(static_cast<size_t>(P) * P);
    dcnt.upload(cnt.data(), cnt.size());
    comm->allgather_inplace(dcnt.p, static_cast<size_t>(P), s);
    HIP_CHECK(hipStreamSynchronize(s));
    dcnt.download(cnt.data(), cnt.size());
    total_requests = 0;
    for (double v : cnt) total_requests += static_cast<long>(v);
    const double dense = static_cast<double>(P - 1) * static_cast<double>(total);
    sparse = static_cast<double>(total_requests) <= 0.5 * dense;
    if (const char *e = env_get("HPRLP_DIST_EXCHANGE")) {  // "sparse" / "allgather": same value on every rank
        if (e[0] == 's') sparse = true;
        if (e[0] == 'a') sparse = false;
    }
    if (!sparse) return;

    nrecv = static_cast<int>(want.size());
    recv_idx.alloc(want.size());
    recv_idx.upload(want.data(), want.size());
    std::vector<long> send_off(P + 1, 0);
    for (int p = 0; p < P; ++p) send_off[p + 1] = send_off[p] + static_cast<long>(cnt[static_cast<size_t>(p) * P + r]);
    nsend = static_cast<int>(send_off[P]);
    send_idx.alloc(static_cast<size_t>(nsend));
    // tell every peer which of its entries we read (our request list becomes its send list)
    std::vector<P2P> idx_ops;
    for (int p = 0; p < P; ++p) {
        if (p == r) continue;
        const size_t nw = static_cast<size_t>(want_off[p + 1] - want_off[p]), ns = static_cast<size_t>(send_off[p + 1] - send_off[p]);
        if (nw || ns) idx_ops.push_back(P2P{p, recv_idx.p + want_off[p], nw * sizeof(int), send_idx.p + send_off[p], ns * sizeof(int)});
    }
    comm->exchange(idx_ops.data(), static_cast<int>(idx_ops.size()), s);
    HIP_CHECK(hipStreamSynchronize(s));
    {   // the kernels index without bounds checks: every requested entry must lie in our own slice
        std::vector<int> chk(static_cast<size_t>(nsend));
        send_idx.download(chk.data(), chk.size());
        const int lo = std::min(total, r * chunk), hi = std::min(total, (r + 1) * chunk);
        for (int v : chk)
            if (v < lo || v >= hi) throw std::runtime_error("halo plan: a peer requested an entry this rank does not own");
    }
    sendbuf.alloc_zero(static_cast<size_t>(nsend));
    recvbuf.alloc_zero(static_cast<size_t>(nrecv));
    ops.clear();
    for (int p = 0; p < P; ++p) {
        if (p == r) continue;
        const size_t nw = static_cast<size_t>(want_off[p + 1] - want_off[p]), ns = static_cast<size_t>(send_off[p + 1] - send_off[p]);
        if (nw || ns) ops.push_back(P2P{p, sendbuf.p + send_off[p], ns * sizeof(double), recvbuf.p + want_off[p], nw * sizeof(double)});
    }
}

void Solver::gather_on(double *gbuf, bool is_m, hipStream_t s) {
    if (!comm) return;  // (a one-rank communicator still runs the collective: used to test the RCCL path)
    // the exchange stream has its own communicator (one communicator, one stream)
    Comm *cm = (xcomm && comm_stream && s == comm_stream) ? xcomm : comm;
    HaloPlan &h = is_m ? halo_m : halo_n;
    if (!h.sparse) {
        const size_t chunk = static_cast<size_t>(is_m ? m_pad : n_pad) / cm->size;
        cm->allgather_inplace(gbuf, chunk, s);
        return;
    }
    launch_pack(gbuf, h.send_idx.p, h.sendbuf.p, h.nsend, s);
    cm->exchange(h.ops.data(), static_cast<int>(h.ops.size()), s);
    launch_scatter(gbuf, h.recv_idx.p, h.recvbuf.p, h.nrecv, s);
}

// One exchange of a vector whose entry j is j + 1 on its owner and -1 elsewhere: every entry this shard reads must
// arrive as j + 1.  All ranks agree on the outcome (all-reduce of the failure count); a neighbour exchange that fails
// is replaced by the all-gather on every rank, an all-gather that fails is an error.  Costs one exchange per vector
// length at set-up and keeps a transport problem from turning into silently wrong iterates.
void Solver::verify_exchange() {
    if (!comm || comm->size <= 1) return;
    const bool inject = env_get("HPRLP_DIST_SELFTEST_FAIL") != nullptr;  // tests: first verdict reads "failed"
    DBuf<double> flag(1);
    // with a second communicator both transports are tested: passes 0, 1 through comm on the solver stream, passes 2, 3
    // through xcomm on the exchange stream
    if (xcomm) ensure_comm_stream();
    for (int pass = 0; pass < (xcomm ? 4 : 2); ++pass) {
        const bool is_m = (pass & 1) == 0;
        hipStream_t vs = pass >= 2 ? comm_stream : stream;
        HaloPlan &h = is_m ? halo_m : halo_n;
        const int total = is_m ? m : n, pad = is_m ? m_pad : n_pad;
        const int chunk = pad / comm->size;
        const int lo = std::min(total, comm->rank * chunk), hi = std::min(total, (comm->rank + 1) * chunk);
        std::vector<double> init(static_cast<size_t>(pad), -1.0), got(static_cast<size_t>(pad));
        for (int j = lo; j < hi; ++j) init[j] = j + 1.0;
        DBuf<double> g(static_cast<size_t>(pad));
        for (int attempt = 0;; ++attempt) {
            g.upload(init.data(), init.size());
            gather_on(g.p, is_m, vs);
            HIP_CHECK(hipStreamSynchronize(vs));
            g.download(got.data(), got.size());
            double bad = 0.0;
            if (h.sparse) {
                std::vector<int> want(static_cast<size_t>(h.nrecv));
                h.recv_idx.download(want.data(), want.size());
                for (int j : want) bad += got[j] != j + 1.0;
                for (int j = lo; j < hi; ++j) bad += got[j] != j + 1.0;
                if (inject && attempt == 0) bad += 1.0;
            } else {
                for (int j = 0; j < total; ++j) bad += got[j] != j + 1.0;
            }
            flag.upload(&bad, 1);
            comm->allreduce_sum(flag.p, 1, stream);
            HIP_CHECK(hipStreamSynchronize(stream));
            flag.download(&bad, 1);
            if (bad == 0.0) break;
            if (!h.sparse || attempt > 0)
                throw std::runtime_error("multi-GPU exchange self-test failed: the all-gather did not deliver every rank's slice");
            if (comm->rank == 0)
                std::fprintf(stderr, "hprlp: neighbour exchange failed its self-test (%g wrong entries); using the all-gather\n", bad);
            h.sparse = false;
        }
    }
}

void Solver::fetch_scalars() {
    const auto t0 = time_now();
    HIP_CHECK(hipMemcpyAsync(scal_h.p, scal.p, kNumScalars * sizeof(double), hipMemcpyDeviceToHost, stream));
    const auto t1 = time_now();
    HIP_CHECK(hipStreamSynchronize(stream));
    fetch_enqueue_s += std::chrono::duration<double>(t1 - t0).count();
    fetch_wait_s += time_since(t1);
    ++fetches;
}

static void allreduce_slots(Solver *s, int first, int count) {
    if (!s->comm) return;
    s->comm->allreduce_sum(s->scal.p + first, count, s->stream);
}

double Solver::reduce_sum_sq(const double *v, int n_local) {
    launch_norm2(v, n_local, part_v.p, kReduceBlocks, stream);
    FinalizeArgs f{};
    f.n = 1;
    f.item[0] = {part_v.p, kReduceBlocks, S_TMP0};
    launch_finalize(f, scal.p, stream);
    allreduce_slots(this, S_TMP0, 1);
    fetch_scalars();
    return scal_h[S_TMP0];
}

static double bnorm_sq(Solver *s) {
    launch_bnorm2(s->AL.p, s->AU.p, s->m_loc, s->part_v.p, kReduceBlocks, s->stream);
    FinalizeArgs f{};
    f.n = 1;
    f.item[0] = {s->part_v.p, kReduceBlocks, S_TMP0};
    launch_finalize(f, s->scal.p, s->stream);
    allreduce_slots(s, S_TMP0, 1);
    s->fetch_scalars();
    return s->scal_h[S_TMP0];
}

// ------------------------------------------------------------------------------------------------
// scaling (reference src/scaling.cu:88-216).  t1 lives in the gathered m-vector gsm, t2 in gsn, so
// that the column-side scaling of each stored matrix can index the full vector.
// ------------------------------------------------------------------------------------------------
void Solver::scale() {
    invalidate_far();
    const auto t0 = time_now();
    overlap_ready = false;  // the split copies of the shards carry matrix values
    ovA.reset();
    ovAT.reset();
    // Every matrix-scaling pass that a Ruiz pass follows also leaves that pass's row norms (max |a| of the scaled rows, kernels.hip:
    // k_scale_matrix<.., NEXT>) in a second pair of gathered vectors, so the norm passes over the matrices are not run; the pairs
    // change roles pass by pass.  HPRLP_NO_FUSED_NORMS=1: separate norm passes (A/B runs, tests).
    const bool fuse_norms = prm.use_Ruiz_scaling && env_get("HPRLP_NO_FUSED_NORMS") == nullptr;
    DBuf<double> gsm2, gsn2;
    if (fuse_norms) {
        gsm2.alloc(static_cast<size_t>(m_pad));
        gsn2.alloc(static_cast<size_t>(n_pad));
        HIP_CHECK(hipMemsetAsync(gsm2.p, 0, sizeof(double) * m_pad, stream));
        HIP_CHECK(hipMemsetAsync(gsn2.p, 0, sizeof(double) * n_pad, stream));
    }
    double *gm = gsm.p, *gn = gsn.p, *gm_next = gsm2.p, *gn_next = gsn2.p;
    double *t1 = gm + row_off, *t2 = gn + col_off;
    double *t1n = fuse_norms ? gm_next + row_off : nullptr, *t2n = fuse_norms ? gn_next + col_off : nullptr;
    bool have_norms = false;  // t1n / t2n hold the next pass's row norms
    const bool max_next = fuse_norms;  // (a CR scaling pass is followed by Ruiz pass 0)
    launch_fill(row_norm.p, 1.0, m_loc, stream);
    launch_fill(col_norm.p, 1.0, n_loc, stream);
    norm_b_org = 1.0 + std::sqrt(bnorm_sq(this));
    norm_c_org = 1.0 + std::sqrt(reduce_sum_sq(c.p, n_loc));

    if (prm.use_CR_scaling) {  // :40-83
        HIP_CHECK(hipMemsetAsync(gsm.p, 0, sizeof(double) * m_pad, stream));
        HIP_CHECK(hipMemsetAsync(gsn.p, 0, sizeof(double) * n_pad, stream));
        // Matrices with a tiled copy run the 40 passes through the tiled kernel on a second value array of the copy that
        // holds -log|a| (formed once; released when the passes are done).  HPRLP_NO_TILED_CR=1: stream kernel (A/B runs).
        finish_tiling();
        struct LogValues {
            DBuf<double> tile, far;
        } logA, logAT;
        const bool tiled_cr = env_get("HPRLP_NO_TILED_CR") == nullptr;
        auto log_values = [&](DeviceMatrix &M, LogValues &lv) {
            if (!tiled_cr || !cr_runs_tiled(M.view)) return;
            lv.tile.alloc(static_cast<size_t>(std::max<long>(M.tiled.n_tile, 1)));
            lv.far.alloc(static_cast<size_t>(std::max<long>(M.tiled.n_rem, 1)));
            launch_tiled_refresh_log(M.tiled, M.val.p, lv.tile.p, lv.far.p, stream);
        };
        log_values(A, logA);
        log_values(AT, logAT);
        for (int it = 0; it < 20; ++it) {
            launch_cr_log_update(A.view, gsn.p, t1, stream, logA.tile.p, logA.far.p);
            gather(gsm.p, true);
            launch_cr_log_update(AT.view, gsm.p, t2, stream, logAT.tile.p, logAT.far.p);
            gather(gsn.p, false);
        }
        if (logA.tile.p || logAT.tile.p) HIP_CHECK(hipStreamSynchronize(stream));  // the log values are released below
        launch_exp_clamp(t1, m_loc, stream);
        launch_exp_clamp(t2, n_loc, stream);
        gather(gsm.p, true);
        gather(gsn.p, false);
        launch_vec_scale(row_norm.p, t1, m_loc, true, stream);
        launch_vec_scale(col_norm.p, t2, n_loc, true, stream);
        launch_scale_matrix(A.view, t1, gsn.p, /*row_first=*/true, /*divide=*/false, stream, max_next ? t1n : nullptr);
        launch_scale_matrix(AT.view, t2, gsm.p, /*row_first=*/false, /*divide=*/false, stream, max_next ? t2n : nullptr);
        have_norms = max_next;
        launch_vec_scale(AL.p, t1, m_loc, false, stream);
        launch_vec_scale(AU.p, t1, m_loc, false, stream);
        launch_vec_scale(c.p, t2, n_loc, false, stream);
        launch_vec_scale(l.p, t2, n_loc, true, stream);
        launch_vec_scale(u.p, t2, n_loc, true, stream);
    }

    const int passes = (prm.use_Ruiz_scaling ? 10 : 0) + (prm.use_Pock_Chambolle_scaling ? 1 : 0);
    for (int it = 0; it < passes; ++it) {  // Ruiz :123-153 then Pock-Chambolle :157-183
        const int norm = (prm.use_Ruiz_scaling && it < 10) ? 99 : 1;
        if (have_norms) {  // left by the previous scaling pass in the other pair of vectors
            std::swap(gm, gm_next);
            std::swap(gn, gn_next);
            t1 = gm + row_off, t2 = gn + col_off, t1n = gm_next + row_off, t2n = gn_next + col_off;
        } else {
            launch_row_norm(A.view, t1, norm, stream);
            launch_row_norm(AT.view, t2, norm, stream);
        }
        gather(gm, true);
        gather(gn, false);
        launch_vec_scale(row_norm.p, t1, m_loc, false, stream);
        launch_vec_scale(AL.p, t1, m_loc, true, stream);
        launch_vec_scale(AU.p, t1, m_loc, true, stream);
        launch_vec_scale(col_norm.p, t2, n_loc, false, stream);
        const bool next = fuse_norms && it + 1 < passes && it + 1 < 10;  // a max-norm (Ruiz) pass follows
        launch_scale_matrix(A.view, t1, gn, true, true, stream, next ? t1n : nullptr);
        launch_scale_matrix(AT.view, t2, gm, false, true, stream, next ? t2n : nullptr);
        have_norms = next;
        launch_vec_scale(c.p, t2, n_loc, true, stream);
        launch_vec_scale(l.p, t2, n_loc, false, stream);
        launch_vec_scale(u.p, t2, n_loc, false, stream);
    }

    if (prm.use_bc_scaling) {  // :185-202
        b_scale = 1.0 + std::sqrt(bnorm_sq(this));
        c_scale = 1.0 + std::sqrt(reduce_sum_sq(c.p, n_loc));
        const double bs = 1.0 / b_scale, cs = 1.0 / c_scale;
        launch_vec_scal(AU.p, bs, m_loc, stream);
        launch_vec_scal(AL.p, bs, m_loc, stream);
        launch_vec_scal(l.p, bs, n_loc, stream);
        launch_vec_scal(u.p, bs, n_loc, stream);
        launch_vec_scal(c.p, cs, n_loc, stream);
    } else {
        b_scale = 1.0;
        c_scale = 1.0;
    }
    norm_b = std::sqrt(bnorm_sq(this));
    norm_c = std::sqrt(reduce_sum_sq(c.p, n_loc));
    finish_tiling();  // tiled copies still being built pick up the scaled values here ...
    A.refresh_tiled(stream);  // ... finished ones are refreshed
    AT.refresh_tiled(stream);
    refresh_bound_codes();
    HIP_CHECK(hipMemsetAsync(gsm.p, 0, sizeof(double) * m_pad, stream));
    HIP_CHECK(hipMemsetAsync(gsn.p, 0, sizeof(double) * n_pad, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    scaling_time = time_since(t0);
}

// ------------------------------------------------------------------------------------------------
// lambda_max(A A^T) by the power method (reference src/power_iteration.cu:20-119).  q lives in gsm,
// A^T q in gsn, z in sm1; the dots ride on the second SpMV's epilogue and stay on the device; the
// host reads back only at the every-10th-iteration convergence check.
// ------------------------------------------------------------------------------------------------
double Solver::power_iteration(int max_iter, double tol, int *iters) {
    finish_tiling();
    invalidate_far();
    const auto t0 = time_now();
    double *q = gsm.p + row_off, *ATq = gsn.p + col_off, *z = sm1.p;
    // large vectors in the caller's numbering are filled on the device (last-place differences from the host's libm are
    // possible there; below the threshold the start vector is the oracle's bit for bit)
    const bool host_start = env_get("HPRLP_HOST_POWER_START") != nullptr;  // (tests)
    if (m_loc > kDeviceStartRows && perm_r.empty() && !host_start) {
        launch_pw_start(m_loc, 1ULL, row_off, z, stream);
    } else {
        std::vector<double> z0(static_cast<size_t>(std::max(m_loc, 1)));
        power_start_vector(m_loc, 1ULL, row_off, z0.data());
        if (!perm_r.empty()) {  // the start vector is defined in the caller's row numbering
            std::vector<double> zp(z0.size());
            for (int i = 0; i < m_loc; ++i) zp[i] = z0[perm_r[i]];
            z0.swap(zp);
        }
        HIP_CHECK(hipMemcpyAsync(z, z0.data(), sizeof(double) * m_loc, hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
    }
    const bool no_small_power = env_get("HPRLP_NO_SMALL_POWER") != nullptr;  // tests: the regular kernels instead
    if (use_small && !comm && !no_small_power) {
        // Netlib-scale LP: the whole power iteration in one launch of the single-workgroup kernel (small.hip), stopping test on
        // the device; the host waits once
        const SmallArgs a{m, n, A.view.nnz, A.view.rowptr, AT.view.rowptr, AT.view.val, small_ij.p, small_posA.p,
                          small_order_x.p, small_order_y.p, x.p, x_hat, y, l.p, u.p, c.p, last_x.p, AL.p, AU.p, last_y.p, ctrl.p};
        HIP_CHECK(hipMemsetAsync(scal.p + S_SMALL_PW_LAMBDA, 0, 2 * sizeof(double), stream));
        launch_small_power(a, z, max_iter, tol, scal.p + S_SMALL_PW_LAMBDA, stream);
        fetch_scalars();
        const double lambda_dev = scal_h[S_SMALL_PW_LAMBDA];
        const int done_dev = static_cast<int>(scal_h[S_SMALL_PW_ITERS]);
        // a kernel that did not run leaves the zeroed slots; a lambda that is not a positive finite number is no eigenvalue
        // estimate either: the regular kernels below take over (same start vector, still in z)
        if (done_dev > 0 && std::isfinite(lambda_dev) && lambda_dev > 0.0) {
            if (iters) *iters = done_dev;
            power_iters = done_dev;
            power_time = time_since(t0);
            return lambda_dev;
        }
        if (verbose) std::cerr << "[hprlp] single-launch power iteration returned lambda = " << lambda_dev << " after " << done_dev
                               << " iterations: falling back to the regular kernels" << std::endl;
    }
    launch_norm2(z, m_loc, part_v.p, kReduceBlocks, stream);
    FinalizeArgs f0{};
    f0.n = 1;
    f0.item[0] = {part_v.p, kReduceBlocks, S_PW_ZZ};
    launch_finalize(f0, scal.p, stream);
    allreduce_slots(this, S_PW_ZZ, 1);

    double lambda = 1.0;
    int done = max_iter;
    const int gridA = A.view.grid();
    auto one_iteration = [&]() {
        launch_pw_normalize(z, q, m_loc, scal.p, stream);
        gather(gsm.p, true);
        // A^T q: its epilogue writes the products A's remainder needs (hand-off, as between the half-steps; A then runs without a
        // pre-pass) -- where the remainder is a large part of A.  With a few per cent of the entries in it the pre-pass is the
        // cheaper way since round 4: config 5 (5 %), same box: push 104 us of the launch against 64 us of k_far_products, power
        // iteration 0.385 -> 0.374 s
        const bool push_pays = A.view.tiled.valid && static_cast<double>(A.tiled.n_rem) >= 0.3 * static_cast<double>(A.view.nnz);
        const bool handed = launch_spmv_push(AT.view, gsm.p, ATq, push_pays ? push_into(A, AT) : FarPush{}, stream);
        gather(gsn.p, false);
        launch_spmv_plain(A.view, gsn.p, z, q, true, part_y.p, stride_y, stream, handed);
        FinalizeArgs f{};
        f.n = 2;
        f.item[0] = {part_y.p, gridA, S_PW_ZZ};
        f.item[1] = {part_y.p + stride_y, gridA, S_PW_QZ};
        launch_finalize(f, scal.p, stream);
        allreduce_slots(this, S_PW_ZZ, 2);
    };
    auto error_terms = [&]() {
        launch_pw_err(z, q, m_loc, scal.p, part_v.p, kReduceBlocks, stream);
        FinalizeArgs fe{};
        fe.n = 1;
        fe.item[0] = {part_v.p, kReduceBlocks, S_PW_ERR2};
        launch_finalize(fe, scal.p, stream);
        allreduce_slots(this, S_PW_ERR2, 1);
    };
    // One rank: the ten iterations between two stopping tests are one graph launch (about 50 kernels; enqueued one by one the
    // launches of config 5 leave 0.2 ms of every 1.45 ms iteration idle, on config 3 the iteration is launch-bound altogether).
    hipGraphExec_t block = nullptr;
    if (use_graph && max_iter >= 10) {
        auto it = graphs.find(kPowerBlockKey);
        if (it != graphs.end()) {
            block = it->second;
        } else {
            hipGraph_t g = nullptr;
            HIP_CHECK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
            for (int k = 0; k < 10; ++k) one_iteration();
            error_terms();
            HIP_CHECK(hipStreamEndCapture(stream, &g));
            HIP_CHECK(hipGraphInstantiate(&block, g, nullptr, nullptr, 0));
            HIP_CHECK(hipGraphDestroy(g));
            graphs[kPowerBlockKey] = block;
        }
    }
    for (int i = 1; i <= max_iter; ++i) {
        if (block && i % 10 == 1 && i + 9 <= max_iter) {
            HIP_CHECK(hipGraphLaunch(block, stream));
            i += 9;
        } else {
            one_iteration();
            if (i % 10 == 0) error_terms();
        }
        if (i % 10 == 0) {
            fetch_scalars();
            lambda = scal_h[S_PW_QZ];
            const double err = std::sqrt(scal_h[S_PW_ERR2]);
            if (err < tol) {
                done = i;
                break;
            }
        }
    }
    if (iters) *iters = done;
    power_iters = done;
    // leave the scratch vectors clean
    HIP_CHECK(hipMemsetAsync(gsm.p, 0, sizeof(double) * m_pad, stream));
    HIP_CHECK(hipMemsetAsync(gsn.p, 0, sizeof(double) * n_pad, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    power_time = time_since(t0);
    return lambda;
}

// ------------------------------------------------------------------------------------------------
void Solver::set_sigma_lambda(double sigma_, double lambda_, bool reset_k) {
    sigma = sigma_;
    lambda_max = lambda_;
    launch_set_ctrl(ctrl.p, sigma, lambda_max, reset_k ? 1 : 0, stream);
}

// hooks that the iteration path consults: read when the solver is set up (a later change of the environment does not reach it)
void Solver::read_hooks() {
    hook_no_far_push = env_get("HPRLP_NO_FAR_PUSH") && env_get("HPRLP_NO_FAR_PUSH")[0] == '1';
    hook_no_bound_codes = env_get("HPRLP_NO_BOUND_CODES") != nullptr;  // A/B runs: always read l and u
    hook_store_x = env_get("HPRLP_STORE_X") != nullptr;                // A/B runs: every x-half reads and stores x
}

void Solver::refresh_bound_codes() {
    if (hook_no_bound_codes) return;
    if (n_loc > 0) {
        if (lu_code.n != static_cast<size_t>(n_loc)) lu_code.alloc(static_cast<size_t>(n_loc));
        launch_bound_codes(n_loc, l.p, u.p, lu_code.p, stream);
    }
    if (m_loc > 0) {
        if (row_code.n != static_cast<size_t>(m_loc)) row_code.alloc(static_cast<size_t>(m_loc));
        launch_row_codes(m_loc, AL.p, AU.p, row_code.p, stream);
    }
}

// Back to the state of a fresh solver after scale() and power_iteration(): every iterate and work vector zero, nothing handed over.
// (bench.py: the timed iterations and the solve to tolerance of a multi-GPU run use ONE solver -- a second one would need a
// second set of communicators.)  Call init_iteration_state() / set_sigma_lambda() afterwards, as after create.
void Solver::reset_iterates() {
    if (y_exchange_pending) throw std::runtime_error("reset_iterates: an exchange is still pending");
    invalidate_far();
    auto zero = [&](double *p, size_t n) {
        if (p && n > 0) HIP_CHECK(hipMemsetAsync(p, 0, sizeof(double) * n, stream));
    };
    zero(x.p, x.n); zero(last_x.p, last_x.n); zero(z_bar.p, z_bar.n);
    zero(last_y.p, last_y.n); zero(y_obj.p, y_obj.n); zero(y_temp.p, y_temp.n);
    zero(gy.p, gy.n); zero(gyb.p, gyb.n); zero(gsm.p, gsm.n);
    zero(gxh.p, gxh.n); zero(gxb.p, gxb.n); zero(gxt.p, gxt.n); zero(gsn.p, gsn.n);
    zero(sm1.p, sm1.n); zero(sn1.p, sn1.n);
    zero(scal.p, scal.n);
    HIP_CHECK(hipMemsetAsync(ctrl.p, 0, sizeof(Ctrl), stream));
    HIP_CHECK(hipStreamSynchronize(stream));
}

void Solver::init_iteration_state() {  // reference src/HPRLP.cu:154-167
    finish_tiling();
    invalidate_far();
    refresh_bound_codes();
    if (overlap_enabled && !overlap_ready) prepare_overlap();  // set-up work, not part of the first iteration
    const double s0 = (norm_b > 1e-8 && norm_c > 1e-8) ? norm_b / norm_c : 1.0;
    set_sigma_lambda(s0, lambda_max, true);
}

// Column split of one shard on the device: local = entries whose column lies in [lo, hi).
static void split_shard(const DeviceMatrix &M, int lo, int hi, Solver::SplitShard *out, hipStream_t s) {
    const int rows = M.view.rows, cols = M.view.cols;
    device_split_columns(rows, M.view.nnz, M.rowptr.p, M.col.p, M.val.p, lo, hi, out->loc.rowptr, out->loc.col, out->loc.val,
                         out->rem.rowptr, out->rem.col, out->rem.val, s);
    std::vector<int> rp(static_cast<size_t>(rows) + 1);
    out->loc.rowptr.download(rp.data(), rp.size());
    out->loc.describe(rows, cols, rp.data(), nullptr, nullptr);
    out->rem.rowptr.download(rp.data(), rp.size());
    out->rem.describe(rows, cols, rp.data(), nullptr, nullptr);
    out->loc.finish_tiling(s);
    out->rem.finish_tiling(s);
    out->part.alloc_zero(static_cast<size_t>(std::max(rows, 1)));
    // The local part runs BESIDE the exchange's kernels.  The persistent schedule of the tiled kernel assumes that all
    // of its workgroups are resident; a CU that also hosts a workgroup of the exchange would leave one of them waiting
    // for a whole cohort list.  One workgroup per super-block (dispatched as slots free up) has no such tail.
    if (out->loc.view.tiled.valid) out->loc.view.tiled.grid = 8 * out->loc.view.tiled.per;
}

void Solver::ensure_comm_stream() {
    if (!comm_stream) {
        // highest priority: the few workgroups of the exchange must not queue behind the half-step's grid
        int prio_low = 0, prio_high = 0;
        HIP_CHECK(hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
        HIP_CHECK(hipStreamCreateWithPriority(&comm_stream, hipStreamDefault, prio_high));
        HIP_CHECK(hipEventCreateWithFlags(&ev_ready, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&ev_done_x, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&ev_done_y, hipEventDisableTiming));
    }
}

void Solver::prepare_overlap() {
    finish_tiling();
    HIP_CHECK(hipStreamSynchronize(stream));
    ensure_comm_stream();
    ovAT.reset(new SplitShard);  // A^T shard: n_loc rows, columns = rows of A; this rank owns y[row_off, row_off + m_loc)
    split_shard(AT, row_off, row_off + m_loc, ovAT.get(), stream);
    ovA.reset(new SplitShard);   // A shard: m_loc rows; this rank owns x_hat[col_off, col_off + n_loc)
    split_shard(A, col_off, col_off + n_loc, ovA.get(), stream);
    HIP_CHECK(hipDeviceSynchronize());
    overlap_ready = true;
}

// The hand-off needs the PRODUCER to run the fused tiled kernel on one GPU (a super-block's rows = one source group of
// the consumer's remainder) and the consumer to have remainder lists; HPRLP_NO_FAR_PUSH=1 keeps the pre-pass (A/B runs).
FarPush Solver::push_into(const DeviceMatrix &consumer, const DeviceMatrix &producer) const {
    const TiledDev &pt = producer.view.tiled;
    if (hook_no_far_push || comm || !pt.valid || pt.n_pieces > 0) return FarPush{};
    if (consumer.view.tiled.valid && consumer.view.tiled.G != pt.R) return FarPush{};  // a source group must be ONE super-block of the producer
    return far_push_of(consumer.view);
}

int Solver::x_mode_of(int i, int count) const {
    if (hook_store_x || overlap_enabled || !AT.view.tiled.valid) return 0;
    return (i > 0 ? kXRebuild : 0) | (i + 1 < count ? kXNoStore : 0);
}

void Solver::launch_normal_pair(bool more_follow, hipEvent_t *ev, int x_mode) {
    XHalfArgs xa{gy.p, x.p, x_hat, l.p, u.p, c.p, last_x.p, nullptr, nullptr, nullptr, ctrl.p, nullptr, 0};
    xa.lu_code = lu_code.p;
    xa.x_mode = x_mode;
    YHalfArgs ya{gxh.p, y, AL.p, AU.p, last_y.p, nullptr, nullptr, nullptr, ctrl.p, nullptr, 0};  // (push / far_ready set below)
    ya.row_code = row_code.p;
    if (ev) HIP_CHECK(hipEventRecord(ev[0], stream));
    if (!overlap_enabled) {
        xa.push = push_into(A, AT);
        xa.far_ready = far_AT_ready;
        far_A_ready = launch_x_half(AT.view, xa, false, stream);
        if (ev) HIP_CHECK(hipEventRecord(ev[1], stream));
        gather(gxh.p, false);
        ya.push = push_into(AT, A);
        ya.far_ready = far_A_ready;
        far_AT_ready = launch_y_half(A.view, ya, false, stream);
        if (ev) HIP_CHECK(hipEventRecord(ev[2], stream));
        gather(gy.p, true);
        return;
    }
    invalidate_far();
    if (!overlap_ready) prepare_overlap();
    // Exchange on comm_stream behind ev_ready, beside the local-column SpMV on the solver stream.  RCCL only enqueues,
    // so the exchange goes first and its workgroups are placed before the SpMV's grid fills the chip; the in-process
    // group blocks the host inside the exchange, so there the SpMV is launched first to run beside it.
    auto exchange_beside = [&](double *gbuf, bool is_m, hipEvent_t done, const CsrDev *local, double *part) {
        HIP_CHECK(hipEventRecord(ev_ready, stream));
        const bool spmv_first = overlap_spmv_first;
        if (local && spmv_first) launch_spmv_plain(*local, gbuf, part, nullptr, false, nullptr, 0, stream);
        HIP_CHECK(hipStreamWaitEvent(comm_stream, ev_ready, 0));
        gather_on(gbuf, is_m, comm_stream);
        HIP_CHECK(hipEventRecord(done, comm_stream));
        if (local && !spmv_first) launch_spmv_plain(*local, gbuf, part, nullptr, false, nullptr, 0, stream);
    };
    // ---- x-half: if the exchange of y is still in flight, the local-column part runs beside it
    if (y_exchange_pending) {
        if (!overlap_spmv_first)  // (the in-process group launched it before its blocking exchange, below)
            launch_spmv_plain(ovAT->loc.view, gy.p, ovAT->part.p, nullptr, false, nullptr, 0, stream);
        HIP_CHECK(hipStreamWaitEvent(stream, ev_done_y, 0));
        launch_x_half_base(ovAT->rem.view, xa, ovAT->part.p, stream);
        y_exchange_pending = false;
    } else {
        launch_x_half(AT.view, xa, false, stream);  // gathered y complete: the unsplit shard in one launch
    }
    if (ev) HIP_CHECK(hipEventRecord(ev[1], stream));
    // ---- exchange of x_hat beside the local-column part of the y-half
    exchange_beside(gxh.p, false, ev_done_x, &ovA->loc.view, ovA->part.p);
    HIP_CHECK(hipStreamWaitEvent(stream, ev_done_x, 0));
    launch_y_half_base(ovA->rem.view, ya, ovA->part.p, stream);
    if (ev) HIP_CHECK(hipEventRecord(ev[2], stream));
    // ---- exchange of y: beside the local-column part of the next pair's x-half, or waited for
    exchange_beside(gy.p, true, ev_done_y, more_follow && overlap_spmv_first ? &ovAT->loc.view : nullptr, ovAT->part.p);
    if (more_follow) y_exchange_pending = true;
    else HIP_CHECK(hipStreamWaitEvent(stream, ev_done_y, 0));
}

void Solver::step(bool check) {
    finish_tiling();
    if (!check) {
        launch_normal_pair();
        return;
    }
    XHalfArgs xa{gy.p, x.p, x_hat, l.p, u.p, c.p, last_x.p, x_bar, z_bar.p, x_temp, ctrl.p, part_x.p, stride_x};
    xa.lu_code = lu_code.p;
    xa.push = push_into(A, AT);
    xa.far_ready = far_AT_ready;
    far_A_ready = launch_x_half(AT.view, xa, true, stream);
    gather(gxh.p, false);
    YHalfArgs ya{gxh.p, y, AL.p, AU.p, last_y.p, y_bar, y_obj.p, y_temp.p, ctrl.p, part_y.p, stride_y};
    ya.row_code = row_code.p;
    ya.push = push_into(AT, A);
    ya.far_ready = far_A_ready;
    far_AT_ready = launch_y_half(A.view, ya, true, stream);
    gather(gy.p, true);
    FinalizeArgs f{};
    const int gx = AT.view.grid(), gyy = A.view.grid();
    f.n = 5;
    f.item[0] = {part_x.p, gx, S_CX};
    f.item[1] = {part_x.p + stride_x, gx, S_XZ};
    f.item[2] = {part_x.p + 2 * static_cast<size_t>(stride_x), gx, S_DX2};
    f.item[3] = {part_y.p, gyy, S_YOBJ_Y};
    f.item[4] = {part_y.p + stride_y, gyy, S_DY2};
    launch_finalize(f, scal.p, stream);
}

hipGraphExec_t Solver::graph_for(int len) {
    auto it = graphs.find(len);
    if (it != graphs.end()) return it->second;
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    // a replayed graph cannot look at the hand-off flags: it always starts with the pre-pass of A^T (as if nothing had been
    // handed over) and ends with both buffers handed over, whatever ran before it
    far_AT_ready = false;
    HIP_CHECK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < len; ++i) launch_normal_pair(false, nullptr, x_mode_of(i, len));
    HIP_CHECK(hipStreamEndCapture(stream, &g));
    graph_end_A = far_A_ready;
    graph_end_AT = far_AT_ready;
    invalidate_far();  // nothing has run yet
    HIP_CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    HIP_CHECK(hipGraphDestroy(g));
    graphs[len] = ge;
    return ge;
}

void Solver::run_normal(int count) {
    if (count <= 0) return;
    finish_tiling();
    if (use_small && !comm) {
        // Netlib-scale LP: all `count` iterations in one single-workgroup launch, matrices in registers (small.hip)
        const SmallArgs a{m, n, A.view.nnz, A.view.rowptr, AT.view.rowptr, AT.view.val, small_ij.p, small_posA.p,
                          small_order_x.p, small_order_y.p, x.p, x_hat, y, l.p, u.p, c.p, last_x.p, AL.p, AU.p, last_y.p, ctrl.p};
        launch_small_iterations(a, count, stream);
        return;
    }
    if (!use_graph) {
        for (int i = 0; i < count; ++i) launch_normal_pair(i + 1 < count, nullptr, x_mode_of(i, count));
        return;
    }
    while (count > 0) {
        const int len = std::min(count, kMaxGraphIters);
        HIP_CHECK(hipGraphLaunch(graph_for(len), stream));
        far_A_ready = graph_end_A;
        far_AT_ready = graph_end_AT;
        count -= len;
    }
}

void Solver::run_normal_then_check(int count) {
    // (a fused single-workgroup check + residual launch for Netlib-scale LPs was built and measured in round 4 -- no faster than
    // the eight small launches it replaced, profiles/r04_small_check.txt -- and taken out again in round 5)
    run_normal(count);
    step(true);
}

// ------------------------------------------------------------------------------------------------
// residuals (reference src/main_iterate.cu:229-309).  c.x_bar, y_obj.y_bar, x_bar.z_bar, |x_temp|^2
// and |y_temp|^2 were already reduced by the check-variant epilogues of the step that produced them.
// ------------------------------------------------------------------------------------------------
static double weighted_norm_from(Solver *s, double dot_adx_dy, double dy2, double dx2) {
    const double dot_prod = 2.0 * dot_adx_dy;
    double wn = s->sigma * (s->lambda_max * dy2) + dx2 / s->sigma + dot_prod;
    if (wn < 0) {
        if (s->verbose)
            std::cout << "The estimated maximum eigenvalue is too small! Current value is " << s->lambda_max << "\n";
        s->lambda_max = -(dot_prod + dx2 / s->sigma) / (s->sigma * dy2) * 1.05;
        if (s->verbose) std::cout << "The new estimated maximum eigenvalue is " << s->lambda_max << "\n";
        wn = std::sqrt(-(dot_prod + dx2 / s->sigma) * 0.05);
        s->set_sigma_lambda(s->sigma, s->lambda_max, false);
    } else {
        wn = std::sqrt(wn);
    }
    return wn;
}

void Solver::compute_residuals(int iter, bool compute_gap, Residuals *r, RestartState *rs) {
    invalidate_far();  // the residual SpMVs refill the remainder buffers for x_bar / y_bar
    finish_tiling();
    const int gx = AT.view.grid(), gyy = A.view.grid();
    const int rstride = std::max(stride_x, stride_y);
    gather(gyb.p, true);
    gather(gxb.p, false);
    if (compute_gap) gather(gxt.p, false);
    FinalizeArgs f{};
    launch_resid_d(AT.view, gyb.p, c.p, z_bar.p, col_norm.p, part_x.p, stream);
    f.item[f.n++] = {part_x.p, gx, S_RD2};
    launch_resid_p(A.view, gxb.p, gxt.p, AL.p, AU.p, row_norm.p, y_temp.p, compute_gap, part_r.p, rstride, stream);
    f.item[f.n++] = {part_r.p, gyy, S_RP2};
    if (compute_gap) f.item[f.n++] = {part_r.p + rstride, gyy, S_ADX_DY};
    if (iter == 0) {
        launch_lu(n_loc, x_bar, l.p, u.p, col_norm.p, x_temp, part_v.p, kReduceBlocks, stream);
        f.item[f.n++] = {part_v.p, kReduceBlocks, S_LU2};
    }
    launch_finalize(f, scal.p, stream);
    allreduce_slots(this, S_CX, 8);
    if (iter == 0) allreduce_slots(this, S_LU2, 1);
    fetch_scalars();

    const double obj_scale = b_scale * c_scale;
    r->primal_obj = obj_scale * scal_h[S_CX] + obj_constant;
    r->dual_obj = obj_scale * (scal_h[S_YOBJ_Y] + scal_h[S_XZ]) + obj_constant;
    r->rel_gap = std::abs(r->primal_obj - r->dual_obj) / (1.0 + std::abs(r->primal_obj) + std::abs(r->dual_obj));
    r->err_Rd = c_scale * std::sqrt(scal_h[S_RD2]) / norm_c_org;
    r->err_Rp = b_scale * std::sqrt(scal_h[S_RP2]) / norm_b_org;
    if (iter == 0) r->err_Rp = std::max(r->err_Rp, b_scale * std::sqrt(scal_h[S_LU2]));
    r->kkt = std::max(std::max(r->err_Rd, r->err_Rp), r->rel_gap);
    if (compute_gap && rs) rs->current_gap = weighted_norm_from(this, scal_h[S_ADX_DY], scal_h[S_DY2], scal_h[S_DX2]);
}

double Solver::weighted_norm_after_restart() {
    invalidate_far();
    gather(gxt.p, false);
    launch_gap(A.view, gxt.p, y_temp.p, part_r.p, stream);
    FinalizeArgs f{};
    f.n = 1;
    f.item[0] = {part_r.p, A.view.grid(), S_ADX_DY};
    launch_finalize(f, scal.p, stream);
    allreduce_slots(this, S_ADX_DY, 3);  // S_ADX_DY, S_DY2, S_DX2 are adjacent
    fetch_scalars();
    return weighted_norm_from(this, scal_h[S_ADX_DY], scal_h[S_DY2], scal_h[S_DX2]);
}

static void check_restart(RestartState *rs, int iter, int check_iter, double sigma, bool verbose) {
    rs->flag = 0;
    if (rs->first) {
        if (iter == check_iter) {
            rs->first = false;
            rs->flag = 1;
            rs->best_gap = rs->current_gap;
            rs->best_sigma = sigma;
        }
    } else if (iter % check_iter == 0) {
        if (rs->current_gap < 0) {
            rs->current_gap = 1e-6;
            if (verbose) std::cout << "current_gap < 0" << std::endl;
        }
        if (rs->current_gap <= 0.2 * rs->last_gap) { rs->sufficient += 1; rs->flag = 1; }
        if (rs->current_gap <= 0.6 * rs->last_gap && rs->current_gap > 1.00 * rs->save_gap) { rs->necessary += 1; rs->flag = 2; }
        if (rs->inner >= 0.2 * iter) { rs->long_ += 1; rs->flag = 3; }
        if (rs->best_gap > rs->current_gap) { rs->best_gap = rs->current_gap; rs->best_sigma = sigma; }
        rs->save_gap = rs->current_gap;
    }
}

void Solver::update_sigma_and_restart(RestartState *rs, const Residuals &r) {
    if (rs->flag <= 0) return;
    // movement x_bar - last_x, y_bar - last_y and their norms (update_sigma, main_iterate.cu:367-404)
    launch_movement(n_loc, m_loc, x_bar, last_x.p, x_temp, y_bar, last_y.p, y_temp.p, part_v.p, kReduceBlocks,
                    kReduceBlocks, stream);
    FinalizeArgs f{};
    f.n = 2;
    f.item[0] = {part_v.p, kReduceBlocks, S_MOVE_X2};
    f.item[1] = {part_v.p + kReduceBlocks, kReduceBlocks, S_MOVE_Y2};
    launch_finalize(f, scal.p, stream);
    allreduce_slots(this, S_MOVE_X2, 2);
    fetch_scalars();
    const double primal_move = std::sqrt(scal_h[S_MOVE_X2]), dual_move = std::sqrt(scal_h[S_MOVE_Y2]);
    double new_sigma = 1.0;
    if (primal_move > 1e-16 && dual_move > 1e-16 && primal_move < 1e12 && dual_move < 1e12) {
        const double ratio = (primal_move / dual_move) / std::sqrt(lambda_max);
        const double fact = std::exp(-0.05 * (rs->current_gap / rs->best_gap));
        const double temp1 = std::max(std::min(r.err_Rd, r.err_Rp), std::min(r.rel_gap, rs->current_gap));
        const double sigma_cand = std::exp(fact * std::log(ratio) + (1 - fact) * std::log(rs->best_sigma));
        double kappa;
        if (temp1 > 9e-10) {
            kappa = 1.0;
        } else if (temp1 > 5e-10) {
            kappa = std::max(std::min(std::sqrt(r.err_Rd / r.err_Rp), 100.0), 1e-2);
        } else {
            kappa = std::max(std::min(r.err_Rd / r.err_Rp, 100.0), 1e-2);
        }
        new_sigma = kappa * sigma_cand;
    }
    // do_restart (main_iterate.cu:312-322) + Halpern reset (:54-66)
    invalidate_far();  // y changes under the remainder buffer of A^T
    launch_restart_copy(n_loc, m_loc, x_bar, x.p, last_x.p, y_bar, y, last_y.p, ctrl.p, stream);
    gather(gy.p, true);
    set_sigma_lambda(new_sigma, lambda_max, true);
    rs->inner = 0;
    rs->times += 1;
    rs->save_gap = std::numeric_limits<double>::infinity();
}

// ------------------------------------------------------------------------------------------------
// the outer loop (reference src/HPRLP.cu:154-310).  `iter` only ever stops at event iterations
// (periodic check, log line, iteration limit); everything between two events is enqueued at once.
// ------------------------------------------------------------------------------------------------
static int next_event(int iter, int check_iter, int max_iter) {
    int j = iter + 1;
    while (true) {
        if (j % check_iter == 0 || j % log_step(j) == 0 || j >= max_iter) return j;
        ++j;
    }
}

void Solver::solve_loop(HPRLP_results *out) {
    const auto t_loop = time_now();
    const double t_before = power_time;  // reported `time` includes the power iteration (HPRLP.cu:150)
    Residuals r;
    RestartState rs;
    rs.best_sigma = sigma;
    bool first4 = true, first6 = true, first8 = true;
    const int check_iter = std::max(prm.check_iter, 1);
    const int max_iter = std::max(prm.max_iter, 0);
    *out = HPRLP_results();
    std::string status = "CONTINUE";
    trace_n = 0;
    if (verbose)
        std::cout << " iter     errRp        errRd         p_obj            d_obj          gap         sigma       time\n"
                  << std::flush;
    int iter = 0;
    while (true) {
        const bool at_limit = iter >= max_iter;
        const bool periodic = (iter % check_iter == 0);
        compute_residuals(iter, periodic && iter > 0, &r, &rs);
        const double elapsed = t_before + time_since(t_loop);
        bool timed_out = elapsed > prm.time_limit;
        if (comm && comm->size > 1) {
            // the residuals are all-reduced, the clocks are not: every rank must take the same TIME_LIMIT decision, or
            // one leaves the loop while its peers enter the next exchange (a collective hang).  Any rank over its limit
            // stops the whole group at this event.
            const double flag = timed_out ? 1.0 : 0.0;
            HIP_CHECK(hipMemcpyAsync(scal.p + S_TMP1, &flag, sizeof(double), hipMemcpyHostToDevice, stream));
            allreduce_slots(this, S_TMP1, 1);
            fetch_scalars();
            timed_out = scal_h[S_TMP1] > 0.0;
        }
        if (r.kkt < prm.stop_tol) status = "OPTIMAL";
        else if (at_limit) status = "ITER_LIMIT";
        else if (timed_out) status = "TIME_LIMIT";
        if (periodic && !at_limit) check_restart(&rs, iter, check_iter, sigma, verbose);
        else rs.flag = 0;
        if (trace && trace_n < trace_cap)
            trace[trace_n++] = TraceRow{iter, rs.flag, r.err_Rp, r.err_Rd, r.primal_obj, r.dual_obj, r.rel_gap, r.kkt,
                                        sigma, rs.current_gap, lambda_max};
        if (verbose) {
            std::cout << std::setw(5) << iter << "    " << std::scientific << std::setprecision(2) << r.err_Rp << "    "
                      << r.err_Rd << "    " << std::setprecision(6) << std::showpos << r.primal_obj << "    "
                      << r.dual_obj << "    " << std::setprecision(2) << std::noshowpos << r.rel_gap << "    " << sigma
                      << "      " << std::fixed << std::setprecision(2) << elapsed << "\n" << std::defaultfloat
                      << std::flush;
        }
        auto mark = [&](bool &first, double thr, int &it_out, double &t_out, const char *label) {
            if (first && r.kkt < thr) {
                it_out = iter;
                t_out = elapsed;
                first = false;
                if (verbose) std::cout << "Residual < " << label << " at iter = " << iter << "\n" << std::flush;
            }
        };
        mark(first4, 1e-4, out->iter4, out->time4, "1e-4");
        mark(first6, 1e-6, out->iter6, out->time6, "1e-6");
        mark(first8, 1e-8, out->iter8, out->time8, "1e-8");
        if (status != "CONTINUE") break;

        const int flag = rs.flag;
        update_sigma_and_restart(&rs, r);
        const int next = next_event(iter, check_iter, max_iter);
        int it = iter;
        if (flag > 0) {
            step(true);
            rs.last_gap = weighted_norm_after_restart();
            ++it;
        }
        if (it < next) run_normal_then_check(next - 1 - it);
        rs.inner += next - iter;
        iter = next;
    }
    if (env_get("HPRLP_TIMING"))
        std::cerr << "[timing] loop: " << time_since(t_loop) << " s, " << iter << " iterations; " << fetches << " scalar fetches: enqueue "
                  << fetch_enqueue_s << " s, wait " << fetch_wait_s << " s" << std::endl;
    std::strncpy(out->status, status.c_str(), sizeof(out->status) - 1);
    out->status[sizeof(out->status) - 1] = '\0';
    out->iter = iter;
    out->gap = r.rel_gap;
    out->residuals = r.kkt;
    out->primal_obj = r.primal_obj;
    out->time = t_before + time_since(t_loop);
    if (out->time4 == 0.0) out->time4 = out->time;
    if (out->time6 == 0.0) out->time6 = out->time;
    if (out->time8 == 0.0) out->time8 = out->time;
    if (out->iter4 == 0) out->iter4 = out->iter;
    if (out->iter6 == 0) out->iter6 = out->iter;
    if (out->iter8 == 0) out->iter8 = out->iter;
}

void Solver::collect_solution(HPRLP_results *out) {
    // scratch: sn1 (x), sm1 (y), gsn local slice (z)
    double *zo = gsn.p + col_off;
    launch_unscale(n_loc, m_loc, x_bar, y_bar, z_bar.p, col_norm.p, row_norm.p, b_scale, c_scale, sn1.p, sm1.p, zo,
                   stream);
    out->x = static_cast<double *>(std::malloc(sizeof(double) * std::max(n_loc, 1)));
    out->y = static_cast<double *>(std::malloc(sizeof(double) * std::max(m_loc, 1)));
    out->z = static_cast<double *>(std::malloc(sizeof(double) * std::max(n_loc, 1)));
    if (!out->x || !out->y || !out->z) throw std::runtime_error("host allocation of the solution failed");
    // (three copies side by side, each driven by its own thread on its own stream, were measured on config 5's 3 x 80 MB: 0.036-0.045 s
    // against 0.017 s for this sequence -- pageable copies serialise inside the runtime)
    HIP_CHECK(hipMemcpyAsync(out->x, sn1.p, sizeof(double) * n_loc, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipMemcpyAsync(out->y, sm1.p, sizeof(double) * m_loc, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipMemcpyAsync(out->z, zo, sizeof(double) * n_loc, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    if (!perm_r.empty()) {  // back to the caller's numbering (reference collect_solution returns the model's order)
        auto unpermute = [](double *v, const std::vector<int> &perm) {
            std::vector<double> tmp(v, v + perm.size());
            for (size_t i = 0; i < perm.size(); ++i) v[perm[i]] = tmp[i];
        };
        unpermute(out->x, perm_c);
        unpermute(out->z, perm_c);
        unpermute(out->y, perm_r);
    }
}

}  // namespace hprlp
