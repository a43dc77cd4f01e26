// tiled.h -- column-tiled copy of a CSR matrix for the fused kernels (private).
//
// Why: with ~20 nonzeros per row and a column window far wider than L1, every gathered 8-byte vector
// element costs a 128-byte L2->L1 line, and the stream kernel is bound by that (profiles/r01_spmv_probe.txt).
// Here a workgroup owns a super-block of kTileRows rows, sweeps the column space in tiles of kTileCols
// columns, stages each tile of the gathered vector in LDS (coalesced, every line fetched once per
// super-block) and accumulates row sums in LDS.  Entries of a (super-block, tile) pair are sorted by
// (row, column) and packed in chunks of 4; a row segment never straddles a chunk (zero-valued
// padding continues the previous row), so exactly one lane touches a given accumulator in a step:
// no atomics, deterministic; the per-row summation order is fixed by the matrix (tiles in the order of
// the super-block's rotated sweep -- finish_schedule() in tiled_build.hip -- layer by layer inside a tile, then the remainder).
// A row's entries beyond four in one tile go to further LAYERS of the tile's list (kTileLayers below; round 5), what the layers
// do not hold and the entries of sparse tiles (the far columns) go to a remainder list.  A random 8-byte gather
// costs a whole 128-byte line of fabric traffic whatever the load flavour (tools/gather_probe.hip), so the remainder is
// not gathered by the row side at all: a pre-pass kernel (k_far_products) walks the list in SOURCE order -- one workgroup
// per group of kFarGroup columns, that slice of the vector staged in LDS with coalesced loads -- and writes every product
// a * v[col] to its slot of a buffer P laid out in DESTINATION order ([super-block][source group]); the tiled kernel then
// streams its super-block's slice of P and adds the products row by row (propagation blocking: both sides stream).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "common.h"

namespace hprlp {

constexpr int kTileThreads = 512;  // workgroup of the tiled kernel (8 waves)
constexpr int kTileRows = 8192;    // rows per super-block of the tall form: 64 KiB of accumulators in LDS
constexpr int kTileCols = 2048;    // columns per tile: 16 KiB of the gathered vector in LDS
constexpr int kTileColsNarrow = 1024;  // round 4: tile width of a copy whose rows would put more than ~1.2 entries into a 2048-column tile
                                   // (narrow bands: 1M x 1M, band 1e4 -- 10 % of the entries sat in row segments longer than a chunk and
                                   // went to the 34-bytes-per-entry remainder); TiledDev::T, chosen by Solver::choose_sb_rows
constexpr int kTileChunk = 4;      // entries per lane per step
constexpr int kTileStepCap = kTileThreads * kTileChunk;                       // entries per tile step
constexpr int kTileRemK = 4;                                                  // remainder entries per lane per step (kernels.hip: remainder_steps)
constexpr int kTileRemCap = (kTileThreads - 4) * kTileRemK;                   // 2032: the step's products fill the tile buffer but for 16 doubles of scratch
constexpr int kTileRemRun = 16;    // (the builder still marks remainder steps in which a row holds more consecutive entries than this -- TileStep::col0 --
                                   // for the statistics line; since round 4 the kernels add rows of any length the same way)
// All-remainder form (round 4; a matrix without column locality: no tile is staged, every entry goes through the
// propagation-blocking lists): the copy has its own fused kernel (kernels.hip: k_pb_fused) whose LDS holds the accumulators of
// at most kPbRowsMax rows and remainder steps of kPbRemCap entries -- twice the kTileRemCap of a copy that also stages
// tiles (16 KiB of scratch there) -- kPbRemK entries per lane, added by a segmented reduction over the lanes.
#ifndef HPRLP_PB_REM_K
#define HPRLP_PB_REM_K 8  // (6 and 8 measured same-box: 8 is 0-2.5 % faster on the unstructured and expander ladder points)
#endif
constexpr int kPbRemK = HPRLP_PB_REM_K;            // remainder entries per lane per step
constexpr int kPbRemCap = kTileThreads * kPbRemK;  // 4096
constexpr int kPbRowsMax = 4096;
constexpr int kPbRunTabCap = 3072;                 // run-table entries a producer holds in LDS (two ints each: 24 KiB)
constexpr int kTileMaxRow = 1024;  // matrices with a longer row are not tiled: a long row's remainder entries all go through ONE workgroup
                                   // (2M x 2M, five rows of L entries, per launch: L = 1000 192 us, 3000 265-327 us, 8000 433-470 us; stream kernel 262 us)
constexpr int kTileDenseMin = 256;                                            // entries for a tile to be staged
static_assert(kTileRemCap % kTileRemK == 0 && kTileRemCap + (kTileThreads / 64) * 3 / 2 <= kTileCols, "tile buffer too small for the remainder scratch");
constexpr int kTileRowBits = 13;  // local row in an entry code of the tall form
// Super-block height (round 3).  The accumulators of a super-block take 8 bytes of LDS per row, so 8192 rows is the most;
// a copy may use fewer (TiledDev::R, any multiple of 64 from kTileRowsMin up; the codes keep 13 bits for the local row).
// Why: a matrix with fewer than 512 tall super-blocks does not fill the chip's 512 workgroup slots with whole super-blocks
// and ran the piece form (three launches, partial sums through memory) or, below 2^20 columns, the stream kernel.  With
// R = rows / 512 every slot gets exactly one super-block, the epilogue stays fused and a half-step is ONE launch with no
// tail: 1M x 1M, band 1e4: 3658 it/s (stream kernel) / 3393 (pieces) -> 5287 (2048-row super-blocks, same-box A/B,
// profiles/r03_ab_rows.txt).  The price is tile traffic -- a staged tile serves R rows -- so the height is only lowered
// when a super-block's column window stays narrow against its entries (Solver::choose_sb_rows).
constexpr int kTileRowsMin = 1024;
static_assert((1 << kTileRowBits) == kTileRows && kTileCols <= (1 << (24 - kTileRowBits)) && kTileChunk == 4,
              "entry codes are 24 bits: 13 of local row, 11 of local column, four per chunk");
inline uint32_t tile_code(int lcol, int row) { return (static_cast<uint32_t>(lcol) << kTileRowBits) | static_cast<uint32_t>(row); }

// Chunk layout of one (super-block, tile) list (round 4).  A row's segment (1-4 entries; longer ones go to the remainder) must
// not straddle a chunk of 4, and rows may come in ANY order inside a tile (the accumulators are addressed by the row code), so the
// segments are bin-packed by length instead of laid down in row order: 4 | 3 + 1 | 2 + 2 | 2 + 1 + 1 | 1 + 1 + 1 + 1.  In row order
// a narrow-band matrix padded 11.7 % of its tile entries -- enough to push every full tile of the banded 2e7 ladder point just
// over a step's 2048 entries and into a second, nearly empty step (40 steps per super-block instead of 25).  Both builders
// use these formulas; a segment's position depends only on its length and its rank among the tile's segments of that length
// (row order).  Padding (value 0, no CSR entry) repeats the row of the slot before it.
struct PackLayout {
    int n1 = 0, n2 = 0, n3 = 0, n4 = 0;  // segments of each length
    int u3 = 0, u2 = 0, r1 = 0, c2 = 0;  // ones that complete the 3-chunks / the odd 2-chunk, ones left over, chunks of the 2-class
    __host__ __device__ void set(int a1, int a2, int a3, int a4) {
        n1 = a1; n2 = a2; n3 = a3; n4 = a4;
        u3 = n1 < n3 ? n1 : n3;
        c2 = (n2 + 1) / 2;
        const int left = n1 - u3;
        u2 = (n2 & 1) ? (left < 2 ? left : 2) : 0;
        r1 = left - u2;
    }
    __host__ __device__ int entries() const { return kTileChunk * (n4 + n3 + c2 + (r1 + 3) / 4); }   // padded length of the list
    // offset (in the tile's list) of the first entry of the j-th segment of length L
    __host__ __device__ int pos(int L, int j) const {
        if (L == 4) return 4 * j;
        if (L == 3) return 4 * (n4 + j);
        if (L == 2) return 4 * (n4 + n3 + j / 2) + 2 * (j & 1);
        if (j < u3) return 4 * (n4 + j) + 3;
        j -= u3;
        if (j < u2) return 4 * (n4 + n3 + n2 / 2) + 2 + j;
        j -= u2;
        return 4 * (n4 + n3 + c2) + j;
    }
    // padding slots that directly follow that segment (they carry its row)
    __host__ __device__ int pads_after(int L, int j) const {
        if (L == 3) return j >= u3 ? 1 : 0;
        if (L == 2) return ((n2 & 1) && j == n2 - 1) ? (u2 == 0 ? 2 : 0) : 0;
        if (L == 1) {
            if (j < u3) return 0;
            j -= u3;
            if (j < u2) return (u2 == 1) ? 1 : 0;   // the single one behind the odd 2-segment leaves one slot
            j -= u2;
            return (j == r1 - 1) ? (4 - (r1 & 3)) & 3 : 0;
        }
        return 0;
    }
};

// Layers of a tile's list (round 5).  A row's entries in ONE tile beyond the first kTileChunk used to send the row's whole segment
// to the remainder lists (34 bytes of traffic per entry against 11): a band of 4 000 columns with 20 entries per row puts five
// entries of a row into a 1024-column tile, and two thirds of such a matrix ended up there.  Now the segment is cut into pieces of
// at most kTileChunk entries; piece q of every row goes to LAYER q of the tile's list.  A layer is laid down like the whole list
// used to be (PackLayout), the layers follow each other, and every layer has its own steps -- a row has at most one piece per
// layer, so inside a step still exactly one lane touches a given accumulator.  At most kTileLayers layers (entries beyond go to
// the remainder), and a layer above the first is kept only while it holds kTileLayerMin entries (half a step's worth: a step
// costs the same whatever it holds); the layers kept are contiguous from the first.  Per-row summation order inside a tile:
// layer by layer.
#ifndef HPRLP_TILE_LAYERS
#define HPRLP_TILE_LAYERS 4       // (developer variants: make variant NAME=l8 DEFS=-DHPRLP_TILE_LAYERS=8)
#endif
#ifndef HPRLP_TILE_LAYER_MIN
#define HPRLP_TILE_LAYER_MIN 1024
#endif
constexpr int kTileLayers = HPRLP_TILE_LAYERS;
constexpr int kTileLayerMin = HPRLP_TILE_LAYER_MIN;

constexpr int kTileResidentPerCu = 2;  // 80 KiB of LDS per workgroup, 160 KiB per CU
constexpr int kFarGroup = kTileRows;   // most source columns per workgroup of the remainder pre-pass (64 KiB of LDS); TiledDev::G
constexpr int kFarThreads = 512;
constexpr long kFarWorkMin = 32768;     // pre-pass work list (TiledDev::f_work): a source group is cut up from this many entries ...
constexpr long kFarWorkOverMean = 4;    // ... and this many times the mean group on

struct TileStep {
    int col0;     // first column of the tile
    int e_begin;  // entry range in tval/tidx (tile step) or rval/rcol/rrow (remainder step)
    int e_end;
    int rot;      // first step of a super-block: offset of the step its rotated sweep starts with (finish_schedule)
};

struct TiledDev {
    bool valid = false;
    int R = kTileRows;         // rows per super-block (kTileRowsMin .. kTileRows, multiple of 64)
    int rem_cap = kTileRemCap; // most entries of a remainder step (kTileRemCap; kPbRemCap for the all-remainder form, which k_pb_fused runs)
    int T = kTileCols;         // columns per tile of this copy (kTileCols or kTileColsNarrow; the codes keep 11 bits for the local column)
    int G = kTileRows;         // columns per source group of the remainder lists (= R of the matrix whose half-step produces
                               // the gathered vector, so that its epilogue can hand the products over: kernels.h FarPush)
    int nsb = 0;               // super-blocks
    int per = 0;               // super-blocks per XCD: nsb rounded up to a multiple of 8, / 8
    int grid = 0;              // launch grid: 8 * min(per, resident workgroups of one XCD)
    bool repeats = false;      // some tile has more than one step: the kernels skip re-staging an unchanged tile
    // piece form (k_tiled_part / k_tiled_finish, kernels.hip): n_pieces > 0 -- the tile steps of all super-blocks, laid end
    // to end, are cut into n_pieces equal ranges, one workgroup each; a piece is a list of segments (a range of one
    // super-block's steps), every segment stores its partial row sums in its own slot of `parts`
    int n_pieces = 0;
    const int *piece_ptr = nullptr;    // n_pieces + 1: segments of a piece
    const int4 *segs = nullptr;        // {super-block, first step (relative), steps, 1 if the super-block's last segment}
    const int *slot_ptr = nullptr;     // nsb + 1: segments (= slots of `parts`) of a super-block, in summation order
    double *parts = nullptr;           // (number of segments) * kTileRows partial row sums
    // Long rows kept OUT of the tiled copy (rows over kTileMaxRow entries: DeviceMatrix::describe): their sums are formed by
    // the stream kernel's vector / split-row mode over the CSR arrays (block list side_blk: descriptor {slot, 1, first
    // nonzero, count} or the chunks of a split row) into base[row], and every tiled launch adds base through the WithBase
    // epilogue (kernels.hip, launch_fused).  side_nblk == 0: no such rows.
    const int4 *side_blk = nullptr;
    int side_nblk = 0;
    const int4 *side_long = nullptr;  // split rows of the side list: {slot, first chunk slot, one past the last, 0}
    int side_nlong = 0;
    double *side_partial = nullptr;
    const int *side_rows = nullptr;   // slot -> row
    double *base = nullptr;           // rows doubles, zero except for the long rows
    unsigned long long *stamps = nullptr;  // HPRLP_TILE_STAMPS=1 (diagnostic): 16 shader-clock sums per piece
    unsigned long long *wgtimes = nullptr;  // HPRLP_WG_TIMES=1 (diagnostic): 8 wall-clock stamps per workgroup of the fused kernel
    int wg_filter = 0;                      // HPRLP_WG_TIMES=2: only the normal half-step kernels (hand-off form, no reductions) stamp
    const int *sb_ptr = nullptr;   // nsb+1: steps of a super-block
    const int *sb_mid = nullptr;   // nsb: first remainder step
    const TileStep *steps = nullptr;
    const double *tval = nullptr;  // tile entries: value
    // tile entries: 24-bit codes (local column << 13 | local row), four per lane chunk packed in three 32-bit words
    // (w0 = e0 | e1 << 24, w1 = e1 >> 8 | e2 << 16, w2 = e2 >> 16 | e3 << 8): 44 instead of 48 bytes per chunk
    const uint32_t *tidx3 = nullptr;
    // remainder entries, destination side: step [e_begin, e_end) owns P[e_begin..e_end) (products, written by the
    // pre-pass of this launch) and rq[e_begin..e_end): the step's entries in (row, CSR) order, slot in P << 16 | local row
    double *P = nullptr;
    const uint32_t *rq = nullptr;
    // remainder entries, source side (grouped by column / kFarGroup, ascending P position inside a group)
    int n_groups = 0;
    const int *f_gptr = nullptr;      // n_groups + 1
    const double *f_val = nullptr;
    const int *f_pos = nullptr;       // position in P
    const uint16_t *f_lcol = nullptr; // column - group * kFarGroup
    // Run tables of the source side (all-remainder form): inside a source group's list the P positions ascend in runs -- one run
    // per destination super-block -- so position = f_rp[run] + (entry - f_rk[run]); the producers look the run up (a table of
    // at most kPbRunTabCap runs per group in LDS, binary search) instead of reading 4 bytes of f_pos per entry.
    const int *f_rptr = nullptr;      // n_groups + 1: runs of a group
    const int *f_rk = nullptr;        // first entry (index into the f lists) of a run
    const int *f_rp = nullptr;        // its position in P
    int f_maxruns = 0;                // most runs in one group (0: no tables)
    // Work list of the pre-pass (null: one workgroup per source group).  Where one group holds several times the mean -- popular
    // columns of a set-covering LP: 8 % of the entries in the first of 490 groups -- its list is cut into chunks of f_work_cap
    // entries, a workgroup each: {group, first entry, one past the last, 0}, ordered by group.
    const int4 *f_work = nullptr;
    int n_work = 0;
};

// Host-side result of the analysis; perm arrays give, for every stored entry, its index in the CSR
// value array (-1 for padding) so that values can be refreshed on the device after scaling.
struct TiledHost {
    std::vector<int> sb_ptr, sb_mid;
    std::vector<TileStep> steps;
    // The entry arrays stay in the pieces the builder threads produced (one per contiguous range of super-blocks,
    // in order); the device copy is their concatenation -- the steps already carry the concatenated offsets.
    struct Piece {
        std::vector<uint32_t> tidx;
        std::vector<int> tperm;
        std::vector<int> rcol, rperm;
        std::vector<uint16_t> rrow;
    };
    std::vector<Piece> pieces;
    size_t n_tile = 0, n_rem = 0;  // total entries over the pieces
    long dense_entries = 0;        // CSR entries that landed in staged tiles
    long padding = 0;
};

// Builds the tiled structure of a CSR pattern (rows x cols).  Returns false (and leaves `out` empty)
// when the matrix is too small or too scattered for the tiled kernel to pay off.
bool build_tiled(int rows, int cols, const int *rowptr, const int *col, TiledHost *out, int min_rows,
                 double min_dense_fraction, int R = kTileRows, int T = kTileCols, int rem_cap = kTileRemCap);

// host only: build_tiled on the pattern + the invariants the kernels rely on (tiled.cpp); throws on the first violation
void tiled_host_check(int rows, int cols, const int *rp, const int *ci, int R, int T, double min_dense, long out[6]);

// col_c / map_c (device, compact nnz entries): the CSR pattern without the rows whose compact length is zero (tiled_build.hip)
void compact_without_rows(long nnz, int rows, const int *rp_dev, const int *rp_c_dev, const int *col_dev, int *col_c, int *map_c, hipStream_t s);

struct DeviceTiled {
    DBuf<int> sb_ptr, sb_mid, tperm, rperm, rcol;
    DBuf<TileStep> steps;
    DBuf<uint32_t> tidx;   // one code per entry: filled by the builders, released by pack_indices()
    DBuf<uint32_t> tidx3;  // packed codes, what the kernel reads
    DBuf<uint16_t> rrow;  // rcol / rrow / rperm: remainder entries in (super-block, row, CSR) order as the builders emit them;
                          // build_far() derives the two-sided lists from them and releases them
    DBuf<double> tval;
    DBuf<double> P, f_val;
    DBuf<uint32_t> rq;
    DBuf<int> f_gptr, f_pos, f_perm;
    DBuf<int> f_rptr, f_rk, f_rp;
    DBuf<int4> f_work;
    DBuf<uint16_t> f_lcol;
    TiledDev view;
    long n_tile = 0, n_rem = 0;
    double rem_top_share = 0.0;  // build_far: share of the remainder entries in the heaviest source groups that together cover 2 MB of the gathered vector
    long dense_entries = 0, padding = 0;
    int n_steps = 0;
    void upload(const TiledHost &h, int R = kTileRows, int T = kTileCols, int rem_cap = kTileRemCap);
    void pack_indices(hipStream_t s);  // tidx -> tidx3
    // remainder lists (rcol, rrow, rperm) -> the propagation-blocking lists with source groups of G columns
    void build_far(int cols, hipStream_t s, int G = kTileRows);
    // launch shape (persistent workgroups per XCD) and the rotation of every super-block's sweep; rot_period is the
    // alignment period in tiles (0 = no rotation)
    void finish_schedule(hipStream_t s);
    int rot_period = 0;
    DBuf<double> parts;
    DBuf<int> piece_ptr, slot_ptr;
    DBuf<unsigned long long> stamps;
    void dump_stamps() const;  // diagnostic: phase times of the piece form on stderr
    DBuf<unsigned long long> wgtimes;
    // long rows aside (TiledDev::side_*): built by set_side from the host row pointers
    DBuf<int4> side_blk, side_long;
    DBuf<int> side_rows;
    DBuf<double> side_partial, base;
    void set_side(int rows, const int *rowptr_host, const std::vector<int> &long_rows);
    // the copy was built from a CSR without the long rows: translate its value indices back (map: compact -> original position)
    void compose_perms(const int *map_dev, hipStream_t s);
    void dump_wgtimes() const;  // diagnostic: start / per-super-block / end times of the fused kernel's workgroups (last launch)
    DBuf<int4> segs;
    // Builds the same structure from the DEVICE CSR index arrays (tiled_build.hip); false: declined (too small,
    // too scattered, or too large for 32-bit entry offsets) and nothing is valid.
    bool build_on_device(int rows, int cols, long nnz, const int *rowptr, const int *col, int min_rows,
                         double min_dense_fraction, hipStream_t s, int R = kTileRows, int T = kTileCols, int rem_cap = kTileRemCap);
    // throws std::runtime_error naming the first difference between this (device-built) copy and the host builder's
    void compare_with(const TiledHost &h) const;
};

// tval[e] = csr_val[tperm[e]] (0 for padding), f_val likewise
void launch_tiled_refresh(const DeviceTiled &t, const double *csr_val, hipStream_t s);
// tval_log[e] = -log(max(|csr_val[tperm[e]]|, 1e-300)) (NaN for padding), fval_log likewise: what the Curtis-Reid passes sum
// when they run through the tiled kernel (kernels.hip: CrEpi)
void launch_tiled_refresh_log(const DeviceTiled &t, const double *csr_val, double *tval_log, double *fval_log, hipStream_t s);

}  // namespace hprlp
