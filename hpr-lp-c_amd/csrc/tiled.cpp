// tiled.cpp -- host-side construction of the column-tiled matrix copy (see tiled.h).
#include "tiled.h"
#include "env.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <type_traits>

namespace hprlp {

namespace {

struct Local {  // what one builder thread produces for a contiguous range of super-blocks
    std::vector<int> sb_steps, sb_mid;  // per super-block: number of steps, index of first remainder step (local)
    std::vector<TileStep> steps;        // e_begin/e_end local to this thread's tile / remainder arrays
    std::vector<char> step_is_rem;
    std::vector<uint32_t> tidx;
    std::vector<int> tperm, rcol, rperm;
    std::vector<uint16_t> rrow;
    long dense = 0, pad = 0;
};

void build_range(int rows, int cols, const int *rp, const int *ci, int sb0, int sb1, Local *L, int R, int T, int rem_cap) {
    const int K = kTileChunk;
    const int ntile = (cols + T - 1) / T;
    std::vector<int> cnt(static_cast<size_t>(ntile), 0), slot(static_cast<size_t>(ntile), -1);
    std::vector<int> touched;
    struct Ent {
        int row, lcol, k;
    };
    std::vector<Ent> ents;                 // dense entries of the current super-block, grouped by tile
    std::vector<int> b_begin, b_cursor;    // per dense tile slot: first entry / fill cursor in `ents`
    std::vector<std::pair<int, int>> rem, rem_long;  // (local row, csr index)
    {   // one allocation per output array instead of doubling growth (page faults and copies dominate otherwise)
        const size_t r_lo = static_cast<size_t>(sb0) * R, r_hi = std::min<size_t>(static_cast<size_t>(rows), static_cast<size_t>(sb1) * R);
        const size_t nz = static_cast<size_t>(rp[r_hi] - rp[r_lo]);
        L->tidx.reserve(nz + nz / 8 + 64);
        L->tperm.reserve(nz + nz / 8 + 64);
        L->rcol.reserve(nz / 8 + 64);
        L->rperm.reserve(nz / 8 + 64);
        L->rrow.reserve(nz / 8 + 64);
    }
    for (int sb = sb0; sb < sb1; ++sb) {
        const int r0 = sb * R, r1 = std::min(rows, r0 + R);
        touched.clear();
        for (int k = rp[r0]; k < rp[r1]; ++k) {
            const int tl = ci[k] / T;
            if (cnt[tl]++ == 0) touched.push_back(tl);
        }
        std::sort(touched.begin(), touched.end());
        int nd = 0;
        b_begin.clear();
        int dense_total = 0;
        for (int tl : touched) {
            if (rem_cap != kPbRemCap && cnt[tl] >= kTileDenseMin) {  // (the all-remainder form stages no tile at all)
                slot[tl] = nd++;
                b_begin.push_back(dense_total);
                dense_total += cnt[tl];
            } else {
                slot[tl] = -1;
            }
        }
        b_begin.push_back(dense_total);
        b_cursor.assign(b_begin.begin(), b_begin.end() - 1);
        if (ents.size() < static_cast<size_t>(dense_total)) ents.resize(static_cast<size_t>(dense_total));
        rem.clear();
        rem_long.clear();
        for (int r = r0; r < r1; ++r)
            for (int k = rp[r]; k < rp[r + 1]; ++k) {
                const int tl = ci[k] / T;
                const int sl = slot[tl];
                if (sl >= 0) ents[b_cursor[sl]++] = Ent{r - r0, ci[k] - tl * T, k};
                else rem.emplace_back(r - r0, k);
            }
        const size_t first_step = L->steps.size();
        for (int tl : touched) {
            if (slot[tl] < 0) continue;
            const Ent *b = ents.data() + b_begin[slot[tl]];
            const size_t bn = static_cast<size_t>(b_begin[slot[tl] + 1] - b_begin[slot[tl]]);
            const size_t tile_begin = L->tidx.size();
            // a row's segment is cut into pieces of at most K entries, piece q goes to layer q of the tile's list (tiled.h: kTileLayers);
            // every layer is laid down by the bin-packed layout its piece counts define (PackLayout)
            int cnt[kTileLayers][5] = {};
            for (size_t i = 0; i < bn;) {
                size_t j = i;
                while (j < bn && b[j].row == b[i].row) ++j;
                const int len = static_cast<int>(j - i);
                for (int q = 0; q < kTileLayers && q * K < len; ++q) ++cnt[q][std::min(K, len - q * K)];
                i = j;
            }
            int nl = 1;
            auto layer_entries = [&](int q) { return cnt[q][1] + 2 * cnt[q][2] + 3 * cnt[q][3] + 4 * cnt[q][4]; };
            while (nl < kTileLayers && layer_entries(nl) >= kTileLayerMin) ++nl;
            PackLayout lay[kTileLayers];
            size_t layer_at[kTileLayers + 1];
            layer_at[0] = tile_begin;
            for (int q = 0; q < nl; ++q) {
                lay[q].set(cnt[q][1], cnt[q][2], cnt[q][3], cnt[q][4]);
                layer_at[q + 1] = layer_at[q] + static_cast<size_t>(lay[q].entries());
                L->dense += layer_entries(q);
                L->pad += lay[q].entries() - layer_entries(q);
            }
            L->tidx.resize(layer_at[nl], 0u);
            L->tperm.resize(layer_at[nl], -1);
            int rank[kTileLayers][5] = {};
            for (size_t i = 0; i < bn;) {
                size_t j = i;
                while (j < bn && b[j].row == b[i].row) ++j;
                const int len = static_cast<int>(j - i);
                for (int q = 0; q * K < len; ++q) {
                    const size_t p0 = i + static_cast<size_t>(q) * K;
                    if (q >= nl) {  // pieces beyond the kept layers: remainder
                        for (size_t e = p0; e < j; ++e) rem_long.emplace_back(b[e].row, b[e].k);
                        break;
                    }
                    const int pl = std::min(K, len - q * K);
                    const int rk = rank[q][pl]++;
                    const size_t at = layer_at[q] + static_cast<size_t>(lay[q].pos(pl, rk));
                    for (int e = 0; e < pl; ++e) {
                        L->tidx[at + static_cast<size_t>(e)] = tile_code(b[p0 + e].lcol, b[p0 + e].row);
                        L->tperm[at + static_cast<size_t>(e)] = b[p0 + e].k;
                    }
                    const int np = lay[q].pads_after(pl, rk);
                    for (int e = 0; e < np; ++e) {
                        L->tidx[at + static_cast<size_t>(pl + e)] = static_cast<uint32_t>(b[i].row);
                        L->tperm[at + static_cast<size_t>(pl + e)] = -1;
                    }
                }
                i = j;
            }
            for (int q = 0; q < nl; ++q) {   // every layer has its own steps
                size_t p = layer_at[q];
                while (p < layer_at[q + 1]) {
                    const size_t c = std::min<size_t>(layer_at[q + 1] - p, kTileStepCap);
                    L->steps.push_back(TileStep{tl * T, static_cast<int>(p), static_cast<int>(p + c), 0});
                    L->step_is_rem.push_back(0);
                    p += c;
                }
            }
        }
        L->sb_mid.push_back(static_cast<int>(L->steps.size() - first_step));
        // the far entries were collected in (row, csr index) order; only the long segments need sorting before the merge
        if (!rem_long.empty()) {
            std::sort(rem_long.begin(), rem_long.end());
            const size_t mid = rem.size();
            rem.insert(rem.end(), rem_long.begin(), rem_long.end());
            std::inplace_merge(rem.begin(), rem.begin() + static_cast<long>(mid), rem.end());
        }
        size_t p = L->rcol.size();
        for (const auto &e : rem) {
            L->rrow.push_back(static_cast<uint16_t>(e.first));
            L->rcol.push_back(ci[e.second]);
            L->rperm.push_back(e.second);
        }
        const size_t end = L->rcol.size();
        while (p < end) {
            const size_t c = std::min<size_t>(end - p, static_cast<size_t>(rem_cap));
            L->steps.push_back(TileStep{0, static_cast<int>(p), static_cast<int>(p + c), 0});
            L->step_is_rem.push_back(1);
            p += c;
        }
        L->sb_steps.push_back(static_cast<int>(L->steps.size() - first_step));
        for (int tl : touched) {
            cnt[tl] = 0;
            slot[tl] = -1;
        }
    }
}

}  // namespace

bool build_tiled(int rows, int cols, const int *rowptr, const int *col, TiledHost *out, int min_rows,
                 double min_dense_fraction, int R, int T, int rem_cap) {
    *out = TiledHost();
    if (rows < min_rows || rows <= 0 || cols <= 0) return false;
    const long nnz = rowptr[rows];
    if (nnz <= 0) return false;
    if (R < 64 || R > kTileRows || R % 64 != 0) throw std::runtime_error("tiled build: unsupported super-block height");
    if (T != kTileCols && T != kTileColsNarrow) throw std::runtime_error("tiled build: unsupported tile width");
    if (rem_cap != kTileRemCap && rem_cap != kPbRemCap) throw std::runtime_error("tiled build: unsupported remainder step size");
    if (rem_cap == kPbRemCap && R > kPbRowsMax) throw std::runtime_error("tiled build: the all-remainder form takes super-blocks of at most 4096 rows");
    const int nsb = (rows + R - 1) / R;
    // 16: a GPU's usual share of the host's cores (measured on the 16-CPU quota of the test box: 8 threads 1.01 s
    // of set-up, 16 0.69 s, 32 0.73 s, 64 0.70 s)
    int nt = static_cast<int>(std::min(16u, std::max(1u, std::thread::hardware_concurrency())));
    if (const char *e = env_get("HPRLP_TILE_THREADS")) nt = std::max(1, std::atoi(e));
    nt = std::min(nt, std::max(1, nsb / 4));
    std::vector<Local> loc(static_cast<size_t>(nt));
    std::vector<std::thread> th;
    const int per = (nsb + nt - 1) / nt;
    for (int t = 0; t < nt; ++t) {
        const int a = t * per, b = std::min(nsb, a + per);
        if (a >= b) break;
        th.emplace_back(build_range, rows, cols, rowptr, col, a, b, &loc[t], R, T, rem_cap);
    }
    for (auto &t : th) t.join();
    long dense = 0, pad = 0;
    size_t n_t = 0, n_r = 0, n_s = 0;
    for (const Local &L : loc) {
        dense += L.dense;
        pad += L.pad;
        n_t += L.tidx.size();
        n_r += L.rcol.size();
        n_s += L.steps.size();
    }
    if (static_cast<double>(dense) < min_dense_fraction * static_cast<double>(nnz) || n_t > 2000000000UL ||
        n_r > 2000000000UL)
        return false;
    out->dense_entries = dense;
    out->padding = pad;
    out->sb_ptr.reserve(static_cast<size_t>(nsb) + 1);
    out->sb_mid.reserve(nsb);
    out->steps.reserve(n_s);
    out->pieces.reserve(loc.size());
    out->sb_ptr.push_back(0);
    for (Local &L : loc) {
        const int t_off = static_cast<int>(out->n_tile), r_off = static_cast<int>(out->n_rem);
        size_t sp = 0;
        for (size_t i = 0; i < L.sb_steps.size(); ++i) {
            const int base = out->sb_ptr.back();
            out->sb_mid.push_back(base + L.sb_mid[i]);
            out->sb_ptr.push_back(base + L.sb_steps[i]);
            for (int q = 0; q < L.sb_steps[i]; ++q, ++sp) {
                TileStep s = L.steps[sp];
                const int off = L.step_is_rem[sp] ? r_off : t_off;
                s.e_begin += off;
                s.e_end += off;
                out->steps.push_back(s);
            }
        }
        out->n_tile += L.tidx.size();
        out->n_rem += L.rcol.size();
        TiledHost::Piece pc;  // moved, not copied: 1.8 GB on the 2e8-nonzero matrix
        pc.tidx = std::move(L.tidx);
        pc.tperm = std::move(L.tperm);
        pc.rcol = std::move(L.rcol);
        pc.rperm = std::move(L.rperm);
        pc.rrow = std::move(L.rrow);
        out->pieces.push_back(std::move(pc));
        L = Local();  // release
    }
    return true;
}

// Host only: build the tiled structure of a pattern and verify what the kernels rely on -- every CSR entry exactly once (in a
// tile list or in the remainder), codes consistent with the entries they stand for, inside a step one chunk per accumulator at
// most (a row appears in ONE chunk of a step: padding slots carry the row of the slot before them), steps within their capacity
// and tile, at most kTileLayers steps-runs per (super-block, tile).  out = {tile entries incl. padding, remainder entries, steps,
// padding, most consecutive steps of one tile in a super-block (its layers' steps), share of entries staged x 1e6}.
void tiled_host_check(int rows, int cols, const int *rp, const int *ci, int R, int T, double min_dense, long out[6]) {
    TiledHost h;
    if (!build_tiled(rows, cols, rp, ci, &h, 1, min_dense, R, T, kTileRemCap)) throw std::runtime_error("tiled check: the build declined the pattern");
    const long nnz = rp[rows];
    std::vector<uint32_t> tidx;
    std::vector<int> tperm, rperm;
    std::vector<uint16_t> rrow;
    for (const TiledHost::Piece &pc : h.pieces) {
        tidx.insert(tidx.end(), pc.tidx.begin(), pc.tidx.end());
        tperm.insert(tperm.end(), pc.tperm.begin(), pc.tperm.end());
        rperm.insert(rperm.end(), pc.rperm.begin(), pc.rperm.end());
        rrow.insert(rrow.end(), pc.rrow.begin(), pc.rrow.end());
    }
    std::vector<char> seen(static_cast<size_t>(nnz), 0);
    auto row_of = [&](int k) { return static_cast<int>(std::upper_bound(rp, rp + rows + 1, k) - rp) - 1; };
    const int nsb = static_cast<int>(h.sb_mid.size());
    long pads = 0, most_run = 0;
    std::vector<int> stamp(static_cast<size_t>(R), -1);
    for (int sb = 0; sb < nsb; ++sb) {
        long run = 0;
        for (int s = h.sb_ptr[sb]; s < h.sb_mid[sb]; ++s) {
            const TileStep &st = h.steps[s];
            if (st.e_end - st.e_begin > kTileStepCap || st.e_begin % kTileChunk != 0 || st.col0 % T != 0) throw std::runtime_error("tiled check: bad tile step");
            run = (s > h.sb_ptr[sb] && h.steps[s - 1].col0 == st.col0) ? run + 1 : 1;
            most_run = std::max(most_run, run);
            for (int e = st.e_begin; e < st.e_end; ++e) {
                const int lrow = static_cast<int>(tidx[e] & (kTileRows - 1)), lcol = static_cast<int>(tidx[e] >> kTileRowBits);
                if (lrow >= R) throw std::runtime_error("tiled check: local row beyond the super-block");
                const int k = tperm[e];
                if (k < 0) {
                    ++pads;
                    if (e % kTileChunk == 0 || static_cast<int>(tidx[e - 1] & (kTileRows - 1)) != lrow) throw std::runtime_error("tiled check: padding does not continue its row");
                    continue;
                }
                if (k >= nnz || seen[k]++) throw std::runtime_error("tiled check: an entry twice (or out of range)");
                if (row_of(k) != sb * R + lrow || ci[k] != st.col0 + lcol) throw std::runtime_error("tiled check: a code does not name its entry");
            }
            // one chunk per accumulator inside the step
            for (int e = st.e_begin; e < st.e_end; ++e) {
                const int lrow = static_cast<int>(tidx[e] & (kTileRows - 1)), chunk = e / kTileChunk;
                if (stamp[lrow] == -1) stamp[lrow] = chunk;
                else if (stamp[lrow] != chunk) throw std::runtime_error("tiled check: a row in two chunks of one step");
            }
            for (int e = st.e_begin; e < st.e_end; ++e) stamp[tidx[e] & (kTileRows - 1)] = -1;
        }
        for (int s = h.sb_mid[sb]; s < h.sb_ptr[sb + 1]; ++s) {
            const TileStep &st = h.steps[s];
            if (st.e_end - st.e_begin > kTileRemCap) throw std::runtime_error("tiled check: remainder step too long");
            for (int e = st.e_begin; e < st.e_end; ++e) {
                const int k = rperm[e];
                if (k < 0 || k >= nnz || seen[k]++) throw std::runtime_error("tiled check: a remainder entry twice (or out of range)");
                if (row_of(k) != sb * R + rrow[e]) throw std::runtime_error("tiled check: a remainder entry in the wrong super-block / row");
            }
        }
    }
    for (long k = 0; k < nnz; ++k)
        if (!seen[k]) throw std::runtime_error("tiled check: an entry is missing");
    if (most_run > kTileLayers * ((static_cast<long>(R) * kTileChunk + kTileStepCap - 1) / kTileStepCap + 1)) throw std::runtime_error("tiled check: too many steps for one tile");
    out[0] = static_cast<long>(h.n_tile);
    out[1] = static_cast<long>(h.n_rem);
    out[2] = static_cast<long>(h.steps.size());
    out[3] = pads;
    out[4] = most_run;
    out[5] = static_cast<long>(1e6 * static_cast<double>(h.dense_entries) / static_cast<double>(nnz));
}

void DeviceTiled::upload(const TiledHost &h, int R, int T, int rem_cap) {
    n_tile = static_cast<long>(h.n_tile);
    n_rem = static_cast<long>(h.n_rem);
    n_steps = static_cast<int>(h.steps.size());
    const int nsb = static_cast<int>(h.sb_mid.size());
    sb_ptr.alloc(h.sb_ptr.size()); sb_ptr.upload(h.sb_ptr.data(), h.sb_ptr.size());
    sb_mid.alloc(h.sb_mid.size()); sb_mid.upload(h.sb_mid.data(), h.sb_mid.size());
    steps.alloc(h.steps.size()); steps.upload(h.steps.data(), h.steps.size());
    // +8 entries of slack: the kernel's clamped 16-byte loads never read past e_begin+3 of a valid
    // chunk, the slack only keeps empty arrays addressable
    tidx.alloc_zero(h.n_tile + 8);
    tperm.alloc(h.n_tile + 8);
    tval.alloc_zero(h.n_tile + 8);
    rcol.alloc_zero(h.n_rem + 8);
    rperm.alloc(h.n_rem + 8);
    rrow.alloc_zero(h.n_rem + 8);
    size_t to = 0, ro = 0;
    for (const TiledHost::Piece &pc : h.pieces) {
        if (!pc.tidx.empty()) {
            HIP_CHECK(hipMemcpy(tidx.p + to, pc.tidx.data(), pc.tidx.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            HIP_CHECK(hipMemcpy(tperm.p + to, pc.tperm.data(), pc.tperm.size() * sizeof(int), hipMemcpyHostToDevice));
        }
        if (!pc.rcol.empty()) {
            HIP_CHECK(hipMemcpy(rcol.p + ro, pc.rcol.data(), pc.rcol.size() * sizeof(int), hipMemcpyHostToDevice));
            HIP_CHECK(hipMemcpy(rperm.p + ro, pc.rperm.data(), pc.rperm.size() * sizeof(int), hipMemcpyHostToDevice));
            HIP_CHECK(hipMemcpy(rrow.p + ro, pc.rrow.data(), pc.rrow.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        }
        to += pc.tidx.size();
        ro += pc.rcol.size();
    }
    view.valid = true;
    view.R = R;
    view.T = T;
    view.rem_cap = rem_cap;
    view.nsb = nsb;
    view.sb_ptr = sb_ptr.p;
    view.sb_mid = sb_mid.p;
    view.steps = steps.p;
    view.tval = tval.p;
    pack_indices(nullptr);
    view.tidx3 = tidx3.p;
    finish_schedule(nullptr);
}

void DeviceTiled::compare_with(const TiledHost &h) const {
    auto fail = [](const std::string &what) { throw std::runtime_error("tiling check: " + what); };
    if (n_tile != static_cast<long>(h.n_tile)) fail("tile entries " + std::to_string(n_tile) + " vs " + std::to_string(h.n_tile));
    if (n_rem != static_cast<long>(h.n_rem)) fail("remainder entries " + std::to_string(n_rem) + " vs " + std::to_string(h.n_rem));
    if (dense_entries != h.dense_entries || padding != h.padding) fail("dense / padding totals");
    auto same = [&](const char *name, const auto &dbuf, const auto &host) {
        using T = typename std::decay<decltype(host[0])>::type;
        std::vector<T> dev(host.size());
        if (!host.empty()) HIP_CHECK(hipMemcpy(dev.data(), dbuf.p, host.size() * sizeof(T), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < host.size(); ++i)
            if (std::memcmp(&dev[i], &host[i], sizeof(T)) != 0) fail(std::string(name) + " differs at " + std::to_string(i));
    };
    same("sb_ptr", sb_ptr, h.sb_ptr);
    same("sb_mid", sb_mid, h.sb_mid);
    if (static_cast<size_t>(n_steps) != h.steps.size()) fail("step count");
    {  // the steps without the rotation offsets, which finish_schedule() adds on the device for either builder
        std::vector<TileStep> dev(h.steps.size());
        if (!dev.empty()) HIP_CHECK(hipMemcpy(dev.data(), steps.p, dev.size() * sizeof(TileStep), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < dev.size(); ++i)
            if (dev[i].col0 != h.steps[i].col0 || dev[i].e_begin != h.steps[i].e_begin || dev[i].e_end != h.steps[i].e_end)
                fail("steps differ at " + std::to_string(i));
    }
    std::vector<uint32_t> a_tidx;
    std::vector<int> a_tperm, a_rcol, a_rperm;
    std::vector<uint16_t> a_rrow;
    for (const TiledHost::Piece &pc : h.pieces) {
        a_tidx.insert(a_tidx.end(), pc.tidx.begin(), pc.tidx.end());
        a_tperm.insert(a_tperm.end(), pc.tperm.begin(), pc.tperm.end());
        a_rcol.insert(a_rcol.end(), pc.rcol.begin(), pc.rcol.end());
        a_rperm.insert(a_rperm.end(), pc.rperm.begin(), pc.rperm.end());
        a_rrow.insert(a_rrow.end(), pc.rrow.begin(), pc.rrow.end());
    }
    {   // the device copy holds the packed form only
        std::vector<uint32_t> packed(a_tidx.size() / 4 * 3);
        for (size_t c = 0; c < a_tidx.size() / 4; ++c) {
            const uint32_t e0 = a_tidx[4 * c], e1 = a_tidx[4 * c + 1], e2 = a_tidx[4 * c + 2], e3 = a_tidx[4 * c + 3];
            packed[3 * c] = e0 | (e1 << 24);
            packed[3 * c + 1] = (e1 >> 8) | (e2 << 16);
            packed[3 * c + 2] = (e2 >> 16) | (e3 << 8);
        }
        same("tidx3", tidx3, packed);
    }
    same("tperm", tperm, a_tperm);
    same("rcol", rcol, a_rcol);
    same("rperm", rperm, a_rperm);
    same("rrow", rrow, a_rrow);
}

}  // namespace hprlp
