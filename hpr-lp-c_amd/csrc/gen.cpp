// gen.cpp -- deterministic generator of the banded-random benchmark matrix (BASELINE.json config 5:
// "Synthetic random CSR 10M x 10M, ~200M nnz").  Benchmark/test utility, not part of the solve path.
// Every row is a pure function of (seed, row), so any rank can produce any row range on its own.
#include <algorithm>
#include <atomic>
#include <exception>
#include <cmath>
#include <cstdlib>
#include <cstdint>
#include <thread>
#include <vector>

#include "common.h"
#include "env.h"
#include "hprlp_amd.h"

namespace {

inline uint64_t mix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

struct RowRng {
    uint64_t state;
    RowRng(uint64_t seed, uint64_t row) : state(mix64(seed * 0x2545F4914F6CDD1DULL + row)) {}
    uint64_t next() { return state = mix64(state); }
    double uniform() { return static_cast<double>(next() >> 11) * (1.0 / 9007199254740992.0); }
    // uniform integer in [0, bound)
    uint64_t below(uint64_t bound) { return static_cast<uint64_t>(uniform() * static_cast<double>(bound)); }
    double normal() {
        const double u1 = (static_cast<double>(next() >> 11) + 1.0) * (1.0 / 9007199254740992.0);
        const double u2 = uniform();
        return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586476925286766559 * u2);
    }
};

// the sorted, distinct columns of global row `grow` into c[0 .. per_row); leaves rng in front of the row's value draws
inline void gen_row_columns(int m, int n, int per_row, int band, int width, double far_share, int grow, RowRng &rng, int *c) {
    const long center = (static_cast<long>(grow) * n) / m;
    long base = center - band;
    if (base < 0) base = 0;
    if (base > n - width) base = n - width;
    for (int k = 0; k < per_row; ++k) {
        const bool far = rng.uniform() < far_share;
        c[k] = far ? static_cast<int>(rng.below(static_cast<uint64_t>(n)))
                   : static_cast<int>(base + static_cast<long>(rng.below(static_cast<uint64_t>(width))));
    }
    std::sort(c, c + per_row);
    for (int k = 1; k < per_row; ++k)
        if (c[k] <= c[k - 1]) c[k] = c[k - 1] + 1;  // make the row's columns distinct
    if (c[per_row - 1] >= n) {                      // pushed past the last column: pack against it
        int top = n - 1;
        for (int k = per_row - 1; k >= 0 && c[k] > top; --k, --top) c[k] = top;
    }
}

// 5 % of a row's entries fall anywhere (BASELINE config 5); HPRLP_GEN_FAR overrides the share for kernel experiments
double gen_far_share() {
    static const double far_share = hprlp::env_get("HPRLP_GEN_FAR") ? std::atof(hprlp::env_get("HPRLP_GEN_FAR")) : 0.05;
    return far_share;
}

void gen_rows(int m, int n, int per_row, int band, uint64_t seed, int row0, int r_begin, int r_end, int *col,
              double *val) {
    const int width = std::min(2 * band + 1, n);
    const double far_share = gen_far_share();
    std::vector<int> c(static_cast<size_t>(per_row));
    for (int r = r_begin; r < r_end; ++r) {
        const int grow = row0 + r;
        RowRng rng(seed, static_cast<uint64_t>(grow));
        gen_row_columns(m, n, per_row, band, width, far_share, grow, rng, c.data());
        int *co = col + static_cast<size_t>(r) * per_row;
        double *vo = val + static_cast<size_t>(r) * per_row;
        for (int k = 0; k < per_row; ++k) {
            co[k] = c[k];
            vo[k] = rng.normal();
        }
    }
}

// entries of rows [r_begin, r_end) whose column lies in [col_lo, col_hi), in row order: (local column, row, value) appended
struct Kept {
    std::vector<int> lc, row;
    std::vector<double> val;
};
void gen_rows_keep_columns(int m, int n, int per_row, int band, uint64_t seed, int r_begin, int r_end, int col_lo, int col_hi, Kept *out) {
    const int width = std::min(2 * band + 1, n);
    const double far_share = gen_far_share();
    std::vector<int> c(static_cast<size_t>(per_row));
    {   // expected share of this row range's entries, with some room: no reallocation of three large vectors on the way
        const double share = static_cast<double>(col_hi - col_lo) / n;
        const size_t guess = static_cast<size_t>(static_cast<double>(r_end - r_begin) * per_row * share * 1.05) + 4096;
        out->lc.reserve(guess); out->row.reserve(guess); out->val.reserve(guess);
    }
    for (int grow = r_begin; grow < r_end; ++grow) {
        RowRng rng(seed, static_cast<uint64_t>(grow));
        gen_row_columns(m, n, per_row, band, width, far_share, grow, rng, c.data());
        // (columns ascend: the owned ones are a contiguous run)
        const int kb = static_cast<int>(std::lower_bound(c.begin(), c.end(), col_lo) - c.begin());
        const int ke = static_cast<int>(std::lower_bound(c.begin(), c.end(), col_hi) - c.begin());
        if (kb >= ke) continue;  // nothing of this row here: its value draws are never made
        for (int k = 0; k < kb; ++k) { rng.next(); rng.next(); }  // a value draw is two steps of the row's stream (RowRng::normal)
        for (int k = kb; k < ke; ++k) {
            out->lc.push_back(c[k] - col_lo);
            out->row.push_back(grow);
            out->val.push_back(rng.normal());
        }
    }
}

}  // namespace

extern "C" int hprlp_gen_banded_csr(int m, int n, int per_row, int band, unsigned long long seed, int row0, int rows,
                                    int *rowptr, int *col, double *val, int nthreads) {
    try {
        if (m <= 0 || n <= 0 || per_row <= 0 || per_row > n || band < 0 || row0 < 0 || rows < 0 || row0 + rows > m ||
            static_cast<long>(rows) * per_row > 2147483647L)
            throw std::runtime_error("hprlp_gen_banded_csr: bad arguments");
        for (int r = 0; r <= rows; ++r) rowptr[r] = r * per_row;
        if (nthreads <= 0) nthreads = static_cast<int>(std::max(1u, std::thread::hardware_concurrency()));
        nthreads = std::min(nthreads, std::max(1, rows / 4096));
        std::vector<std::thread> th;
        const int chunk = (rows + nthreads - 1) / std::max(nthreads, 1);
        for (int t = 0; t < nthreads; ++t) {
            const int b = t * chunk, e = std::min(rows, b + chunk);
            if (b >= e) break;
            th.emplace_back(gen_rows, m, n, per_row, band, static_cast<uint64_t>(seed), row0, b, e, col, val);
        }
        for (auto &t : th) t.join();
        return 0;
    } catch (const std::exception &e) {
        hprlp::set_last_error(e.what());
        return -1;
    }
}

// Rows [col_off, col_off + n_loc) of the TRANSPOSE of the same matrix -- what the owner of those columns holds in a row-partitioned
// run (hprlp_shard::AT_*) -- without anybody holding the matrix: the generator is a pure function of (seed, row), so the caller
// sweeps all m rows (threads over row ranges) and keeps the entries whose column it owns; a stable counting sort by column then
// leaves the rows ascending inside every column, i.e. exactly the stable transpose the reference builds on one host
// (reference src/utils.cu:203-232).  trp: n_loc + 1 entries.  *tci_out / *tv_out: malloc'd, the caller releases them with
// hprlp_host_free.  No communication: this replaces round 2's all-to-all of (column, row, value) triples.
extern "C" int hprlp_gen_banded_csr_transposed(int m, int n, int per_row, int band, unsigned long long seed, int col_off, int n_loc,
                                               int *trp, int **tci_out, double **tv_out, long *nnz_out, int nthreads) {
    try {
        if (m <= 0 || n <= 0 || per_row <= 0 || per_row > n || band < 0 || col_off < 0 || n_loc < 0 || col_off + n_loc > n || !trp ||
            !tci_out || !tv_out || !nnz_out)
            throw std::runtime_error("hprlp_gen_banded_csr_transposed: bad arguments");
        if (nthreads <= 0) nthreads = static_cast<int>(std::max(1u, std::thread::hardware_concurrency()));
        nthreads = std::min(nthreads, std::max(1, m / 4096));
        // row chunks claimed from a counter (a column range's near entries sit in a few row chunks: static row ranges would
        // leave most threads with the 5 % far entries only); chunk order = row order
        const int chunk = 1 << 16;
        const int nchunks = (m + chunk - 1) / chunk;
        std::vector<Kept> kept(static_cast<size_t>(nchunks));
        std::atomic<int> next_chunk{0};
        {
            // (an exception inside a raw thread would end the process: each thread keeps its own and the first one is rethrown here)
            std::vector<std::thread> th;
            std::vector<std::exception_ptr> failed(static_cast<size_t>(nthreads));
            for (int t = 0; t < nthreads; ++t)
                th.emplace_back([&, t]() {
                    try {
                        for (int q = next_chunk.fetch_add(1); q < nchunks; q = next_chunk.fetch_add(1))
                            gen_rows_keep_columns(m, n, per_row, band, static_cast<uint64_t>(seed), q * chunk, std::min(m, (q + 1) * chunk),
                                                  col_off, col_off + n_loc, &kept[q]);
                    } catch (...) {
                        failed[t] = std::current_exception();
                        next_chunk.store(nchunks);  // the others stop at their next claim
                    }
                });
            for (auto &t : th) t.join();
            for (const std::exception_ptr &e : failed)
                if (e) std::rethrow_exception(e);
        }
        long total = 0;
        for (const Kept &k : kept) total += static_cast<long>(k.lc.size());
        if (total > 2147483647L) throw std::runtime_error("hprlp_gen_banded_csr_transposed: more than 2^31 - 1 entries in one shard");
        for (int j = 0; j <= n_loc; ++j) trp[j] = 0;
        for (const Kept &k : kept)
            for (int lc : k.lc) ++trp[lc + 1];
        for (int j = 0; j < n_loc; ++j) trp[j + 1] += trp[j];
        int *tci = static_cast<int *>(std::malloc(sizeof(int) * static_cast<size_t>(std::max<long>(total, 1))));
        double *tv = static_cast<double *>(std::malloc(sizeof(double) * static_cast<size_t>(std::max<long>(total, 1))));
        if (!tci || !tv) {
            std::free(tci);
            std::free(tv);
            throw std::runtime_error("hprlp_gen_banded_csr_transposed: out of host memory");
        }
        // stable scatter, threads over disjoint column ranges (each walks all chunks in row order and takes its columns)
        {
            std::vector<int> next(trp, trp + n_loc);
            const int parts = std::max(1, std::min(nthreads, n_loc / 4096));
            std::vector<std::thread> th;
            for (int t = 0; t < parts; ++t) {
                const int lo = static_cast<int>(static_cast<long>(n_loc) * t / parts), hi = static_cast<int>(static_cast<long>(n_loc) * (t + 1) / parts);
                th.emplace_back([&, lo, hi]() {
                    for (const Kept &k : kept)
                        for (size_t e = 0; e < k.lc.size(); ++e) {
                            const int lc = k.lc[e];
                            if (lc < lo || lc >= hi) continue;
                            const int q = next[lc]++;
                            tci[q] = k.row[e];
                            tv[q] = k.val[e];
                        }
                });
            }
            for (auto &t : th) t.join();
        }
        *tci_out = tci;
        *tv_out = tv;
        *nnz_out = total;
        return 0;
    } catch (const std::exception &e) {
        hprlp::set_last_error(e.what());
        return -1;
    }
}

extern "C" void hprlp_host_free(void *p) { std::free(p); }

// P A Q for the benchmark's permuted workload: out row i = row row_new2old[i] of A with columns renumbered by
// col_old2new and sorted.  Multi-threaded over row ranges; out arrays sized like the inputs.  Not part of the solve path.
extern "C" int hprlp_permute_csr_host(int m, int n, const int *rowptr, const int *col, const double *val, const int *row_new2old,
                                      const int *col_old2new, int *rowptr_out, int *col_out, double *val_out, int nthreads) {
    try {
        if (m <= 0 || n <= 0 || !rowptr || !col || !val || !row_new2old || !col_old2new || !rowptr_out || !col_out || !val_out)
            throw std::runtime_error("hprlp_permute_csr_host: bad arguments");
        rowptr_out[0] = 0;
        for (int i = 0; i < m; ++i) {
            const int o = row_new2old[i];
            if (o < 0 || o >= m) throw std::runtime_error("hprlp_permute_csr_host: row permutation out of range");
            rowptr_out[i + 1] = rowptr_out[i] + (rowptr[o + 1] - rowptr[o]);
        }
        if (nthreads <= 0) nthreads = static_cast<int>(std::max(1u, std::thread::hardware_concurrency()));
        nthreads = std::min(nthreads, std::max(1, m / 4096));
        auto work = [&](int b, int e) {
            std::vector<std::pair<int, double>> row;
            for (int i = b; i < e; ++i) {
                const int o = row_new2old[i];
                row.clear();
                for (int k = rowptr[o]; k < rowptr[o + 1]; ++k) row.emplace_back(col_old2new[col[k]], val[k]);
                std::sort(row.begin(), row.end(), [](const std::pair<int, double> &a, const std::pair<int, double> &b2) { return a.first < b2.first; });
                int p = rowptr_out[i];
                for (auto &pr : row) {
                    col_out[p] = pr.first;
                    val_out[p++] = pr.second;
                }
            }
        };
        std::vector<std::thread> th;
        const int chunk = (m + nthreads - 1) / nthreads;
        for (int t = 0; t < nthreads; ++t) {
            const int b = t * chunk, e = std::min(m, b + chunk);
            if (b >= e) break;
            th.emplace_back(work, b, e);
        }
        for (auto &t : th) t.join();
        return 0;
    } catch (const std::exception &e) {
        hprlp::set_last_error(e.what());
        return -1;
    }
}
