// host_model.cpp -- host side of the boundary: model construction/destruction and small host
// utilities.  Mirrors the behaviour of reference src/HPRLP.cu:321-446,529-537 and
// src/mps_reader.cpp:1397-1510 (deep copies, NULL + stderr message on bad input).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <iostream>
#include <thread>

#include "HPRLP.h"
#include "common.h"

namespace hprlp {

static thread_local std::string g_last_error;
void set_last_error(const std::string &msg) { g_last_error = msg; }
const char *last_error_cstr() { return g_last_error.c_str(); }

void csr_transpose_host(int rows, int cols, long nnz, const int *rp, const int *ci, const double *v,
                        std::vector<int> &trp, std::vector<int> &tci, std::vector<double> &tv) {
    (void)nnz;
    csr_transpose_range_host(rows, 0, cols, rp, ci, v, trp, tci, tv);
}

// Rows [c0, c1) of the transpose (= columns [c0, c1) of the matrix), stable in row order: a counting sort over the
// entries whose column lies in the range.  Large matrices: T threads over contiguous row ranges, each with its own
// column histogram, which gives every thread its own write cursor per column, ordered by thread = ordered by row,
// so the result is identical to the sequential sort.  A rank of the row-partitioned solve calls this with its own
// column range only (hprlp_extract_shard): no rank ever builds the whole transpose.
void csr_transpose_range_host(int rows, int c0, int c1, const int *rp, const int *ci, const double *v,
                              std::vector<int> &trp, std::vector<int> &tci, std::vector<double> &tv) {
    const int nc = std::max(0, c1 - c0);
    const long nnz = rp[rows];
    trp.assign(static_cast<size_t>(nc) + 1, 0);
    int T = 1;
    if (nnz > 4000000) {
        T = static_cast<int>(std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency())));
        while (T > 1 && static_cast<size_t>(T) * static_cast<size_t>(nc) > 400000000UL) --T;
    }
    const int chunk = (rows + T - 1) / std::max(T, 1);
    std::vector<std::vector<int>> hist(static_cast<size_t>(T));
    auto count = [&](int t) {
        hist[t].assign(static_cast<size_t>(nc), 0);
        const int r0 = std::min(rows, t * chunk), r1 = std::min(rows, r0 + chunk);
        for (int k = rp[r0]; k < rp[r1]; ++k) {
            const int j = ci[k] - c0;
            if (j >= 0 && j < nc) hist[t][j]++;
        }
    };
    auto fill = [&](int t) {
        const int r0 = std::min(rows, t * chunk), r1 = std::min(rows, r0 + chunk);
        std::vector<int> &next = hist[t];
        for (int i = r0; i < r1; ++i)
            for (int k = rp[i]; k < rp[i + 1]; ++k) {
                const int j = ci[k] - c0;
                if (j < 0 || j >= nc) continue;
                const int pos = next[j]++;
                tv[pos] = v[k];
                tci[pos] = i;
            }
    };
    auto run = [&](auto &&fn) {
        if (T == 1) {
            fn(0);
            return;
        }
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back(fn, t);
        for (auto &x : th) x.join();
    };
    run(count);
    long total = 0;
    for (int j = 0; j < nc; ++j) {  // cursor of (column j, thread t) = start of column j + entries of earlier threads
        trp[j] = static_cast<int>(total);
        for (int t = 0; t < T; ++t) {
            const int c = hist[t][j];
            hist[t][j] = static_cast<int>(total);
            total += c;
        }
    }
    trp[nc] = static_cast<int>(total);
    tci.resize(static_cast<size_t>(total));
    tv.resize(static_cast<size_t>(total));
    run(fill);
}

static inline uint64_t splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// Replaces curandGenerateNormalDouble(seed 1) + add_epsilon (reference src/power_iteration.cu:44-57)
// with a counter RNG that any rank can evaluate for its own slice of rows.
void power_start_vector(int m, unsigned long long seed, long long offset, double *z) {
    const double two53 = 1.0 / 9007199254740992.0;
    const double twopi = 6.283185307179586476925286766559;
    const uint64_t base = static_cast<uint64_t>(seed) * 0x100000001B3ULL;
    auto fill = [=](int i0, int i1) {
        for (int i = i0; i < i1; ++i) {
            const uint64_t g = static_cast<uint64_t>(offset + i);
            const uint64_t h1 = splitmix64(base + 2 * g);
            const uint64_t h2 = splitmix64(base + 2 * g + 1);
            const double u1 = static_cast<double>((h1 >> 11) + 1) * two53;
            const double u2 = static_cast<double>(h2 >> 11) * two53;
            z[i] = std::sqrt(-2.0 * std::log(u1)) * std::cos(twopi * u2) + 1e-8;
        }
    };
    // every element depends only on its own counter: large vectors are filled by 8 threads (0.2 s -> 0.03 s at 1e7)
    const int T = m > 1000000 ? static_cast<int>(std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency()))) : 1;
    if (T == 1) {
        fill(0, m);
        return;
    }
    std::vector<std::thread> th;
    const int chunk = (m + T - 1) / T;
    for (int t = 0; t < T; ++t) th.emplace_back(fill, std::min(m, t * chunk), std::min(m, (t + 1) * chunk));
    for (auto &x : th) x.join();
}

int log_step(int iter) {
    const double p = std::pow(10.0, std::floor(std::log10(static_cast<double>(iter))));
    const int v = static_cast<int>(p / 10.0);
    return v > 10 ? v : 10;
}

void free_lp_info_cpu_members(LP_info_cpu *model) {
    if (!model) return;
    if (model->A) {
        std::free(model->A->rowPtr);
        std::free(model->A->colIndex);
        std::free(model->A->value);
        std::free(model->A);
        model->A = nullptr;
    }
    std::free(model->AL); model->AL = nullptr;
    std::free(model->AU); model->AU = nullptr;
    std::free(model->c);  model->c = nullptr;
    std::free(model->l);  model->l = nullptr;
    std::free(model->u);  model->u = nullptr;
}

template <class T>
static T *dup_array(const T *src, size_t n) {
    T *d = static_cast<T *>(std::malloc((n ? n : 1) * sizeof(T)));
    if (d && n) std::memcpy(d, src, n * sizeof(T));
    return d;
}

// Builds the model from a CSR triple that has already been validated.
LP_info_cpu *model_from_csr(int m, int n, long nnz, const int *rp, const int *ci, const double *v,
                            const double *AL, const double *AU, const double *l, const double *u,
                            const double *c, double obj_constant) {
    LP_info_cpu *model = static_cast<LP_info_cpu *>(std::calloc(1, sizeof(LP_info_cpu)));
    if (!model) return nullptr;
    model->m = m;
    model->n = n;
    model->obj_constant = obj_constant;
    model->A = static_cast<sparseMatrix *>(std::calloc(1, sizeof(sparseMatrix)));
    if (model->A) {
        model->A->row = m;
        model->A->col = n;
        model->A->numElements = static_cast<int>(nnz);
        model->A->rowPtr = dup_array(rp, static_cast<size_t>(m) + 1);
        model->A->colIndex = dup_array(ci, static_cast<size_t>(nnz));
        model->A->value = dup_array(v, static_cast<size_t>(nnz));
    }
    model->AL = dup_array(AL, m);
    model->AU = dup_array(AU, m);
    model->l = dup_array(l, n);
    model->u = dup_array(u, n);
    model->c = dup_array(c, n);
    if (!model->A || !model->A->rowPtr || !model->A->colIndex || !model->A->value || !model->AL ||
        !model->AU || !model->l || !model->u || !model->c) {
        std::cerr << "[error] Memory allocation failed while building the model" << std::endl;
        free_lp_info_cpu_members(model);
        std::free(model);
        return nullptr;
    }
    return model;
}

}  // namespace hprlp

using namespace hprlp;

extern "C" LP_info_cpu *create_model_from_arrays(int m, int n, int nnz, const int *rowPtr,
                                                 const int *colIndex, const HPRLP_FLOAT *values,
                                                 const HPRLP_FLOAT *AL, const HPRLP_FLOAT *AU,
                                                 const HPRLP_FLOAT *l, const HPRLP_FLOAT *u,
                                                 const HPRLP_FLOAT *c, bool is_csc) {
    try {
        if (m <= 0 || n <= 0 || nnz <= 0) {
            std::cerr << "[error] Invalid dimensions: m=" << m << ", n=" << n << ", nnz=" << nnz << std::endl;
            return nullptr;
        }
        if (!rowPtr || !colIndex || !values || !AL || !AU || !l || !u || !c) {
            std::cerr << "[error] Null pointer in input arrays" << std::endl;
            return nullptr;
        }
        // The pointer array has one entry per stored row: columns of A when is_csc.
        const int srows = is_csc ? n : m;
        const int scols = is_csc ? m : n;
        if (rowPtr[0] != 0 || rowPtr[srows] != nnz) {
            std::cerr << "[error] Invalid " << (is_csc ? "CSC" : "CSR") << " format: pointer[0]=" << rowPtr[0]
                      << ", pointer[" << srows << "]=" << rowPtr[srows] << ", expected 0 and " << nnz << std::endl;
            return nullptr;
        }
        // Not checked by the reference; an out-of-range index would fault on the device.
        for (int i = 0; i < srows; ++i) {
            if (rowPtr[i + 1] < rowPtr[i]) {
                std::cerr << "[error] Invalid sparse format: pointer array decreases at " << i << std::endl;
                return nullptr;
            }
        }
        for (int k = 0; k < nnz; ++k) {
            if (colIndex[k] < 0 || colIndex[k] >= scols) {
                std::cerr << "[error] Invalid sparse format: index " << colIndex[k] << " at position " << k
                          << " outside [0," << scols << ")" << std::endl;
                return nullptr;
            }
        }
        std::cout << "problem information: nRow = " << m << ", nCol = " << n << ", nnz A = " << nnz << std::endl
                  << std::endl;
        if (is_csc) {
            // CSC of A is CSR of A^T: transpose on the host (reference src/HPRLP.cu:354-396).
            std::vector<int> rp, ci;
            std::vector<double> v;
            csr_transpose_host(n, m, nnz, rowPtr, colIndex, values, rp, ci, v);
            LP_info_cpu *mt = model_from_csr(m, n, nnz, rp.data(), ci.data(), v.data(), AL, AU, l, u, c, 0.0);
            warm_for_first_solve();
            return mt;
        }
        LP_info_cpu *mo = model_from_csr(m, n, nnz, rowPtr, colIndex, values, AL, AU, l, u, c, 0.0);
        warm_for_first_solve();  // (abi.cpp: the process-wide part of the first solve, on the calling thread, once)
        return mo;
    } catch (const std::exception &e) {
        std::cerr << "[error] Failed to build model: " << e.what() << std::endl;
        return nullptr;
    }
}

extern "C" void free_model(LP_info_cpu *model) {
    if (!model) return;
    free_lp_info_cpu_members(model);
    std::free(model);
}
